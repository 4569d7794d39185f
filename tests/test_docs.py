"""CPU: the documents point at things that exist — every file under profiles/ that DESIGN.md / README.md / INTEGRATION.md / ROUNDLOG.md
cite is committed, every function include/dbhip.h declares is introduced in INTEGRATION.md or DESIGN.md, and every test DESIGN.md names
is a test of this repository (names of the reference's own tests aside)."""
import glob
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DOCS = ["DESIGN.md", "README.md", "INTEGRATION.md", "ROUNDLOG.md", "profiles/README.md", "oracle/README.md"]


def read(name):
    return open(os.path.join(ROOT, name), encoding="utf-8").read()


def test_cited_profile_files_exist():
    missing = []
    for doc in DOCS:
        for m in re.findall(r"profiles/([A-Za-z0-9_.\-{},*…]+)", read(doc)):
            m = m.rstrip(".,;:)")
            if not m or m == "README.md":
                continue
            if "…" in m or "*" in m:
                if not glob.glob(os.path.join(ROOT, "profiles", m.replace("…", "*"))):
                    missing.append((doc, m))
                continue
            mm = re.match(r"(.*)\{([^}]*)\}(.*)", m)
            names = [mm.group(1) + alt + mm.group(3) for alt in mm.group(2).split(",")] if mm else [m]
            for f in names:
                if not os.path.exists(os.path.join(ROOT, "profiles", f)):
                    missing.append((doc, f))
    assert not missing, missing


def test_every_declared_function_is_introduced_in_the_documents():
    syms = sorted(set(re.findall(r"\b(dbhip_[a-z0-9_]+)\s*\(", read("include/dbhip.h"))))
    text = read("INTEGRATION.md") + read("DESIGN.md")
    assert len(syms) >= 150
    assert not [s for s in syms if s not in text]


def test_tests_named_in_the_design_exist():
    src = "".join(open(f, encoding="utf-8").read() for f in glob.glob(os.path.join(ROOT, "tests", "*.py")))
    defs = set(re.findall(r"def (test_[A-Za-z0-9_]+)", src))
    names = set(re.findall(r"`(test_[A-Za-z0-9_]+)[`\[]", read("DESIGN.md")))
    reference_own = {"test_block_sort"}      # (`sort.rs:28-241` of the reference)
    missing = [n for n in sorted(names - reference_own)
               if n not in defs and not any(d.startswith(n) for d in defs) and not glob.glob(os.path.join(ROOT, "tests", n + "*"))]
    assert not missing, missing
