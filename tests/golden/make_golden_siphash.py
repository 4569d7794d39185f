#!/usr/bin/env python3
"""Extracts the reference's known answers for siphash64 into tests/golden/siphash.json (run where /root/reference exists):
  - the scalar and column cases of src/query/functions/tests/it/scalars/testdata/hash.txt whose argument is a String, Timestamp,
    UInt32, Decimal, Date or Boolean (arrays / variants / NULL are outside the scatter path);
  - the bucket_hash_v1 vectors of src/query/functions/src/scalars/hash.rs:563-600 (the same SipHasher13::new_with_keys(0, 0)
    over raw bytes).
Each case = {"what", "type", "value" (or "bytes" hex), "precision", "scale", "expected"}.

    python tests/golden/make_golden_siphash.py
"""
import json
import os
import re

REF = "/root/reference/src/query/functions"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "siphash.json")


def main():
    text = open(os.path.join(REF, "tests/it/scalars/testdata/hash.txt"), encoding="utf-8").read()
    cases = []
    blocks = re.split(r"\n\n\n", text)
    for b in blocks:
        m = re.search(r"^ast\s+: (siphash(?:64)?\(.*\))$", b, re.M)
        if not m:
            continue
        ast = m.group(1)
        chk = re.search(r"^checked expr\s+: (.*)$", b, re.M).group(1)
        out = re.search(r"^output\s+: (\S+)$", b, re.M)
        if out and out.group(1).isdigit():
            exp = int(out.group(1))
            if chk.startswith("siphash64<String>(\""):
                s = re.search(r'<String>\("(.*)"\)', chk).group(1)
                cases.append({"what": ast, "type": "string", "value": s, "expected": exp})
            elif chk.startswith("siphash64<Timestamp>("):
                secs = int(re.search(r"(\d+)_u32", chk).group(1))
                cases.append({"what": ast, "type": "timestamp", "value": secs * 1_000_000, "expected": exp})
            elif chk.startswith("siphash64<UInt32>("):
                cases.append({"what": ast, "type": "u32", "value": int(re.search(r"(\d+)_u32", chk).group(1)), "expected": exp})
            elif chk.startswith("siphash64<Date>("):
                cases.append({"what": ast, "type": "date", "value": int(re.search(r"(\d+)_u32", chk).group(1)), "expected": exp})
            elif chk.startswith("siphash64<Boolean>("):
                cases.append({"what": ast, "type": "bool", "value": "true" in chk, "expected": exp})
            elif chk.startswith("siphash64<Decimal("):
                p, s = map(int, re.search(r"<Decimal\((\d+), (\d+)\)>", chk).groups())
                lit = re.search(r"\(([-0-9.]+)_d", chk).group(1)
                unscaled = int(lit.replace(".", ""))
                cases.append({"what": ast, "type": "decimal64", "value": unscaled, "precision": p, "scale": s, "expected": exp})
        elif chk.startswith("siphash64<String>(a)"):      # the column case
            rows = re.findall(r"^\| Row \d+\s+\| '(.*?)'\s+\| (\d+)\s+\|$", b, re.M)
            for s, e in rows:
                cases.append({"what": ast + " row", "type": "string", "value": s, "expected": int(e)})
    src = open(os.path.join(REF, "src/scalars/hash.rs"), encoding="utf-8").read()
    tests = src[src.index("fn test_bucket_hash_v1_vectors"):]
    flat = re.sub(r"\s+", " ", tests)
    for arg, e in re.findall(r"bucket_hash_v1\((.*?)\), (\d+)", flat):
        arg = arg.strip()
        if arg.startswith('b"'):
            data = arg[2:-1].encode()
        elif arg.startswith('"'):
            data = arg[1:arg.rindex('"')].encode("utf-8")
        else:
            m = re.match(r"&\(?(-?[\w:]+?)(?:_(i32|i64))?\)?\.to_le_bytes\(\)", arg)
            if not m:
                continue
            lit, ty = m.group(1), m.group(2)
            v = {"i64::MIN": -2**63, "i64::MAX": 2**63 - 1}.get(lit)
            if v is None:
                v = int(lit)
            width = 4 if ty == "i32" else 8
            data = v.to_bytes(width, "little", signed=True)
        cases.append({"what": "bucket_hash_v1(" + arg + ")", "type": "bytes", "bytes": data.hex(), "expected": int(e)})
    json.dump(cases, open(OUT, "w"), indent=1, ensure_ascii=False)
    print(len(cases), "cases ->", OUT)


if __name__ == "__main__":
    main()
