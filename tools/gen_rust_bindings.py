"""include/dbhip.h -> bindings/dbhip_sys.rs: the bindgen-shaped `extern "C"` block, `#[repr(C)]` structs, opaque handle types and the
enum constants a Rust host (Databend's Function / Processor implementations) links against. There is no Rust toolchain
in this image, so the file cannot be compiled here; tests/test_abi.py regenerates it and checks that the committed file is what the
header produces and that every exported function is bound.
    python tools/gen_rust_bindings.py            # rewrite bindings/dbhip_sys.rs
    python tools/gen_rust_bindings.py --check    # exit 1 if the committed file is stale"""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCALARS = {"int8_t": "i8", "int16_t": "i16", "int32_t": "i32", "int64_t": "i64", "uint8_t": "u8", "uint16_t": "u16", "uint32_t": "u32",
           "uint64_t": "u64", "float": "f32", "double": "f64", "size_t": "usize", "char": "c_char", "void": "c_void", "int": "c_int"}


def strip_comments(text):
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return re.sub(r"//[^\n]*", "", text)


def rust_type(ctype, structs):
    """'const void* const*' -> '*const *const c_void'"""
    t = ctype.strip()
    stars = []
    while True:
        t = t.strip()
        m = re.match(r"^(.*?)(\*)\s*(const)?\s*$", t)
        if not m:
            break
        t = m.group(1)
        stars.append(m.group(3) == "const")          # constness of the POINTER itself (irrelevant in Rust signatures)
    t = t.strip()
    const = False
    if t.startswith("const "):
        const, t = True, t[6:].strip()
    if t.endswith(" const"):
        const, t = True, t[:-6].strip()
    t = t.replace("struct ", "")
    base = SCALARS.get(t, t)
    if base not in SCALARS.values() and base not in structs:
        raise ValueError(f"unknown C type {ctype!r}")
    out = base
    # innermost pointer takes the pointee's constness; outer pointers take the constness written after the inner star
    n = len(stars)
    for level in range(n):
        inner_const = const if level == 0 else stars[n - level]
        out = ("*const " if inner_const else "*mut ") + out
    return out


def parse(header_text):
    text = strip_comments(header_text)
    text = re.sub(r"#[^\n]*", "", text)              # preprocessor lines
    consts, structs, opaques, funcs = [], [], [], []
    # enums (named or anonymous): constants only — the ABI passes them as int32_t
    for m in re.finditer(r"(?:typedef\s+)?enum\s*\w*\s*\{(.*?)\}\s*(\w*)\s*;", text, flags=re.S):
        nxt = 0
        for item in m.group(1).split(","):
            item = item.strip()
            if not item:
                continue
            if "=" in item:
                name, val = [x.strip() for x in item.split("=", 1)]
                nxt = int(val, 0)
            else:
                name = item
            consts.append((name, nxt, m.group(2)))
            nxt += 1
    for m in re.finditer(r"#define\s+(DBHIP_\w+)\s+(\d+)", strip_comments(header_text)):
        consts.append((m.group(1), int(m.group(2)), ""))
    for m in re.finditer(r"typedef\s+struct\s+(\w+)\s+(\w+)\s*;", text):
        opaques.append(m.group(2))
    names = set(opaques)
    for m in re.finditer(r"typedef\s+struct\s*\w*\s*\{(.*?)\}\s*(\w+)\s*;", text, flags=re.S):
        names.add(m.group(2))
    for m in re.finditer(r"typedef\s+struct\s*\w*\s*\{(.*?)\}\s*(\w+)\s*;", text, flags=re.S):
        fields = []
        for decl in m.group(1).split(";"):
            decl = " ".join(decl.split())
            if not decl:
                continue
            # 'int32_t a, b' / 'const void* const* buffers' / 'uint8_t _pad[2]'
            first = re.match(r"^(.*?[\s\*])(\w+(?:\[\d+\])?(?:\s*,\s*\w+(?:\[\d+\])?)*)$", decl)
            ctype, fnames = first.group(1).strip(), [x.strip() for x in first.group(2).split(",")]
            for fn in fnames:
                arr = re.match(r"(\w+)\[(\d+)\]", fn)
                rt = rust_type(ctype, names)
                fields.append((arr.group(1), f"[{rt}; {arr.group(2)}]") if arr else (fn, rt))
        structs.append((m.group(2), fields))
    body = re.sub(r"typedef\s+struct\s*\w*\s*\{.*?\}\s*\w+\s*;", "", text, flags=re.S)
    body = re.sub(r"(?:typedef\s+)?enum\s*\w*\s*\{.*?\}\s*\w*\s*;", "", body, flags=re.S)
    for m in re.finditer(r"([\w\s\*]+?)\b(dbhip_\w+)\s*\(([^()]*)\)\s*;", body):
        ret = " ".join(m.group(1).split())
        if "typedef" in ret or "extern" in ret:
            ret = ret.replace("extern", "").strip()
        args = []
        raw = " ".join(m.group(3).split())
        if raw and raw != "void":
            for i, a in enumerate(raw.split(",")):
                a = a.strip()
                am = re.match(r"^(.*?[\s\*])(\w+)$", a)
                ctype, an = (am.group(1).strip(), am.group(2)) if am else (a, f"arg{i}")
                args.append((an, rust_type(ctype, names)))
        funcs.append((m.group(2), args, rust_type(ret, names) if ret != "void" else None))
    return consts, structs, opaques, funcs


def render(header_text):
    consts, structs, opaques, funcs = parse(header_text)
    out = ["// dbhip_sys.rs — GENERATED from include/dbhip.h by tools/gen_rust_bindings.py (do not edit; `--check` in tests/test_abi.py).",
           "// The raw FFI surface of libdbhip.so for a Rust host: link with `cargo:rustc-link-lib=dylib=dbhip`. Safe wrappers that",
           "// implement Databend's Function / Processor traits over these calls are sketched in INTEGRATION.md.",
           "#![allow(non_camel_case_types, non_upper_case_globals, dead_code)]",
           "use std::os::raw::{c_char, c_int, c_void};", ""]
    seen = set()
    for name, val, enum in consts:
        if name in seen:
            continue
        seen.add(name)
        out.append(f"pub const {name}: i32 = {val};" + (f"   // {enum}" if enum else ""))
    out.append("")
    for o in opaques:
        out += ["#[repr(C)]", f"pub struct {o} {{ _private: [u8; 0] }}"]
    out.append("")
    for name, fields in structs:
        out += ["#[repr(C)]", "#[derive(Clone, Copy)]", f"pub struct {name} {{"]
        out += [f"    pub {'r#type' if fn == 'type' else fn}: {ft}," for fn, ft in fields]
        out += ["}", ""]
    out.append('extern "C" {')
    for name, args, ret in funcs:
        a = ", ".join(f"{'r#type' if an == 'type' else an}: {at}" for an, at in args)
        out.append(f"    pub fn {name}({a})" + (f" -> {ret};" if ret else ";"))
    out += ["}", ""]
    return "\n".join(out), [f[0] for f in funcs]


def main():
    hdr = open(os.path.join(ROOT, "include", "dbhip.h")).read()
    text, _ = render(hdr)
    path = os.path.join(ROOT, "bindings", "dbhip_sys.rs")
    if "--check" in sys.argv:
        cur = open(path).read() if os.path.exists(path) else ""
        sys.exit(0 if cur == text else 1)
    os.makedirs(os.path.dirname(path), exist_ok=True)
    open(path, "w").write(text)
    print(f"wrote {path}: {text.count('pub fn ')} functions")


if __name__ == "__main__":
    main()
