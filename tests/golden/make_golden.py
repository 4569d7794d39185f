#!/usr/bin/env python3
"""Extracts known-answer vectors from the reference's own golden files into small
JSON fixtures (run in the authoring container, where /root/reference exists; the
fixtures are committed because the GPU box has no /root/reference).

    python tests/golden/make_golden.py

Sources (src/query/functions/tests/it/scalars/testdata/):
    arithmetic.txt  comparison.txt  vector.txt
Each case keeps: the SQL text, the checked/optimized expression, every input
column (type, values, validity) and the Output column (type, values, validity),
or the error text.  Only cases whose columns are numbers / decimals(<=38 digits)
/ booleans / strings / f32 arrays are kept; the test-side evaluator decides
which expression shapes it can drive.
"""
import json
import os
import re
import sys

REF = "/root/reference/src/query/functions/tests/it/scalars/testdata"
OUT = os.path.dirname(os.path.abspath(__file__))


def split_cases(text):
    cases, cur = [], []
    for line in text.splitlines():
        if line.startswith("ast ") and cur:
            cases.append(cur)
            cur = []
        cur.append(line)
    if cur:
        cases.append(cur)
    return cases


def table_rows(lines, start):
    """rows of an ascii table starting at lines[start] (a +---+ line)."""
    rows = []
    i = start
    while i < len(lines) and (lines[i].startswith("+") or lines[i].startswith("|")):
        if lines[i].startswith("|"):
            rows.append([c.strip() for c in lines[i].strip().strip("|").split("|")])
        i += 1
    return rows, i


def parse_list(body):
    """'1, 2, NaN, -inf' -> python numbers (ints kept exact, decimals as strings)."""
    body = body.strip()
    if not body:
        return []
    out = []
    for tok in body.split(","):
        tok = tok.strip()
        if re.fullmatch(r"-?\d+", tok):
            out.append(int(tok))
        elif tok in ("NaN", "inf", "-inf") or re.fullmatch(r"-?\d+\.\d+(e-?\d+)?|-?\d+e-?\d+", tok):
            out.append(tok)  # keep text; test converts (decimal or float) by type
        else:
            return None
    return out


def parse_validity(v, n):
    # "[0b_____011]" possibly several bytes: "[0b11111111, 0b______01]"
    bits = []
    for byte in v.strip("[]").split(","):
        b = byte.strip().replace("0b", "").replace("_", "0")
        b = b.rjust(8, "0")
        bits += [c == "1" for c in reversed(b)]
    return bits[:n]


def parse_internal(data):
    """'Column(Int8([1, 2, 3]))' / 'NullableColumn { column: UInt8([..]), validity: [..] }' / 'Int32([..])'
    / 'Boolean([0b_____101])' -> (kind, values, validity) or None."""
    d = data.strip()
    m = re.fullmatch(r"Column\((.*)\)", d)
    if m:
        d = m.group(1).strip()
    validity_txt = None
    m = re.fullmatch(r"NullableColumn \{ column: (.*), validity: (\[.*\]) \}", d)
    if m:
        d, validity_txt = m.group(1).strip(), m.group(2)
    m = re.fullmatch(r"(\w+)\(\[(.*)\]\)", d)
    if not m:
        return None
    kind, body = m.group(1), m.group(2)
    if kind == "Boolean":
        return kind, body, validity_txt
    vals = parse_list(body)
    if vals is None:
        return None
    return kind, vals, validity_txt


def parse_case(lines):
    case = {}
    for ln in lines:
        for key in ("ast", "raw expr", "checked expr", "optimized expr", "error"):
            if ln.startswith(key):
                rest = ln[len(key):].lstrip()
                if rest.startswith(":"):
                    case[key.replace(" ", "_")] = rest[1:].strip()
    if "error" in " ".join(lines[:4]) and "evaluation:" not in "\n".join(lines):
        err = [ln for ln in lines if ln.startswith("error:") or "--> SQL" in ln]
        case["error_text"] = "\n".join(lines)
        return case if "ast" in case else None
    try:
        ev = lines.index("evaluation:")
        ev2 = lines.index("evaluation (internal):")
    except ValueError:
        return None  # constant-folded or error case
    rows, _ = table_rows(lines, ev + 1)
    header = rows[0][1:]
    types = None
    for r in rows:
        if r[0] == "Type":
            types = r[1:]
    irows, _ = table_rows(lines, ev2 + 1)
    cols = {}
    for r in irows[1:]:
        name, data = r[0], "|".join(r[1:])
        p = parse_internal(data)
        if p is None:
            return None
        cols[name] = p
    if types is None or "Output" not in cols:
        return None
    nrows = sum(1 for r in rows if r[0].startswith("Row "))
    out = {"ast": case.get("ast"), "expr": case.get("optimized_expr", case.get("checked_expr")), "n": nrows, "columns": {}}
    for name, ty in zip(header, types):
        kind, vals, vtxt = cols[name]
        n = nrows
        if kind == "Boolean":
            vals = parse_validity("[" + vals + "]", n)
        entry = {"type": ty, "kind": kind, "values": vals}
        if vtxt is not None:
            entry["validity"] = parse_validity(vtxt, n)
        out["columns"][name] = entry
    # per-row textual output (exact decimal text as the reference prints it)
    out["rows_text"] = [r[1:] for r in rows if r[0].startswith("Row ")]
    return out


def main():
    if not os.path.isdir(REF):
        print("reference not present; fixtures are already committed", file=sys.stderr)
        return 0
    for fname in ("arithmetic.txt", "comparison.txt", "vector.txt"):
        text = open(os.path.join(REF, fname), encoding="utf-8").read()
        kept, total = [], 0
        for lines in split_cases(text):
            total += 1
            try:
                c = parse_case(lines)
            except Exception:
                c = None
            if c and "columns" in c:
                kept.append(c)
        dst = os.path.join(OUT, fname.replace(".txt", ".json"))
        with open(dst, "w") as f:
            json.dump({"source": f"src/query/functions/tests/it/scalars/testdata/{fname}", "cases": kept}, f, indent=0)
        print(f"{fname}: kept {len(kept)} of {total} cases -> {dst}")
    return 0


if __name__ == "__main__":
    sys.exit(main())
