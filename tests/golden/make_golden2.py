#!/usr/bin/env python3
"""Extracts the reference's known-answer vectors for hash aggregation, filter / take and sort into JSON fixtures
(run in the authoring container, where /root/reference exists; the fixtures are committed because the GPU box has no
/root/reference).

    python tests/golden/make_golden2.py

Sources:
  aggregates.json  src/query/functions/tests/it/aggregates/testdata/{sum,count,avg,min,max}_group_by.txt
                   (simulate_two_groups_group_by, aggregate_simulation_support.rs:231-262: row i belongs to group i % 2;
                   Output = [group 0, group 1]) and the ungrouped {sum,count,avg,min,max}.txt (one state over all rows)
  kernel.json      src/query/expression/tests/it/testdata/kernel-pass.txt: the Filter / Take sections
                   (tests/it/kernel.rs:49-566: DataBlock::filter_with_bitmap, DataBlock::take)
  sort.json        src/query/expression/tests/it/sort.rs:28-241 (test_block_sort: DataBlock::sort with limit)
"""
import json
import os
import re
import sys

REF = "/root/reference/src/query"
OUT = os.path.dirname(os.path.abspath(__file__))


def parse_validity(v, n):
    bits = []
    for byte in v.strip("[]").split(","):
        b = byte.strip().replace("0b", "").replace("_", "0").rjust(8, "0")
        bits += [c == "1" for c in reversed(b)]
    return bits[:n]


def parse_values(kind, body):
    if kind == "Boolean":
        return body
    out = []
    for tok in body.split(","):
        tok = tok.strip()
        if not tok:
            continue
        out.append(int(tok) if re.fullmatch(r"-?\d+", tok) else tok)
    return out


def parse_data(d):
    """-> dict(kind, values, validity | None, const | None, n | None) or None"""
    d = d.strip()
    m = re.fullmatch(r"Column\((.*)\)", d)
    if m:
        d = m.group(1).strip()
    m = re.fullmatch(r"Const\(Number\((-?[\d.]+)_(\w+)\), Number\((\w+)\), (\d+)\)", d)
    if m:
        return {"kind": m.group(3), "const": m.group(1), "n": int(m.group(4))}
    m = re.fullmatch(r"Const\(Null, Nullable\(Number\((\w+)\)\), (\d+)\)", d)
    if m:
        return {"kind": m.group(1), "const": None, "n": int(m.group(2))}
    vtxt = None
    m = re.fullmatch(r"NullableColumn \{ column: (.*), validity: (\[.*\]) \}", d)
    if m:
        d, vtxt = m.group(1).strip(), m.group(2)
    ms = re.fullmatch(r"StringColumn\[(.*)\]", d)      # min(s) / max(s): 'StringColumn[delta, bravo, charlie, alpha]'
    if ms:
        e = {"kind": "String", "values": [x.strip() for x in ms.group(1).split(",") if x.strip()]}
        if vtxt is not None:
            e["validity"] = parse_validity(vtxt, len(e["values"]))
        return e
    m = re.fullmatch(r"(\w+)\(\[(.*)\]\)", d)
    if not m:
        return None
    kind = m.group(1)
    vals = parse_values(kind, m.group(2))
    e = {"kind": kind, "values": vals}
    if vtxt is not None:
        e["validity"] = parse_validity(vtxt, len(vals) if kind != "Boolean" else 64)
    return e


def table(lines, start):
    rows, i = [], start
    while i < len(lines) and (lines[i].startswith("+") or lines[i].startswith("|")):
        if lines[i].startswith("|"):
            rows.append([c.strip() for c in lines[i].strip().strip("|").split("|")])
        i += 1
    return rows, i


def aggregates():
    cases = []
    base = os.path.join(REF, "functions/tests/it/aggregates/testdata")
    for fn in ("sum", "count", "avg", "min", "max"):
        for suffix, grouped in (("_group_by.txt", True), (".txt", False)):
            path = os.path.join(base, fn + suffix)
            if not os.path.exists(path):
                continue
            lines = open(path, encoding="utf-8").read().splitlines()
            i = 0
            while i < len(lines):
                if lines[i].startswith("ast"):
                    ast = lines[i].split(":", 1)[1].strip()
                    j = i + 1
                    while j < len(lines) and not lines[j].startswith("+") and not lines[j].startswith("ast") and not lines[j].startswith("error"):
                        j += 1
                    if j < len(lines) and lines[j].startswith("+"):
                        rows, j2 = table(lines, j)
                        cols = {}
                        ok = True
                        for r in rows[1:]:
                            p = parse_data("|".join(r[1:]))
                            if p is None:
                                ok = False
                                break
                            cols[r[0]] = p
                        if ok and "Output" in cols:
                            cases.append({"file": fn + suffix, "func": fn, "grouped": grouped, "ast": ast, "columns": cols})
                        i = j2
                        continue
                i += 1
    return {"source": "src/query/functions/tests/it/aggregates/testdata/{sum,count,avg,min,max}[_group_by].txt", "cases": cases}


def text_table(lines, start):
    """'| Column 0 | ...' text table -> (header, rows of cell texts), next index"""
    rows, i = table(lines, start)
    return rows[0], rows[1:], i


def kernel():
    path = os.path.join(REF, "expression/tests/it/testdata/kernel-pass.txt")
    lines = open(path, encoding="utf-8").read().splitlines()
    cases = []
    i = 0
    while i < len(lines):
        m = re.match(r"^(Filter|Take):\s+\[(.*)\]\s*$", lines[i])
        if m and i + 1 < len(lines) and lines[i + 1].startswith("Source:"):
            kind = m.group(1).lower()
            arg = [t.strip() for t in m.group(2).split(",") if t.strip()]
            hdr, src, j = text_table(lines, i + 2)
            assert lines[j].startswith("Result:"), lines[j]
            _, res, j2 = text_table(lines, j + 1)
            cases.append({"kind": kind, "arg": [a == "true" for a in arg] if kind == "filter" else [int(a) for a in arg],
                          "header": hdr, "source": src, "result": res})
            i = j2
            continue
        i += 1
    # Take Block indices / Take Block by slices: several source blocks
    i = 0
    while i < len(lines):
        m = re.match(r"^Take Block (indices|by slices \(limit: (None|Some\((\d+)\))\)):\s+\[(.*)\]\s*$", lines[i])
        if m:
            tuples = [tuple(int(x) for x in t.split(",")) for t in re.findall(r"\(([^()]*)\)", m.group(4))]
            blocks, j = [], i + 1
            hdr = None
            while j < len(lines) and re.match(r"^Block\d+:", lines[j]):
                hdr, rows, j = text_table(lines, j + 1)
                blocks.append(rows)
            assert lines[j].startswith("Result:"), lines[j]
            _, res, j2 = text_table(lines, j + 1)
            cases.append({"kind": "chunks" if m.group(1) == "indices" else "slices", "arg": [list(t) for t in tuples],
                          "limit": int(m.group(3)) if m.group(3) else 0, "header": hdr, "blocks": blocks, "result": res})
            i = j2
            continue
        i += 1
    # Scatter: one source block, one result block per destination (kernel.rs test_pass, DataBlock::scatter)
    i = 0
    while i < len(lines):
        m = re.match(r"^Scatter:\s+\[(.*)\]\s*$", lines[i])
        if m and lines[i + 1].startswith("Source:"):
            arg = [int(t) for t in m.group(1).split(",") if t.strip()]
            hdr, src, j = text_table(lines, i + 2)
            results = []
            while j < len(lines) and re.match(r"^Result-\d+:", lines[j]):
                _, res, j = text_table(lines, j + 1)
                results.append(res)
            cases.append({"kind": "scatter", "arg": arg, "header": hdr, "source": src, "results": results})
            i = j
            continue
        i += 1
    # Concat: 'Concat-Column k' = block k as a (Column ID | Type | Column Data) table; columns of the path's types are parsed,
    # the others (Null, Array(Nothing)) are kept as text
    i = 0
    while i < len(lines):
        if lines[i].startswith("Concat-Column 0:"):
            blocks, j = [], i
            while j < len(lines) and re.match(r"^Concat-Column \d+:", lines[j]):
                _, rows, j = text_table(lines, j + 1)
                blocks.append([concat_column(r[1], r[2]) for r in rows])
            assert lines[j].startswith("Result:"), lines[j]
            hdr, res, j2 = text_table(lines, j + 1)
            cases.append({"kind": "concat", "header": hdr, "blocks": blocks, "result": res})
            i = j2
            continue
        i += 1
    return {"source": "src/query/expression/tests/it/testdata/kernel-pass.txt (Filter / Take / Take Block / Scatter / Concat sections; "
                      "kernel.rs:49-566)", "cases": cases}


def concat_column(type_text, data_text):
    """'Column(NullableColumn { column: UInt8([10, 11]), validity: [0b______10] })' -> {type, values, validity}"""
    out = {"type": type_text, "text": data_text}
    m = re.search(r"(Int32|UInt8)\(\[([^\]]*)\]\)", data_text)
    ms = re.search(r"StringColumn\[([^\]]*)\]", data_text)
    if m:
        out["values"] = [int(x) for x in m.group(2).split(",") if x.strip()]
    elif ms:
        out["values"] = [x.strip() for x in ms.group(1).split(",")]
    else:
        return out
    mv = re.search(r"validity: \[([^\]]*)\]", data_text)
    if mv:
        bits = []
        for byte in mv.group(1).split(","):
            v = int(byte.strip().replace("0b", "").replace("_", "") or "0", 2)
            bits += [(v >> k) & 1 for k in range(8)]
        out["validity"] = bits[:len(out["values"])]
    return out


def sort_cases():
    path = os.path.join(REF, "expression/tests/it/sort.rs")
    text = "\n".join(open(path, encoding="utf-8").read().splitlines()[27:241])
    blocks = re.split(r"DataBlock::new_from_columns\(vec!\[", text)[1:]
    col_re = re.compile(r"(\w+)Type::from_data(?:_with_size)?\(\s*vec!\[(.*?)\]", re.S)

    def cols(s):
        out = []
        for m in col_re.finditer(s):
            kind, body = m.group(1), m.group(2)
            if kind == "String":
                vals = re.findall(r'"([^"]*)"', body)
            else:
                vals = [int(re.sub(r"_?i\d+|_?u\d+", "", t.strip())) for t in body.split(",") if t.strip()]
            out.append({"kind": kind, "values": vals})
        return out
    cases = []
    for b in blocks:
        head, _, rest = b.partition("let test_cases")
        src = cols(head)
        if not src or not rest:
            continue
        for tc in re.split(r"\(\s*vec!\[\s*SortColumnDescription", rest)[1:]:
            descs_txt, _, after = tc.partition("],")
            descs = [{"offset": int(o), "asc": a == "true", "nulls_first": nf == "true"}
                     for o, a, nf in re.findall(r"offset:\s*(\d+),\s*asc:\s*(\w+),\s*nulls_first:\s*(\w+)", "SortColumnDescription" + descs_txt)]
            m = re.match(r"\s*(None|Some\((\d+)\))\s*,", after)
            if not m or not descs:
                continue
            limit = int(m.group(2)) if m.group(2) else 0
            exp = cols(after[m.end():])[:len(src)]
            if len(exp) == len(src):
                cases.append({"source": src, "sort": descs, "limit": limit, "expected": exp})
    return {"source": "src/query/expression/tests/it/sort.rs:28-241 (test_block_sort)", "cases": cases}


def main():
    if not os.path.isdir(REF):
        print("reference not present; fixtures are already committed", file=sys.stderr)
        return 0
    for name, fn in (("aggregates.json", aggregates), ("kernel.json", kernel), ("sort.json", sort_cases)):
        data = fn()
        with open(os.path.join(OUT, name), "w") as f:
            json.dump(data, f, indent=0)
        print(f"{name}: {len(data['cases'])} cases")
    return 0


if __name__ == "__main__":
    sys.exit(main())
