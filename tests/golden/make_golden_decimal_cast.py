#!/usr/bin/env python3
"""Extracts the reference's decimal -> decimal CAST / TRY_CAST known answers
(src/query/functions/tests/it/scalars/testdata/decimal_to_decimal_cast.txt, written by
tests/it/scalars/cast.rs:843-1190 test_decimal_to_decimal) into tests/golden/decimal_cast.json.

    python tests/golden/make_golden_decimal_cast.py        (needs /root/reference; the JSON is committed)

The golden file does not print the FunctionContext: test_cast_decimal_scale_reduction runs every one of its statements twice per
prefix, first with rounding_mode = true, then false (cast.rs:857-858); every other statement runs with the default (false).
So the FIRST occurrence of a statement that occurs twice is the rounding one. A case keeps: the statement, is_try, rounding,
the source (storage class as the golden prints it — Decimal64 / 128 / 256 —, DecimalSize, unscaled integers), the destination
DecimalSize and per row the expected unscaled integer, or null for a row error (CAST) / NULL (TRY_CAST)."""
import json
import os
import re
import sys
from decimal import Decimal

SRC = "/root/reference/src/query/functions/tests/it/scalars/testdata/decimal_to_decimal_cast.txt"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "decimal_cast.json")
SIZE = r"Decimal\((\d+), ?(\d+)\)"


def unscaled(txt, scale):
    return str(int(Decimal(txt).scaleb(scale)))


def blocks(text):
    cur = []
    for line in text.splitlines():
        if (line.startswith("ast ") or line.startswith("error:")) and cur:
            yield cur
            cur = []
        cur.append(line)
    if cur:
        yield cur


def kind_by_precision(p):
    return 64 if p <= 18 else (128 if p <= 38 else 256)


def parse_block(lines):
    text = "\n".join(lines)
    if lines[0].startswith("error:"):
        m = re.search(r"\| ((?:TRY_)?CAST\((-?[\d.]+)::DECIMAL\((\d+),(\d+)\) AS DECIMAL\((\d+),(\d+)\)\))", text)
        if not m:
            return None
        sql, lit, fp, fs, dp, ds = m.group(1), m.group(2), *map(int, m.groups()[2:])
        return dict(sql=sql, is_try=sql.startswith("TRY_"), src=dict(kind=kind_by_precision(fp), p=fp, s=fs, values=[unscaled(lit, fs)]),
                    dst=[dp, ds], expected=[None], error=True)
    sql = re.match(r"ast\s*: (.*)", lines[0]).group(1).strip()
    checked = re.search(r"checked expr\s*: (.*)", text).group(1).strip()
    is_try = sql.startswith("TRY_")
    if "evaluation (internal):" in text:
        m = re.fullmatch(r"(?:TRY_)?CAST<" + SIZE + r">\((\w+) AS " + SIZE + r"(?: NULL)?\)", checked)
        if not m:
            return None
        fp, fs, col, dp, ds = int(m.group(1)), int(m.group(2)), m.group(3), int(m.group(4)), int(m.group(5))
        src_line = re.search(r"\| " + col + r"\s+\| Column\(Decimal(64|128|256)\(\[(.*?)\]\)\)", text)
        out_line = re.search(r"\| Output\s+\| (?:NullableColumn \{ column: )?(?:Column\()?Decimal(?:64|128|256)\(\[(.*?)\]\)\)?(?:, validity: \[(.*?)\] \})?", text)
        if not src_line or not out_line:
            return None
        vals = [unscaled(v.strip(), fs) for v in src_line.group(2).split(",")]
        outs = [unscaled(v.strip(), ds) for v in out_line.group(1).split(",")]
        if out_line.group(2):
            bits = []
            for byte in out_line.group(2).split(","):
                b = byte.strip().replace("0b", "").replace("_", "0").rjust(8, "0")
                bits += [c == "1" for c in reversed(b)]
            outs = [o if bits[i] else None for i, o in enumerate(outs)]
        return dict(sql=sql, is_try=is_try, src=dict(kind=int(src_line.group(1)), p=fp, s=fs, values=vals), dst=[dp, ds], expected=outs, error=False)
    # constant-folded scalar: checked expr holds the typed literal, `output` the answer
    m = re.fullmatch(r"(?:TRY_)?CAST<" + SIZE + r">\((minus<" + SIZE + r">\()?(-?[\d.]+)_d(64|128|256)\((\d+),(\d+)\)\)? AS " + SIZE + r"(?: NULL)?\)", checked)
    o = re.search(r"^output\s*: (.*)$", text, re.M)
    if not m or not o:
        return None
    g = m.groups()
    fp, fs, neg, lit, kind, dp, ds = int(g[0]), int(g[1]), g[2] is not None, g[5], int(g[6]), int(g[9]), int(g[10])
    v = int(unscaled(lit, fs))
    outv = o.group(1).strip()
    return dict(sql=sql, is_try=is_try, src=dict(kind=kind, p=fp, s=fs, values=[str(-v if neg else v)]), dst=[dp, ds],
                expected=[None if outv == "NULL" else unscaled(outv, ds)], error=False)


def main():
    if not os.path.exists(SRC):
        print("reference not present; the fixture is already committed", file=sys.stderr)
        return 0
    cases = [c for c in (parse_block(b) for b in blocks(open(SRC, encoding="utf-8").read())) if c]
    total = {}
    for c in cases:
        total[c["sql"]] = total.get(c["sql"], 0) + 1
    seen = {}
    for c in cases:
        seen[c["sql"]] = seen.get(c["sql"], 0) + 1
        c["rounding"] = total[c["sql"]] == 2 and seen[c["sql"]] == 1
    with open(OUT, "w") as f:
        json.dump({"source": "src/query/functions/tests/it/scalars/testdata/decimal_to_decimal_cast.txt", "cases": cases}, f, indent=0)
    print(f"kept {len(cases)} cases ({sum(c['error'] for c in cases)} errors, {sum(c['rounding'] for c in cases)} with rounding) -> {OUT}")
    return 0


if __name__ == "__main__":
    sys.exit(main())
