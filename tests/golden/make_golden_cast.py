#!/usr/bin/env python3
"""Number-to-number CAST / TRY_CAST known answers -> tests/golden/cast.json (run in the authoring container).

Source: src/query/functions/tests/it/scalars/testdata/cast.txt — every case whose expression is CAST(<col> AS <number type>)
or TRY_CAST(...) over ONE number column with an `evaluation (internal)` table (register_number_to_number,
scalars/arithmetic/src/arithmetic.rs:448-700), plus the `number overflowed` error cases, which name the offending value
(`to_uint8(512)`): the value must be one the restatement rejects for that destination type."""
import json
import os
import re
import sys

SRC = "/root/reference/src/query/functions/tests/it/scalars/testdata/cast.txt"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "cast.json")
NUM = r"(U?Int(?:8|16|32|64)|Float(?:32|64))"


def values(body):
    out = []
    for tok in body.split(","):
        tok = tok.strip()
        if tok:
            out.append(tok)
    return out


def main():
    if not os.path.exists(SRC):
        print("reference not present; fixture already committed", file=sys.stderr)
        return 0
    text = open(SRC).read()
    cases, errors = [], []
    for block in text.split("\n\n\n"):
        m = re.search(r"^ast\s*: (TRY_CAST|CAST)\((\w+) AS (\w+)( NULL)?\)$", block, re.M)
        if m and "evaluation (internal)" in block:
            cols = dict()
            for name, data in re.findall(r"^\| (\w+)\s+\| (.*?)\s*\|$", block.split("evaluation (internal)")[1], re.M):
                cols[name] = data
            src = cols.get(m.group(2))
            outc = cols.get("Output")
            if not src or not outc:
                continue
            sm = re.fullmatch(r"Column\(" + NUM + r"\(\[(.*)\]\)\)", src) or re.fullmatch(NUM + r"\(\[(.*)\]\)", src)
            om = re.fullmatch(NUM + r"\(\[(.*)\]\)", outc)
            on = re.fullmatch(r"NullableColumn \{ column: " + NUM + r"\(\[(.*)\]\), validity: \[(.*)\] \}", outc)
            if not sm or not (om or on):
                continue
            case = {"ast": m.group(0).split(": ", 1)[1], "try": m.group(1) == "TRY_CAST", "src_type": sm.group(1), "src": values(sm.group(2))}
            if om:
                case.update({"dst_type": om.group(1), "out": values(om.group(2)), "validity": None})
            else:
                bits = []
                for byte in on.group(3).split(","):
                    b = byte.strip().replace("0b", "").replace("_", "0").rjust(8, "0")
                    bits += [c == "1" for c in reversed(b)]
                case.update({"dst_type": on.group(1), "out": values(on.group(2)), "validity": bits[:len(values(on.group(2)))]})
            cases.append(case)
        for e in re.finditer(r"number overflowed while evaluating function `to_(u?int\d+|float\d+)\((-?[\d.e+-]+)\)`", block):
            errors.append({"dst_type": e.group(1), "value": e.group(2)})
    json.dump({"source": "src/query/functions/tests/it/scalars/testdata/cast.txt (number -> number cases)", "cases": cases, "overflow_errors": errors},
              open(OUT, "w"), indent=0)
    print(f"cast.json: {len(cases)} cases, {len(errors)} overflow errors")
    return 0


if __name__ == "__main__":
    sys.exit(main())
