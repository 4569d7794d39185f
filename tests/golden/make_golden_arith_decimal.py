#!/usr/bin/env python3
"""Extracts the reference's known answers for a TPC-H-shaped decimal expression and a folded decimal constant
(src/query/functions/tests/it/scalars/testdata/arithmetic_decimal.txt, written by tests/it/scalars/arithmetic.rs) into
tests/golden/arithmetic_decimal.json, in the case format of make_golden.py (tests/golden_eval.py evaluates it).

    python tests/golden/make_golden_arith_decimal.py        (needs /root/reference; the JSON is committed)

`cases`: expressions evaluated over columns (the checked expression carries every node's DecimalSize). `folded`: expressions over
constants only, which the reference folds — kept with the checked expression (the typed tree BEFORE folding), the output DecimalSize
and the printed value, so that the same tree can be evaluated and compared."""
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import make_golden as MG  # noqa: E402

FNAME = "arithmetic_decimal.txt"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "arithmetic_decimal.json")


def main():
    if not os.path.isdir(MG.REF):
        print("reference not present; the fixture is already committed", file=sys.stderr)
        return 0
    text = open(os.path.join(MG.REF, FNAME), encoding="utf-8").read()
    cases, folded = [], []
    for lines in MG.split_cases(text):
        c = MG.parse_case(lines)
        if c and "columns" in c:
            cases.append(c)
            continue
        head = {}
        for ln in lines:
            m = re.match(r"([a-z ]+?) +: (.*)$", ln)
            if m:
                head[m.group(1).strip()] = m.group(2).strip()
        if "checked expr" in head and "output" in head:
            folded.append({"ast": head["ast"], "expr": head["checked expr"], "output_type": head["output type"], "output": head["output"]})
    with open(OUT, "w") as f:
        json.dump({"source": f"src/query/functions/tests/it/scalars/testdata/{FNAME}", "cases": cases, "folded": folded}, f, indent=0)
    print(f"{FNAME}: {len(cases)} column cases, {len(folded)} folded constants -> {OUT}")
    return 0


if __name__ == "__main__":
    sys.exit(main())
