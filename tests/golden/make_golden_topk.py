#!/usr/bin/env python3
"""Extracts the reference's exact (non-indexed, table t1) vector top-5 known answers from
tests/sqllogictests/suites/query/index/09_vector_index/09_0000_vector_index_base.test into
tests/golden/vector_topk.json (run where /root/reference exists; the fixture is committed)."""
import json
import os
import re

SRC = "/root/reference/tests/sqllogictests/suites/query/index/09_vector_index/09_0000_vector_index_base.test"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "vector_topk.json")


def main():
    text = open(SRC).read()
    base = {}
    for m in re.finditer(r"\((\d+), \[([^\]]+)\]\)", text):
        rid = int(m.group(1))
        if rid not in base and 1 <= rid <= 16:
            base[rid] = [float(x) for x in m.group(2).split(",")]
    assert sorted(base) == list(range(1, 17))
    queries = []
    pat = re.compile(r"SELECT id, (\w+)\(embedding, \[([^\]]+)\]::vector\(8\)\) AS similarity FROM t1 ORDER BY similarity ASC LIMIT 5;\n----\n((?:\d+ \S+\n){5})")
    for m in pat.finditer(text):
        rows = [ln.split() for ln in m.group(3).strip().splitlines()]
        queries.append({"fn": m.group(1), "query": [float(x) for x in m.group(2).split(",")],
                        "expected": [[int(a), float(b)] for a, b in rows]})
    json.dump({"source": SRC.replace("/root/reference/", ""), "base": [base[i] for i in range(1, 17)], "queries": queries},
              open(OUT, "w"), indent=1)
    print(len(queries), "queries ->", OUT)


if __name__ == "__main__":
    main()
