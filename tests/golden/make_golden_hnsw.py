#!/usr/bin/env python3
"""Known answers of the reference's HNSW + u8 path -> tests/golden/hnsw.json (run in the authoring container).

Source: tests/sqllogictests/suites/query/index/09_vector_index/09_0000_vector_index_base.test:60-335 — tables `t`
(16 x 8-d vectors inserted 4 at a time = four blocks, each with its own index and its own quantisation range),
`t_native` (two blocks of 8), `t2` (4-d columns); index options m=10 ef_construct=40 (:26). Every block holds fewer
vectors than m0, so its graph is complete and a query's printed distances are the post-processed QUANTISED scores
(hnsw.rs:100-140,317-343): they pin EncodedVectorsU8::{encode, encode_query, score_point} and cosine_preprocess to the
printed 8 significant digits, independently of the random graph.
"""
import json
import os
import re
import sys

SRC = "/root/reference/tests/sqllogictests/suites/query/index/09_vector_index/09_0000_vector_index_base.test"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "hnsw.json")


def main():
    if not os.path.exists(SRC):
        print("reference not present; fixture already committed", file=sys.stderr)
        return 0
    text = open(SRC).read()
    tables = {}
    for m in re.finditer(r"INSERT INTO (\w+) VALUES\n(.*?)\n\n", text, re.S):
        rows = []
        for r in re.finditer(r"\((\d+),\s*((?:\[[^\]]*\](?:,\s*)?)+)\)", m.group(2)):
            vecs = [[float(x) for x in v.split(",")] for v in re.findall(r"\[([^\]]*)\]", r.group(2))]
            rows.append({"id": int(r.group(1)), "vectors": vecs})
        tables.setdefault(m.group(1), []).append(rows)          # one entry per INSERT = per block
    queries = []
    for m in re.finditer(r"query IR\nSELECT id, (\w+)_distance\((\w+), \[([^\]]*)\]::vector\(\d+\)\) AS similarity FROM (\w+)( WHERE similarity > ([\d.]+))? "
                         r"ORDER BY similarity (ASC|DESC) LIMIT (\d+);\n----\n(.*?)\n\n", text, re.S):
        res = [(int(a), float(b)) for a, b in (l.split() for l in m.group(9).strip().split("\n"))]
        queries.append({"distance": m.group(1), "column": m.group(2), "query": [float(x) for x in m.group(3).split(",")],
                        "table": m.group(4), "where_gt": float(m.group(6)) if m.group(6) else None, "order": m.group(7),
                        "limit": int(m.group(8)), "expected": res})
    data = {"source": "tests/sqllogictests/suites/query/index/09_vector_index/09_0000_vector_index_base.test:60-335",
            "index_options": {"m": 10, "ef_construct": 40},
            "indexed_tables": ["t", "t_native", "t2"],  # t1 has no index (exact distances)
            "tables": tables, "queries": queries}
    json.dump(data, open(OUT, "w"), indent=0)
    print(f"hnsw.json: {sum(len(b) for b in tables.values())} blocks, {len(queries)} queries")
    return 0


if __name__ == "__main__":
    sys.exit(main())
