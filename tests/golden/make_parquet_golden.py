#!/usr/bin/env python3
"""Generates tests/golden/parquet/*.bin + *.json: small Parquet column chunks written by pyarrow with the reference
writer's settings (storages/common/blocks/src/parquet_rs.rs:91-160) and the values pyarrow reads back from them.
Run from the repo root:  python tests/golden/make_parquet_golden.py   (needs pyarrow; the fixtures are committed)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import pyarrow as pa  # noqa: E402

from tests import parquet_cases as PC  # noqa: E402
from tests import parquet_util as PU  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "parquet")
PICK = {"i64_random_nulls", "i64_lowcard", "i64_runs", "i32_date", "i8", "f64", "bool", "dec15_2_flba", "dec38_6_flba", "dec9_2_int32",
        "str_flag", "str_segment_nulls", "str_unique", "one_null", "single_value_col", "n31_33"}


def main():
    os.makedirs(OUT, exist_ok=True)
    for vi in (0, 1, 4):
        for name, arr, out_type, wkw in PC.make_cases(seed=100 + vi):
            if name not in PICK:
                continue
            arr = arr.slice(0, 300)
            kw = dict(PC.VARIANTS[vi])
            kw.update(wkw)
            if kw.get("page_size"):
                kw["page_size"] = 256
            fb = PU.write_parquet(pa.table({"c": arr}), **kw)
            chunks, back = PU.column_chunks(fb)
            ch = chunks[0]
            exp, valid = PU.expected_of(back.column(0), out_type)
            vals = [None if v is None else (v.hex() if isinstance(v, bytes) else v) for v in exp]
            nm = "%s_v%d" % (name, vi)
            open(os.path.join(OUT, nm + ".bin"), "wb").write(ch["chunk"])
            json.dump(dict(physical=ch["physical"], type_length=ch["type_length"], max_def=ch["max_def"], out_type=out_type, rows=len(exp),
                           nulls=int((~valid).sum()), encodings=list(ch["encodings"]), writer=kw, pyarrow=pa.__version__, values=vals),
                      open(os.path.join(OUT, nm + ".json"), "w"))
    print(len(os.listdir(OUT)), "files")


if __name__ == "__main__":
    main()
