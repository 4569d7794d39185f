"""CPU: the Parquet column-chunk oracle (oracle/parquet_oracle.c) against pyarrow's own reading of the same files and
against the committed fixtures of tests/golden/parquet/ (made by tests/golden/make_parquet_golden.py)."""
import json
import os

import numpy as np
import pytest

from databend_amd import _lib as T
from tests import parquet_cases as PC
from tests import parquet_util as PU

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "parquet")


def run_case(arr, out_type, variant, wkw):
    import pyarrow as pa
    table = pa.table({"c": arr})
    kw = dict(variant)
    kw.update(wkw)
    fb = PU.write_parquet(table, **kw)
    chunks, back = PU.column_chunks(fb)
    ch = chunks[0]
    exp, exp_valid = PU.expected_of(back.column(0), out_type)
    got, valid, rows, nulls, rc = PU.oracle_decode(ch, out_type)
    assert rc == 0, (rc, ch["encodings"])
    assert rows == len(exp) and nulls == int((~exp_valid).sum())
    assert np.array_equal(valid, exp_valid)
    assert got == exp
    return ch


@pytest.mark.parametrize("vi", range(len(PC.VARIANTS)))
def test_oracle_reads_what_pyarrow_reads(vi):
    seen = set()
    for name, arr, out_type, wkw in PC.make_cases(seed=vi):
        ch = run_case(arr, out_type, PC.VARIANTS[vi], wkw)
        seen.update(ch["encodings"])
    assert "PLAIN" in seen
    if PC.VARIANTS[vi]["dictionary"]:
        assert "RLE_DICTIONARY" in seen or "PLAIN_DICTIONARY" in seen


def test_golden_fixtures():
    """the committed chunks decode to the committed values (which pyarrow produced when the fixtures were made)"""
    names = sorted(f[:-5] for f in os.listdir(GOLD) if f.endswith(".json"))
    assert len(names) >= 10
    for nm in names:
        meta = json.load(open(os.path.join(GOLD, nm + ".json")))
        chunk = open(os.path.join(GOLD, nm + ".bin"), "rb").read()
        ch = dict(chunk=chunk, physical=meta["physical"], type_length=meta["type_length"], max_def=meta["max_def"], num_values=meta["rows"])
        got, valid, rows, nulls, rc = PU.oracle_decode(ch, meta["out_type"])
        assert rc == 0 and rows == meta["rows"] and nulls == meta["nulls"], nm
        exp = meta["values"]
        norm = [None if v is None else (v.hex() if isinstance(v, bytes) else (int(v) if not isinstance(v, bool) else v)) for v in got]
        assert norm == exp, nm


def test_unsupported_and_malformed_are_reported():
    import pyarrow as pa
    import pyarrow.parquet as pq
    import io
    t = pa.table({"c": pa.array(list(range(5000)), pa.int64())})
    buf = io.BytesIO()
    pq.write_table(t, buf, compression="snappy", use_dictionary=False)
    chunks, _ = PU.column_chunks(buf.getvalue())
    assert chunks[0]["codec"] != 0
    assert PU.oracle_decode(chunks[0], T.T_I64)[4] == -2        # compressed page: sizes differ
    buf = io.BytesIO()
    pq.write_table(t, buf, compression="none", use_dictionary=False, column_encoding={"c": "DELTA_BINARY_PACKED"})
    chunks, _ = PU.column_chunks(buf.getvalue())
    assert PU.oracle_decode(chunks[0], T.T_I64)[4] == -2        # encoding outside the writer's repertoire
    fb = PU.write_parquet(t, dictionary=True)
    chunks, _ = PU.column_chunks(fb)
    cut = dict(chunks[0])
    cut["chunk"] = cut["chunk"][: len(cut["chunk"]) // 2]
    assert PU.oracle_decode(cut, T.T_I64)[4] == -1              # truncated chunk
