"""Test infrastructure: the reference-held Parquet fixtures (tests/golden/parquet_ref, cut from /root/reference/tests/data by
tests/golden/make_parquet_ref_golden.py) and the checks of a decoder's output against what the REFERENCE's sqllogictests print for those
files. `decode(ch, out_type) -> (python values, valid)` is the decoder under test: the oracle on the CPU, the device through the C-ABI."""
import datetime
import glob
import json
import os

import numpy as np

from databend_amd import _lib as T

HERE = os.path.dirname(os.path.abspath(__file__))
OUT_OF_PHYS = {0: T.T_BOOL, 1: T.T_I32, 2: T.T_I64, 4: T.T_F32, 5: T.T_F64, 6: T.T_STRING}


def fixtures():
    out = {}
    for f in sorted(glob.glob(os.path.join(HERE, "golden", "parquet_ref", "*.json"))):
        d = json.load(open(f))
        blob = open(f[:-5] + ".bin", "rb").read()
        for ch in d["chunks"]:
            ch["chunk"] = blob[ch["offset"]:ch["offset"] + ch["length"]]
        out[os.path.basename(f)[:-5]] = d
    return out


def column(d, name, decode):
    """all row groups of one column, concatenated"""
    vals = []
    for ch in d["chunks"]:
        if ch["column"] != name:
            continue
        v, valid = decode(ch, OUT_OF_PHYS[ch["physical"]])
        assert valid.all()
        vals += v
    return vals


def render(v, kind):
    """a decoded value as the reference's sqllogictest prints it"""
    if kind == "bool":
        return "1" if v else "0"
    if kind in ("f32", "f64"):
        x = float(np.frombuffer(v, np.float32 if kind == "f32" else np.float64)[0])
        return repr(round(x, 6)) if kind == "f32" else repr(x)      # (f32 1.1 prints as 1.1: shortest round-trip of the f32)
    if kind == "str":
        return v.decode()
    if kind == "ts_ns":
        t = datetime.datetime(1970, 1, 1) + datetime.timedelta(microseconds=v // 1000)
        return t.strftime("%Y-%m-%d %H:%M:%S.%f")
    return str(v)


def check_all(decode):
    fx = fixtures()
    checked = 0
    # alltypes_plain: the whole table the reference prints (select_parquet.test:6-16)
    d = fx["alltypes_plain"]
    kinds = {"id": "int", "bool_col": "bool", "tinyint_col": "int", "smallint_col": "int", "int_col": "int", "bigint_col": "int", "float_col": "f32",
             "double_col": "f64", "date_string_col": "str", "string_col": "str", "timestamp_col": "ts_ns"}
    for name, kind in kinds.items():
        got = [render(v, kind) for v in column(d, name, decode)]
        assert got == d["expected"][name], (name, got, d["expected"][name])
        checked += 1
    # binary_view.parquet, written by parquet-rs 58.1.0 (parquet_field_types.test:214-219): the binary column in hex
    d = fx["binary_view"]
    got = [v.hex().upper() for v in column(d, d["chunks"][0]["column"], decode)]
    assert got == d["expected"]["hex"], got
    checked += 1
    # timestamp_{s,ms,us,ns}: four distinct instants, 300 rows each over the 8 row groups (timestamp.test:1-36)
    for unit in ("s", "ms", "us", "ns"):
        d = fx["timestamp_" + unit]
        vals = column(d, "col_timestamp", decode)
        per = d["expected"]["units_per_second"]
        counts = {}
        for v in vals:
            assert v % per == 0
            t = (datetime.datetime(1970, 1, 1) + datetime.timedelta(seconds=v // per)).strftime("%Y-%m-%d %H:%M:%S")
            counts[t] = counts.get(t, 0) + 1
        assert counts == d["expected"]["groups"], (unit, counts)
        checked += 1
    # multi_page_{1..4}: 400 rows over the four files (select_parquet.test:69-72), col_int = 0, 1, 0, 1, ... (gen.py)
    total = 0
    for k in (1, 2, 3, 4):
        d = fx[f"multi_page_{k}"]
        vals = column(d, "col_int", decode)
        assert len(vals) == d["expected"]["rows"] and vals == [0, 1] * (len(vals) // 2)
        total += len(vals)
        checked += 1
    assert total == 400
    # ontime_200 (on_time.test:1-12,54-61)
    d = fx["ontime_200"]
    day, tail, month = column(d, "DayofMonth", decode), column(d, "Tail_Number", decode), column(d, "Month", decode)
    assert len(day) == d["expected"]["rows"] and set(month) == {d["expected"]["month_all"]}
    assert [t.decode() for t, dd in zip(tail, day) if dd == 1] == d["expected"]["tail_number_where_dayofmonth_1"]
    checked += 1
    return checked


def check_lists(decode_list):
    """The List<Int64> column of the reference's multi_page_{1..4}.parquet (col_arr = [[1], [1, 2]] * num_row, gen.py:12; SNAPPY, 128-byte
    data pages: a row group's rows span many pages). `decode_list(ch, list_nullable, element_nullable, out_type) -> python list of lists`."""
    fx = fixtures()
    total = 0
    for k in (1, 2, 3, 4):
        d = fx[f"multi_page_{k}_col_arr"]
        e = d["expected"]
        rows = []
        for ch in d["chunks"]:
            assert ch["max_rep"] == 1 and ch["max_def"] == e["list_nullable"] + 1 + e["element_nullable"]
            rows += decode_list(ch, e["list_nullable"], e["element_nullable"], OUT_OF_PHYS[ch["physical"]])
        assert len(rows) == e["rows"] and rows == e["pattern"] * (e["rows"] // 2), (k, rows[:6])
        total += len(rows)
    assert total == 400
    return 4


def check_tuple(decode):
    """tests/data/parquet/tuple.parquet: `id` as the reference's test implies it (parquet_transform.test:8-17: id + 1 = 2, 3, 4) and the two
    members of the NOT NULL Tuple column t — flat leaves of max_def 0 (a required struct adds no level): a Tuple(..) column is its member
    columns decoded one by one (t.A / t.B expected from pyarrow's reading: the reference's tests do not print them)."""
    d = fixtures()["tuple"]
    e = d["expected"]
    for ch in d["chunks"]:
        assert ch["max_def"] == 0 and ch["max_rep"] == 0
    assert column(d, "id", decode) == e["id"]
    assert column(d, "t.A", decode) == e["t.A"]
    assert [v.decode() for v in column(d, "t.B", decode)] == e["t.B"]
    return 3


def check_map(decode_list):
    """tests/data/parquet/no-stats.parquet, Map(String, String) column `product`: a Map is List<Struct<key NOT NULL, value>>, so its two
    leaves are List leaves over the SAME repetition / definition structure — key: (list_nullable 1, element_nullable 0), value: (1, 1).
    Two List decodes give the Map: equal offsets and list validity, entries = zip(keys, values). 25,825 rows (the count the reference's
    select_parquet.test:53-66 relies on), 188,558 entries; every row against pyarrow's reading (digest) and the first 40 literally."""
    import hashlib
    import json as _json
    d = fixtures()["no_stats_product_map"]
    e = d["expected"]
    kch = [c for c in d["chunks"] if c["column"].endswith(".key")][0]
    vch = [c for c in d["chunks"] if c["column"].endswith(".value")][0]
    assert (kch["max_rep"], kch["max_def"], vch["max_rep"], vch["max_def"]) == (1, 2, 1, 3)
    keys = decode_list(kch, 1, 0, T.T_STRING)
    vals = decode_list(vch, 1, 1, T.T_STRING)
    assert len(keys) == len(vals) == e["rows"]
    maps = []
    for k, v in zip(keys, vals):
        if k is None:
            assert v is None
            maps.append(None)
            continue
        assert v is not None and len(k) == len(v)
        maps.append([[a.decode(), None if b is None else b.decode()] for a, b in zip(k, v)])
    assert sum(len(m) for m in maps if m is not None) == e["entries"]
    assert maps[:40] == e["maps_head"]
    assert hashlib.sha256(_json.dumps(maps).encode()).hexdigest() == e["sha256"]
    return 1
