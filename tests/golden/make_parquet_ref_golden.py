"""Column chunks of the Parquet files the REFERENCE keeps under tests/data, cut out as its block reader fetches them (one byte range per
column chunk), next to the values the reference's own tests expect for them — the pin of the scan-side decode that does
not go through pyarrow's reader: the expected values below are transcribed from the reference's sqllogictest result blocks (file:line
cited per case). pyarrow is used here ONLY to read the footer (which byte range is which column, physical type, codec).

    python tests/golden/make_parquet_ref_golden.py      # needs /root/reference; writes tests/golden/parquet_ref/*.{bin,json}

Files and what the reference's tests say about them:
  tests/data/parquet/alltypes_plain.parquet (parquet-cpp-arrow 14, SNAPPY, dictionary-encoded v1 pages)
      select_parquet.test:6-16 `select *`: 8 rows x 11 columns, all listed
  tests/data/parquet/binary_view.parquet (parquet-rs 58.1.0 — the crate version the reference links — UNCOMPRESSED)
      parquet_field_types.test:214-219: column 0 as hex 6162 / 68656C6C6F / 6C61726765207061796C6F6164206F766572203132206279746573
  tests/data/parquet/timestamp/timestamp_{s,ms,us,ns}.parquet (parquet-cpp-arrow 12, SNAPPY, 8 row groups)
      timestamp.test:1-36: four timestamps (2023-10-13 10:00 ... 2023-10-16 12:00), 300 rows each, all 1200 rows
  tests/data/parquet/multi_page/multi_page_{1..4}.parquet (parquet-cpp-arrow 11, SNAPPY, data_page_size 128: many pages per chunk)
      select_parquet.test:69-72 count() = 400 over the four files; gen.py: col_int = [0, 1] * rows
      the SAME files hold col_arr = [[1], [1, 2]] * rows (gen.py:12, "we need multi pages in a column chunk for list type"):
      List<Int64> leaves col_arr.list.item (max_def 3, max_rep 1) whose rows span many 128-byte pages -> multi_page_{k}_col_arr
  tests/data/parquet/tuple.parquet (parquet-cpp-arrow 14, SNAPPY; id INT32 NOT NULL, t Tuple(A INT32, B STRING) NOT NULL: leaves of max_def 0)
      parquet_transform.test:8-17 `select (t.id+1)` -> 2, 3, 4: id = 1, 2, 3; the members t.A / t.B are flat leaves (a NOT NULL struct
      adds no level): their values (1, 3, 3 / a, b, c) are pyarrow's reading — the reference's tests do not print them
  tests/data/parquet/no-stats.parquet (parquet-mr 1.12.2, SNAPPY; four Map(String, ..) columns): `product` is the one with entries
      (188,558 in 25,825 rows). A Map is List<Struct<key NOT NULL, value>>: its two leaves are List leaves that share their levels — key
      (max_def 2 = list_nullable 1 + 1 + element_nullable 0), value (max_def 3) — so a binding decodes a Map with two List decodes;
      the expected entries are pyarrow's reading (select_parquet.test:53-66 only counts this file's rows: 25,825 = the lists' count)
  tests/data/ontime_200.parquet (parquet-cpp-arrow 14, SNAPPY)
      on_time.test:1-12: the nine tail_number values where dayofmonth = 1; :54-61 month = 12"""
import hashlib
import json
import os

import pyarrow.parquet as pq

REF = "/root/reference/tests/data"
HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "parquet_ref")
PHYS = {"BOOLEAN": 0, "INT32": 1, "INT64": 2, "INT96": 3, "FLOAT": 4, "DOUBLE": 5, "BYTE_ARRAY": 6, "FIXED_LEN_BYTE_ARRAY": 7}
CODEC = {"UNCOMPRESSED": 0, "SNAPPY": 1, "GZIP": 2, "ZSTD": 6, "LZ4": 5, "LZ4_RAW": 7}

# select_parquet.test:9-16 (columns: id bool_col tinyint_col smallint_col int_col bigint_col float_col double_col date_string_col string_col timestamp_col)
ALLTYPES = """4 1 0 0 0 0 0.0 0.0 03/01/09 0 2009-03-01 00:00:00.000000
5 0 1 1 1 10 1.1 10.1 03/01/09 1 2009-03-01 00:01:00.000000
6 1 0 0 0 0 0.0 0.0 04/01/09 0 2009-04-01 00:00:00.000000
7 0 1 1 1 10 1.1 10.1 04/01/09 1 2009-04-01 00:01:00.000000
2 1 0 0 0 0 0.0 0.0 02/01/09 0 2009-02-01 00:00:00.000000
3 0 1 1 1 10 1.1 10.1 02/01/09 1 2009-02-01 00:01:00.000000
0 1 0 0 0 0 0.0 0.0 01/01/09 0 2009-01-01 00:00:00.000000
1 0 1 1 1 10 1.1 10.1 01/01/09 1 2009-01-01 00:01:00.000000"""


def chunks_of(path, columns=None, row_groups=None):
    data = open(path, "rb").read()
    pf = pq.ParquetFile(path)
    md = pf.metadata
    out = []
    for g in range(md.num_row_groups):
        if row_groups is not None and g not in row_groups:
            continue
        rg = md.row_group(g)
        for i in range(rg.num_columns):
            c = rg.column(i)
            if columns is not None and c.path_in_schema not in columns:
                continue
            sc = pf.schema.column(i)
            offs = [o for o in (c.data_page_offset, c.dictionary_page_offset if c.has_dictionary_page else None) if o]
            start = min(offs)
            out.append(dict(column=c.path_in_schema, row_group=g, chunk=data[start:start + c.total_compressed_size], physical=PHYS[c.physical_type],
                            type_length=sc.length if sc.length and sc.length > 0 else 0, max_def=sc.max_definition_level, max_rep=sc.max_repetition_level,
                            codec=CODEC.get(c.compression, 99), codec_name=c.compression, encodings=list(c.encodings), num_values=c.num_values,
                            created_by=md.created_by))
    return out


def emit(name, source, cite, chunks, expected):
    os.makedirs(OUT, exist_ok=True)
    blob = b""
    meta = []
    for ch in chunks:
        m = {k: v for k, v in ch.items() if k != "chunk"}
        m["offset"], m["length"] = len(blob), len(ch["chunk"])
        blob += ch["chunk"]
        meta.append(m)
    open(os.path.join(OUT, name + ".bin"), "wb").write(blob)
    json.dump({"source": source, "expected_from": cite, "chunks": meta, "expected": expected}, open(os.path.join(OUT, name + ".json"), "w"), indent=1)
    print(name, len(chunks), "chunks", len(blob), "bytes")


def main():
    # alltypes_plain: every column, expected = the reference's full result
    rows = [r.split(" ") for r in ALLTYPES.splitlines()]
    cols = ["id", "bool_col", "tinyint_col", "smallint_col", "int_col", "bigint_col", "float_col", "double_col", "date_string_col", "string_col", "timestamp_col"]
    exp = {}
    for j, c in enumerate(cols[:10]):
        exp[c] = [r[j] for r in rows]
    exp["timestamp_col"] = [r[10] + " " + r[11] for r in rows]
    emit("alltypes_plain", "tests/data/parquet/alltypes_plain.parquet", "tests/sqllogictests/suites/stage/formats/parquet/select_parquet.test:6-16",
         chunks_of(os.path.join(REF, "parquet/alltypes_plain.parquet")), exp)
    emit("binary_view", "tests/data/parquet/binary_view.parquet (written by parquet-rs 58.1.0)",
         "tests/sqllogictests/suites/stage/formats/parquet/parquet_field_types.test:214-219",
         [c for c in chunks_of(os.path.join(REF, "parquet/binary_view.parquet")) if c["max_rep"] == 0][:1],
         {"hex": ["6162", "68656C6C6F", "6C61726765207061796C6F6164206F766572203132206279746573"]})
    for unit in ("s", "ms", "us", "ns"):
        # the unit of the stored INT64 is the footer's logical type (the file written from second / nanosecond data by pyarrow 12 holds
        # milliseconds / microseconds): it is schema metadata, which the reader hands to the decoder's caller
        lt = str(pq.ParquetFile(os.path.join(REF, f"parquet/timestamp/timestamp_{unit}.parquet")).schema.column(0).logical_type)
        per_s = 10**3 if "milliseconds" in lt else (10**6 if "microseconds" in lt else 10**9)
        emit("timestamp_" + unit, f"tests/data/parquet/timestamp/timestamp_{unit}.parquet", "tests/sqllogictests/suites/stage/formats/parquet/timestamp.test:1-36",
             chunks_of(os.path.join(REF, f"parquet/timestamp/timestamp_{unit}.parquet"), columns=["col_timestamp"]),
             {"units_per_second": per_s, "groups": {"2023-10-13 10:00:00": 300, "2023-10-14 11:00:00": 300, "2023-10-15 12:00:00": 300, "2023-10-16 12:00:00": 300}})
    for k, nrows in ((1, 40), (2, 120), (3, 80), (4, 160)):
        emit(f"multi_page_{k}", f"tests/data/parquet/multi_page/multi_page_{k}.parquet",
             "tests/sqllogictests/suites/stage/formats/parquet/select_parquet.test:69-72 (400 rows over the four files); tests/data/parquet/multi_page/gen.py "
             "(col_int = [0, 1] * rows, 20-row row groups, data_page_size 128)",
             chunks_of(os.path.join(REF, f"parquet/multi_page/multi_page_{k}.parquet"), columns=["col_int"]), {"rows": nrows, "pattern": [0, 1]})
    for k, nrows in ((1, 40), (2, 120), (3, 80), (4, 160)):
        # the List<Int64> column of the same files: the known answer is gen.py's own literal (the reference's tests only count these rows)
        emit(f"multi_page_{k}_col_arr", f"tests/data/parquet/multi_page/multi_page_{k}.parquet",
             "tests/data/parquet/multi_page/gen.py:12 (col_arr = [[1], [1, 2]] * num_row: 'multi pages in a column chunk for list type', "
             "databend PR 11271); rows counted by tests/sqllogictests/suites/stage/formats/parquet/select_parquet.test:69-72",
             chunks_of(os.path.join(REF, f"parquet/multi_page/multi_page_{k}.parquet"), columns=["col_arr.list.item"]),
             {"rows": nrows, "pattern": [[1], [1, 2]], "list_nullable": 1, "element_nullable": 1})
    emit("tuple", "tests/data/parquet/tuple.parquet", "tests/sqllogictests/suites/stage/formats/parquet/parquet_transform.test:8-17 (id + 1 = 2, 3, 4); t.A / t.B: pyarrow",
         chunks_of(os.path.join(REF, "parquet/tuple.parquet")), {"id": [1, 2, 3], "t.A": [1, 3, 3], "t.B": ["a", "b", "c"]})
    prod = pq.read_table(os.path.join(REF, "parquet/no-stats.parquet"), columns=["product"]).column(0).to_pylist()
    emit("no_stats_product_map", "tests/data/parquet/no-stats.parquet", "pyarrow's reading of the Map column `product`; rows counted by "
         "tests/sqllogictests/suites/stage/formats/parquet/select_parquet.test:53-66",
         chunks_of(os.path.join(REF, "parquet/no-stats.parquet"), columns=["product.key_value.key", "product.key_value.value"]),
         {"rows": len(prod), "entries": sum(len(m) for m in prod if m is not None),
          "maps_head": [None if m is None else [[k, v] for k, v in m] for m in prod[:40]],
          # every row, canonically serialised (the fixture stays small): sha256 of json.dumps([[k, v], ...] | None per row)
          "sha256": hashlib.sha256(json.dumps([None if m is None else [[k, v] for k, v in m] for m in prod]).encode()).hexdigest()})
    emit("ontime_200", "tests/data/ontime_200.parquet", "tests/sqllogictests/suites/stage/formats/parquet/on_time.test:1-12,54-61",
         chunks_of(os.path.join(REF, "ontime_200.parquet"), columns=["DayofMonth", "Tail_Number", "Month"]),
         {"tail_number_where_dayofmonth_1": ["N315PQ", "N835AY", "N606LR", "N606LR", "N301PQ", "N176PQ", "N336PQ", "N901XJ", "N909XJ"], "month_all": 12,
          "rows": 199})


if __name__ == "__main__":
    main()
