"""CPU: the Parquet column-chunk oracle (oracle/parquet_oracle.c) against pyarrow's own reading of the same files and
against the committed fixtures of tests/golden/parquet/ (made by tests/golden/make_parquet_golden.py)."""
import json
import os

import ctypes as C

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
    ts = pa.table({"c": pa.array([b"abc%d" % i for i in range(5000)], pa.binary())})
    pq.write_table(ts, buf, compression="none", use_dictionary=False, column_encoding={"c": "DELTA_LENGTH_BYTE_ARRAY"})
    chunks, _ = PU.column_chunks(buf.getvalue())
    assert PU.oracle_decode(chunks[0], T.T_STRING)[4] == -2     # an encoding that is not decoded (DELTA_LENGTH_BYTE_ARRAY)
    fb = PU.write_parquet(t, dictionary=True)
    chunks, _ = PU.column_chunks(fb)
    cut = dict(chunks[0])
    cut["chunk"] = cut["chunk"][: len(cut["chunk"]) // 2]
    assert PU.oracle_decode(cut, T.T_I64)[4] == -1              # truncated chunk


# ---- the product's HOST-side plan (dbhip_pq_chunk_open touches no device): page / level / dictionary bookkeeping ----
def _open(ch, out_type, chunk=None, **over):
    import ctypes as C
    d = dict(ch)
    d.update(over)
    data = d["chunk"] if chunk is None else chunk
    buf = (C.c_uint8 * max(len(data), 1)).from_buffer_copy(data if data else b"\0")
    h, info = C.c_void_p(), T.PqInfo()
    rc = T.lib().dbhip_pq_chunk_open(buf, C.c_int64(len(data)), d.get("codec", 0), d["physical"], d["type_length"], d["max_def"],
                                     d.get("max_rep", 0), out_type, C.byref(h), C.byref(info))
    if rc == 0:
        T.lib().dbhip_pq_chunk_close(h)
    return rc, info


@pytest.mark.parametrize("vi", [0, 1, 4, 6])
def test_host_plan_counts_rows_nulls_pages_like_pyarrow(vi):
    import io
    import pyarrow as pa
    import pyarrow.parquet as pq
    for name, arr, out_type, wkw in PC.make_cases(seed=vi):
        kw = dict(PC.VARIANTS[vi])
        kw.update(wkw)
        fb = PU.write_parquet(pa.table({"c": arr}), **kw)
        chunks, back = PU.column_chunks(fb)
        rc, info = _open(chunks[0], out_type)
        assert rc == 0, name
        assert info.num_values == len(arr) and info.num_nulls == arr.null_count, name
        assert info.has_validity == chunks[0]["max_def"] and info.out_type == out_type
        es = 0 if out_type == T.T_BOOL else PU.ESIZE[out_type]
        assert info.out_bytes == (len(arr) * es if es else (len(arr) + 63) // 64 * 8), name
        assert info.validity_bytes == (len(arr) + 63) // 64 * 8
        if len(arr):
            assert info.n_pages >= 1
        md = pq.ParquetFile(io.BytesIO(fb)).metadata.row_group(0).column(0)
        assert (info.n_dict_values > 0) <= md.has_dictionary_page


def test_host_plan_survives_mutated_chunks():
    """bit flips / truncation / overwritten bytes: open() answers OK, INVALID or UNSUPPORTED — it never reads outside the
    chunk (a crash of this test process is the failure mode being guarded)"""
    import pyarrow as pa
    rng = np.random.default_rng(5)
    seeds = []
    for vi in (0, 1, 4, 6):
        for name, arr, ot, wkw in PC.make_cases(seed=vi):
            kw = dict(PC.VARIANTS[vi])
            kw.update(wkw)
            chunks, _ = PU.column_chunks(PU.write_parquet(pa.table({"c": arr.slice(0, 1500)}), **kw))
            if len(chunks[0]["chunk"]):
                seeds.append((chunks[0], ot))
    seen = set()
    for it in range(4000):
        ch, ot = seeds[it % len(seeds)]
        b = bytearray(ch["chunk"])
        k = it % 3
        if k == 0:
            for _ in range(int(rng.integers(1, 4))):
                b[int(rng.integers(0, len(b)))] ^= 1 << int(rng.integers(0, 8))
        elif k == 1:
            b = b[: int(rng.integers(0, len(b)))]
        else:
            i = int(rng.integers(0, len(b)))
            b[i:i + 6] = bytes(rng.integers(0, 256, 6).astype(np.uint8))
        rc, _ = _open(ch, ot, chunk=bytes(b))
        assert rc in (T.OK, T.ERR_INVALID, T.ERR_UNSUPPORTED), rc
        seen.add(rc)
    assert seen == {T.OK, T.ERR_INVALID, T.ERR_UNSUPPORTED}


# ---- compressed chunks: pages are decompressed on the host inside open(); the device decodes the decompressed IMAGE ----
CODECS = [("snappy", T.PQ_SNAPPY), ("zstd", T.PQ_ZSTD), ("lz4", T.PQ_LZ4_RAW)]


def _image(ch, out_type, codec):
    import ctypes as C
    data = ch["chunk"]
    buf = (C.c_uint8 * max(len(data), 1)).from_buffer_copy(data if data else b"\0")
    h, info = C.c_void_p(), T.PqInfo()
    rc = T.lib().dbhip_pq_chunk_open(buf, C.c_int64(len(data)), codec, ch["physical"], ch["type_length"], ch["max_def"], 0, out_type,
                                     C.byref(h), C.byref(info))
    if rc:
        return rc, None, info
    p, n = C.POINTER(C.c_uint8)(), C.c_int64()
    T.check(T.lib().dbhip_pq_chunk_image(h, C.byref(p), C.byref(n)))
    img = bytes(np.ctypeslib.as_array(p, shape=(n.value,))) if n.value else b""
    T.lib().dbhip_pq_chunk_close(h)
    return 0, img, info


def _payloads(chunk):
    import ctypes as C
    from tests import oracle_lib
    L = oracle_lib.load()
    L.orc_pq_payloads.restype = C.c_int64
    src = np.frombuffer(chunk, dtype=np.uint8)
    out = np.zeros(len(chunk) + 16, dtype=np.uint8)
    n = L.orc_pq_payloads(src.ctypes.data_as(C.c_void_p), C.c_int64(len(chunk)), out.ctypes.data_as(C.c_void_p), C.c_int64(len(out)))
    assert n >= 0
    return out[:n].tobytes()


@pytest.mark.parametrize("cname,codec", CODECS)
@pytest.mark.parametrize("vi", [0, 1, 4, 6])
def test_compressed_chunks_decompress_to_the_uncompressed_twins_payloads(cname, codec, vi):
    """The same table written with and without compression has the same pages; the image open() builds from the compressed
    chunk (own Snappy decoder, libzstd / liblz4) must be byte-identical to the page payloads of the uncompressed twin, and the
    plan's row / null counts must agree."""
    import pyarrow as pa
    for name, arr, out_type, wkw in PC.make_cases(seed=vi):
        kw = dict(PC.VARIANTS[vi])
        kw.update(wkw)
        t = pa.table({"c": arr})
        plain, _ = PU.column_chunks(PU.write_parquet(t, **kw))
        comp, _ = PU.column_chunks(PU.write_parquet(t, compression=cname, **kw))
        assert comp[0]["codec"] != 0 or len(arr) == 0
        rc, img, info = _image(comp[0], out_type, codec)
        assert rc == 0, (name, rc, T.lib().dbhip_last_error())
        assert img == _payloads(plain[0]["chunk"]), name
        assert info.image_bytes == len(img) and info.num_values == len(arr) and info.num_nulls == arr.null_count


def test_corrupt_compressed_pages_are_rejected_not_trusted():
    import pyarrow as pa
    rng = np.random.default_rng(3)
    arr = pa.array(["row %d %s" % (i, "abc" * (i % 7)) for i in range(4000)], pa.string())
    outcomes = set()
    for cname, codec in CODECS:
        comp, _ = PU.column_chunks(PU.write_parquet(pa.table({"c": arr}), compression=cname, dictionary=False, page_size=2048))
        good = comp[0]["chunk"]
        assert _image(comp[0], T.T_STRING, codec)[0] == 0
        for it in range(600):
            b = bytearray(good)
            for _ in range(int(rng.integers(1, 5))):
                b[int(rng.integers(0, len(b)))] = int(rng.integers(0, 256))
            if it % 5 == 0:
                b = b[: int(rng.integers(1, len(b)))]
            ch = dict(comp[0])
            ch["chunk"] = bytes(b)
            rc, _, _ = _image(ch, T.T_STRING, codec)
            assert rc in (T.OK, T.ERR_INVALID, T.ERR_UNSUPPORTED)
            outcomes.add(rc)
    assert T.ERR_INVALID in outcomes
    # a codec the library does not know
    ch = dict(comp[0])
    assert _image(ch, T.T_STRING, 2)[0] == T.ERR_UNSUPPORTED     # GZIP


@pytest.mark.parametrize("cname,codec", [("none", T.PQ_UNCOMPRESSED), ("zstd", T.PQ_ZSTD)])
@pytest.mark.parametrize("vi", range(len(PC.VARIANTS)))
def test_host_decodes_definition_levels_into_the_validity_bitmap(cname, codec, vi):
    """open() turns the definition levels (RLE / bit-packed hybrid, runs cut at page boundaries) into the column's validity bitmap
    on the host; it must be pyarrow's validity, with the padding bits clear."""
    import ctypes as C
    import pyarrow as pa
    for name, arr, out_type, wkw in PC.make_cases(seed=vi):
        kw = dict(PC.VARIANTS[vi])
        kw.update(wkw)
        chunks, back = PU.column_chunks(PU.write_parquet(pa.table({"c": arr}), compression=cname, **kw))
        ch = chunks[0]
        data = ch["chunk"]
        buf = (C.c_uint8 * max(len(data), 1)).from_buffer_copy(data if data else b"\0")
        h, info = C.c_void_p(), T.PqInfo()
        T.check(T.lib().dbhip_pq_chunk_open(buf, C.c_int64(len(data)), codec, ch["physical"], ch["type_length"], ch["max_def"], 0, out_type,
                                            C.byref(h), C.byref(info)))
        p, n = C.POINTER(C.c_uint8)(), C.c_int64()
        T.check(T.lib().dbhip_pq_chunk_validity(h, C.byref(p), C.byref(n)))
        exp = np.array([v is not None for v in back.column(0).to_pylist()], dtype=bool)
        if ch["max_def"] == 0 or len(exp) == 0:
            assert n.value == 0
        else:
            bits = np.unpackbits(np.ctypeslib.as_array(p, shape=(n.value,)).copy(), bitorder="little").astype(bool)
            assert n.value == (len(exp) + 63) // 64 * 8
            assert np.array_equal(bits[: len(exp)], exp), name
            assert not bits[len(exp):].any(), name
            assert info.num_nulls == int((~exp).sum())
        T.lib().dbhip_pq_chunk_close(h)


def test_headers_that_claim_absurd_sizes_are_refused_before_memory_is_reserved():
    """page headers are untrusted: an all-NULL page can claim 2^31 rows with a 6-byte RLE run, a compressed page any
    uncompressed size — open() must answer from the headers alone, quickly and without allocating for the claim"""
    import time

    def zz(v):          # zigzag varint
        v = (v << 1) ^ (v >> 63)
        out = bytearray()
        while True:
            b = v & 0x7F
            v >>= 7
            out.append(b | (0x80 if v else 0))
            if not v:
                return bytes(out)

    def varint(v):
        out = bytearray()
        while True:
            b = v & 0x7F
            v >>= 7
            out.append(b | (0x80 if v else 0))
            if not v:
                return bytes(out)

    nv = (1 << 31) - 1
    levels = varint(nv << 1) + b"\x00"                      # one RLE run: nv x level 0 (all NULL)
    payload = len(levels).to_bytes(4, "little") + levels
    page = (b"\x15" + zz(0) + b"\x15" + zz(len(payload)) + b"\x15" + zz(len(payload)) + b"\x2c" + b"\x15" + zz(nv) + b"\x15" + zz(0) +
            b"\x15" + zz(3) + b"\x15" + zz(3) + b"\x00" + b"\x00" + payload)
    ch = dict(chunk=page * 3, physical=2, type_length=0, max_def=1)
    t0 = time.perf_counter()
    rc, info = _open(ch, T.T_I64)
    assert rc == T.ERR_UNSUPPORTED and time.perf_counter() - t0 < 5.0       # 3 x 2^31 rows: refused at the second page
    one = dict(ch, chunk=page)
    rc, info = _open(one, T.T_I64)
    assert rc == 0 and info.num_values == nv and info.num_nulls == nv            # a single such page is legitimate
    # a "compressed" page that claims 3 GiB of output for 20 bytes of input
    big = b"\x15" + zz(0) + b"\x15" + zz(3 << 30) + b"\x15" + zz(20) + b"\x2c" + b"\x15" + zz(10) + b"\x15" + zz(0) + b"\x15" + zz(3) + b"\x15" + zz(3) + b"\x00\x00" + b"\x00" * 20
    rc, _ = _open(dict(chunk=big, physical=2, type_length=0, max_def=0, codec=T.PQ_ZSTD), T.T_I64)
    assert rc == T.ERR_INVALID


def test_oracle_decodes_the_reference_held_parquet_files_to_what_its_tests_print():
    """The pin of the scan-side decode that does not go through pyarrow's reader: column chunks cut from the Parquet files the reference
    keeps under tests/data (written by parquet-cpp, parquet-mr and parquet-rs 58.1.0 — the crate the reference links) must decode to the
    values the reference's own sqllogictests print (select_parquet.test:6-16,69-72, parquet_field_types.test:214-219, timestamp.test:1-36,
    on_time.test:1-12,54-61). Fixtures: tests/golden/parquet_ref (make_parquet_ref_golden.py)."""
    from tests import parquet_ref as PR

    import ctypes as C

    def decode(ch, out_type):
        from tests import oracle_lib
        L = oracle_lib.load()
        L.orc_pq_decode_codec.restype = C.c_int
        n = ch["num_values"]
        chunk = np.frombuffer(ch["chunk"], dtype=np.uint8)
        es = 1 if out_type == T.T_BOOL else PU.ESIZE[out_type]
        vals = np.zeros(max(n, 1) * es + 16, dtype=np.uint8)
        valid = np.zeros(max(n, 1), dtype=np.uint8)
        rows, nulls = C.c_int64(), C.c_int64()
        rc = L.orc_pq_decode_codec(chunk.ctypes.data_as(C.c_void_p), C.c_int64(len(chunk)), ch["codec"], ch["physical"], ch["type_length"], ch["max_def"],
                                   out_type, C.c_int64(n), vals.ctypes.data_as(C.c_void_p), valid.ctypes.data_as(C.c_void_p), C.byref(rows), C.byref(nulls))
        assert rc == 0 and rows.value == n, (ch["column"], rc)
        v = valid[:n].astype(bool)
        py = [bool(vals[i]) for i in range(n)] if out_type == T.T_BOOL else PU.decoded_to_python(vals.tobytes(), v, out_type, n, chunk)
        return py, v
    assert PR.check_all(decode) == 21
    assert PR.check_tuple(decode) == 3     # (round 6) the members of a NOT NULL Tuple column are flat leaves


def test_oracle_delta_binary_packed_matches_pyarrow():
    """DELTA_BINARY_PACKED INT32 / INT64 pages (Encodings.md "Delta Encoding"): the oracle's statement against pyarrow's reader."""
    import io
    import pyarrow as pa
    import pyarrow.parquet as pq
    rng = np.random.default_rng(3)
    for typ, ot, hi in ((pa.int64(), T.T_I64, 2**62), (pa.int32(), T.T_I32, 2**30), (pa.int32(), T.T_I64, 2**20)):
        for n, frac in ((1, 0.0), (129, 0.0), (9000, 0.1)):
            a = rng.integers(-hi, hi, n)
            a[n // 4: n // 2] = np.arange(n // 2 - n // 4) * 3 + 7
            a[n // 2: n // 2 + n // 8] = 42
            t = pa.table({"c": pa.array(a, typ, mask=rng.random(n) < frac)})
            buf = io.BytesIO()
            pq.write_table(t, buf, compression="none", use_dictionary=False, column_encoding={"c": "DELTA_BINARY_PACKED"}, data_page_size=3000,
                           write_statistics=False)
            chunks, back = PU.column_chunks(buf.getvalue())
            assert "DELTA_BINARY_PACKED" in chunks[0]["encodings"]
            exp, _ = PU.expected_of(back.column(0), ot)
            got, _, rows, _, rc = PU.oracle_decode(chunks[0], ot)
            assert rc == 0 and rows == n and got == exp


def test_device_mode_open_reads_only_the_page_headers():
    """dbhip_pq_chunk_open_device needs no device: rows / pages / dictionary size come from the thrift headers alone; the null count of v1
    pages is not known before the levels are walked (-1); ZSTD is left to the host mode; the image is the 16-byte aligned page payloads."""
    import ctypes as C
    import pyarrow as pa
    rng = np.random.default_rng(2)
    n = 30_000
    arr = pa.array(rng.integers(0, 50, n), pa.int64(), mask=rng.random(n) < 0.1)

    def open_dev(ch):
        data = ch["chunk"]
        buf = (C.c_uint8 * max(len(data), 1)).from_buffer_copy(data)
        h, info = C.c_void_p(), T.PqInfo()
        rc = T.lib().dbhip_pq_chunk_open_device(buf, C.c_int64(len(data)), ch["codec"], ch["physical"], ch["type_length"], ch["max_def"], 0, T.T_I64,
                                                C.byref(h), C.byref(info))
        if rc == 0:
            T.lib().dbhip_pq_chunk_close(h)
        return rc, info
    for cname in ("none", "snappy", "lz4", "zstd"):
        for dictionary in (True, False):
            chunks, _ = PU.column_chunks(PU.write_parquet(pa.table({"c": arr}), dictionary=dictionary, compression=cname, page_size=8192))
            rc, info = open_dev(chunks[0])
            assert rc == 0 and info.num_values == n and info.has_validity == 1 and info.n_pages > 1
            assert info.num_nulls == -1 and (info.n_dict_values == 50) == dictionary
            assert (info.image_bytes > 0) == (cname != "none")
    chunks, _ = PU.column_chunks(PU.write_parquet(pa.table({"c": arr}), compression="gzip"))     # (the reference writes none / lz4 / snappy / zstd)
    assert open_dev(chunks[0])[0] == T.ERR_UNSUPPORTED
    chunks, _ = PU.column_chunks(PU.write_parquet(pa.table({"c": pa.array(np.arange(n), pa.int64())})))
    rc, info = open_dev(chunks[0])
    assert rc == 0 and info.num_nulls == 0 and info.image_bytes == 0
    cut = dict(chunks[0])
    cut["chunk"] = cut["chunk"][: len(cut["chunk"]) // 2]
    assert open_dev(cut)[0] == T.ERR_INVALID


def test_device_mode_open_survives_mutated_chunks():
    """bit flips / truncation / overwritten bytes through dbhip_pq_chunk_open_device (the thrift header walk is all it does): OK, INVALID or
    UNSUPPORTED — it never reads outside the chunk and never sizes an image out of proportion to it (a crash or a hang of this process is
    what is being guarded)"""
    import ctypes as C
    import pyarrow as pa
    rng = np.random.default_rng(6)
    seeds = []
    for vi, cname in ((0, "none"), (1, "snappy"), (4, "lz4"), (6, "none")):
        for name, arr, ot, wkw in PC.make_cases(seed=vi):
            kw = dict(PC.VARIANTS[vi])
            kw.update(wkw)
            chunks, _ = PU.column_chunks(PU.write_parquet(pa.table({"c": arr.slice(0, 1500)}), compression=cname, **kw))
            if len(chunks[0]["chunk"]):
                seeds.append((chunks[0], ot))
    seen = set()
    for it in range(4000):
        ch, ot = seeds[it % len(seeds)]
        b = bytearray(ch["chunk"])
        k = it % 3
        if k == 0:
            for _ in range(int(rng.integers(1, 4))):
                b[int(rng.integers(0, len(b)))] ^= 1 << int(rng.integers(0, 8))
        elif k == 1:
            b = b[: int(rng.integers(0, len(b)))]
        else:
            i = int(rng.integers(0, len(b)))
            b[i:i + 6] = bytes(rng.integers(0, 256, 6).astype(np.uint8))
        data = bytes(b)
        buf = (C.c_uint8 * max(len(data), 1)).from_buffer_copy(data if data else b"\0")
        h, info = C.c_void_p(), T.PqInfo()
        rc = T.lib().dbhip_pq_chunk_open_device(buf, C.c_int64(len(data)), ch["codec"], ch["physical"], ch["type_length"], ch["max_def"], 0, ot,
                                                C.byref(h), C.byref(info))
        assert rc in (T.OK, T.ERR_INVALID, T.ERR_UNSUPPORTED), rc
        if rc == T.OK:
            assert 0 <= info.image_bytes <= max(1 << 30, 1024 * len(data)) + 16 and info.num_values >= 0
            T.lib().dbhip_pq_chunk_close(h)
        seen.add(rc)
    assert T.OK in seen and T.ERR_INVALID in seen


# ---- List<primitive> (round 5): the oracle's record assembly against pyarrow's reading of the same file -------------------------------
def _list_arrays(rng, n):
    import pyarrow as pa
    lens = rng.integers(0, 6, n)
    lens[rng.random(n) < 0.1] = 0
    null_list = rng.random(n) < 0.08

    def lists(make, elem_null, list_null=True):
        return [None if (list_null and null_list[i]) else [None if (elem_null and rng.random() < 0.2) else make() for _ in range(lens[i])] for i in range(n)]
    return {
        "list<int64> both nullable": (pa.array(lists(lambda: int(rng.integers(-10**12, 10**12)), True), pa.list_(pa.int64())), 1, 1, T.T_I64),
        "list<int32 not null> nullable": (pa.array(lists(lambda: int(rng.integers(0, 50)), False), pa.list_(pa.field("item", pa.int32(), nullable=False))), 1, 0, T.T_I32),
        "required list<double>": (pa.array(lists(lambda: float(rng.integers(-1000, 1000)) / 8, True, False), pa.list_(pa.float64())), 0, 1, T.T_F64),
        "required list<int64 not null>": (pa.array(lists(lambda: int(rng.integers(0, 2**31)), False, False), pa.list_(pa.field("item", pa.int64(), nullable=False))), 0, 0, T.T_I64),
        "list<binary> both nullable": (pa.array(lists(lambda: (b"s%d" % rng.integers(0, 10**6)) * int(rng.integers(1, 4)), True), pa.list_(pa.binary())), 1, 1, T.T_STRING),
        "list<bool> both nullable": (pa.array(lists(lambda: bool(rng.integers(0, 2)), True), pa.list_(pa.bool_())), 1, 1, T.T_BOOL),
    }


@pytest.mark.parametrize("v2,dictionary", [(False, False), (True, True), (True, False)])
@pytest.mark.parametrize("n", [1, 300, 20_000])
def test_oracle_list_decode_equals_pyarrow(n, v2, dictionary):
    """orc_pq_decode_list (repetition / definition levels -> offsets, list validity, elements, element validity) == pyarrow's reading of
    the same file, for every nullability combination of the three-level LIST, v1 / v2 pages, PLAIN and dictionary values, several pages"""
    import io
    import pyarrow as pa
    import pyarrow.parquet as pq
    from tests import oracle_lib
    orc = oracle_lib.load()
    orc.orc_pq_decode_list.restype = C.c_int
    rng = np.random.default_rng(n + 7 * v2 + dictionary)
    for name, (arr, ln, en, ot) in _list_arrays(rng, n).items():
        table = pa.Table.from_arrays([arr], schema=pa.schema([pa.field("c", arr.type, nullable=bool(ln))]))
        data = PU.write_parquet(table, dictionary=dictionary, v2=v2, page_size=4096)
        ch = PU.column_chunks(data)[0][0]
        assert ch["max_rep"] == 1 and ch["max_def"] == ln + 1 + en
        back = pq.read_table(io.BytesIO(data)).column(0).to_pylist()
        ent = ch["num_values"]
        chunk = np.frombuffer(ch["chunk"], dtype=np.uint8)
        es = 1 if ot == T.T_BOOL else PU.ESIZE[ot]
        offs = np.zeros(ent + 2, np.uint64)
        lval = np.zeros(ent + 1, np.uint8)
        vals = np.zeros(max(ent, 1) * es + 16, np.uint8)
        ev = np.zeros(ent + 1, np.uint8)
        rows, elems = C.c_int64(), C.c_int64()
        rc = orc.orc_pq_decode_list(chunk.ctypes.data_as(C.c_void_p), C.c_int64(len(chunk)), ch["physical"], ch["type_length"], ln, en, ot, C.c_int64(ent),
                                    offs.ctypes.data_as(C.c_void_p), lval.ctypes.data_as(C.c_void_p), vals.ctypes.data_as(C.c_void_p), ev.ctypes.data_as(C.c_void_p),
                                    C.byref(rows), C.byref(elems))
        assert rc == 0 and rows.value == n and int(offs[n]) == elems.value, (name, rc)
        m = elems.value
        valid = ev[:m].astype(bool)
        if ot == T.T_BOOL:
            py = [bool(vals[i]) if valid[i] else None for i in range(m)]
        elif ot == T.T_F64:
            py = [float(np.frombuffer(vals[8 * i:8 * i + 8].tobytes(), np.float64)[0]) if valid[i] else None for i in range(m)]
        else:
            py = PU.decoded_to_python(vals.tobytes(), valid, ot, m, chunk)
        got = [None if (ln and not lval[r]) else py[int(offs[r]):int(offs[r + 1])] for r in range(n)]
        assert got == back, name


def _open_list(ch, ln, en, ot):
    data = ch["chunk"]
    buf = (C.c_uint8 * max(len(data), 1)).from_buffer_copy(data if data else b"\0")
    h, info = C.c_void_p(), T.PqInfo()
    rc = T.lib().dbhip_pq_chunk_open_device_list(buf, C.c_int64(len(data)), C.c_int32(ch["codec"]), C.c_int32(ch["physical"]), C.c_int32(ch["type_length"]),
                                                 C.c_int32(ln), C.c_int32(en), C.c_int32(ot), C.byref(h), C.byref(info))
    if rc == 0:
        T.lib().dbhip_pq_chunk_close(h)
    return rc, info


@pytest.mark.parametrize("cname", ["none", "snappy", "lz4", "zstd"])
def test_device_mode_list_open_reads_only_the_page_headers(cname):
    """dbhip_pq_chunk_open_device_list needs no device either: it walks the thrift page headers of a List<primitive> leaf. num_values is
    the number of LEVEL ENTRIES of the leaf (what pyarrow's column-chunk metadata calls num_values), several pages, an image only for
    compressed chunks; the flat open refuses the same chunk (max_rep 1) and names the List entry point; a truncated chunk is INVALID."""
    import pyarrow as pa
    rng = np.random.default_rng(11)
    n = 6000
    for name, (arr, ln, en, ot) in _list_arrays(rng, n).items():
        table = pa.Table.from_arrays([arr], schema=pa.schema([pa.field("c", arr.type, nullable=bool(ln))]))
        for v2 in (False, True):
            ch = PU.column_chunks(PU.write_parquet(table, dictionary=(ot != T.T_BOOL), v2=v2, page_size=4096, compression=cname))[0][0]
            rc, info = _open_list(ch, ln, en, ot)
            assert rc == 0, (name, v2, T.lib().dbhip_last_error())
            assert info.num_values == ch["num_values"] and info.n_pages >= (1 if ot == T.T_BOOL else 2), (name, info.num_values, ch["num_values"], info.n_pages)
            assert (info.image_bytes > 0) == (cname != "none")
            buf = (C.c_uint8 * len(ch["chunk"])).from_buffer_copy(ch["chunk"])
            h, finfo = C.c_void_p(), T.PqInfo()
            rc = T.lib().dbhip_pq_chunk_open_device(buf, C.c_int64(len(ch["chunk"])), ch["codec"], ch["physical"], ch["type_length"], ch["max_def"], ch["max_rep"], ot,
                                                    C.byref(h), C.byref(finfo))
            assert rc == T.ERR_UNSUPPORTED and b"open_device_list" in T.lib().dbhip_last_error()
            cut = dict(ch)
            cut["chunk"] = ch["chunk"][: len(ch["chunk"]) // 2]
            assert _open_list(cut, ln, en, ot)[0] == T.ERR_INVALID


def test_device_mode_list_open_survives_mutated_chunks():
    """the List open under bit flips / truncation / overwritten bytes: OK, INVALID or UNSUPPORTED, an image in proportion to the chunk"""
    import pyarrow as pa
    rng = np.random.default_rng(12)
    seeds = []
    for cname in ("none", "snappy", "zstd"):
        for name, (arr, ln, en, ot) in _list_arrays(rng, 1200).items():
            table = pa.Table.from_arrays([arr], schema=pa.schema([pa.field("c", arr.type, nullable=bool(ln))]))
            ch = PU.column_chunks(PU.write_parquet(table, dictionary=False, v2=(cname == "snappy"), page_size=2048, compression=cname))[0][0]
            seeds.append((ch, ln, en, ot))
    seen = set()
    for it in range(3000):
        ch, ln, en, ot = seeds[it % len(seeds)]
        b = bytearray(ch["chunk"])
        k = it % 3
        if k == 0:
            for _ in range(int(rng.integers(1, 4))):
                b[int(rng.integers(0, len(b)))] ^= 1 << int(rng.integers(0, 8))
        elif k == 1:
            b = b[: int(rng.integers(0, len(b)))]
        else:
            i = int(rng.integers(0, len(b)))
            b[i:i + 6] = bytes(rng.integers(0, 256, 6).astype(np.uint8))
        m = dict(ch)
        m["chunk"] = bytes(b)
        rc, info = _open_list(m, ln, en, ot)
        assert rc in (T.OK, T.ERR_INVALID, T.ERR_UNSUPPORTED), rc
        if rc == T.OK:
            assert 0 <= info.image_bytes <= max(1 << 30, 1024 * len(b)) + 16 and info.num_values >= 0
        seen.add(rc)
    assert T.OK in seen and T.ERR_INVALID in seen


def test_oracle_reads_the_list_column_of_the_reference_held_files():
    """The List<Int64> column the reference keeps in tests/data/parquet/multi_page/multi_page_{1..4}.parquet (col_arr = [[1], [1, 2]] * num_row,
    gen.py:12 — written "to test multi pages in a column chunk for list type"; SNAPPY, dictionary-encoded v1 pages of 128 bytes): the
    List decode's pin on a reference-held fixture (rounds 4-5 pinned it on pyarrow's reading only)."""
    from tests import oracle_lib
    from tests import parquet_ref as PR
    orc = oracle_lib.load()
    orc.orc_pq_decode_list_codec.restype = C.c_int

    def decode_list(ch, ln, en, ot):
        ent = ch["num_values"]
        chunk = np.frombuffer(ch["chunk"], dtype=np.uint8)
        es = PU.ESIZE[ot]
        offs, lval = np.zeros(ent + 2, np.uint64), np.zeros(ent + 1, np.uint8)
        vals, ev = np.zeros(max(ent, 1) * es + 16, np.uint8), np.zeros(ent + 1, np.uint8)
        rows, elems = C.c_int64(), C.c_int64()
        rc = orc.orc_pq_decode_list_codec(chunk.ctypes.data_as(C.c_void_p), C.c_int64(len(chunk)), ch["codec"], ch["physical"], ch["type_length"], ln, en, ot,
                                          C.c_int64(ent), offs.ctypes.data_as(C.c_void_p), lval.ctypes.data_as(C.c_void_p), vals.ctypes.data_as(C.c_void_p),
                                          ev.ctypes.data_as(C.c_void_p), C.byref(rows), C.byref(elems))
        assert rc == 0 and int(offs[rows.value]) == elems.value
        m = elems.value
        py = PU.decoded_to_python(vals.tobytes(), ev[:m].astype(bool), ot, m, chunk)
        return [None if (ln and not lval[r]) else py[int(offs[r]):int(offs[r + 1])] for r in range(rows.value)]
    assert PR.check_lists(decode_list) == 4
