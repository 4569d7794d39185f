"""GPU parity of the scan-side decode: dbhip_pq_chunk_open / _decode through the C-ABI against what
pyarrow reads from the same file (bit-exact, NULL slots zero) and against the CPU oracle, for the writer shapes of
storages/common/blocks/src/parquet_rs.rs:91-160 and the edge cases (empty / one row / all NULL / page-boundary runs /
dictionary overflow / 12- vs 13-byte strings / NaN bit patterns)."""
import json
import os

import numpy as np
import pytest

from databend_amd import _lib as T
from tests import parquet_cases as PC
from tests import parquet_util as PU

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "parquet")


def unpack(bits, n):
    return np.unpackbits(np.frombuffer(bits, dtype=np.uint8), bitorder="little")[:n].astype(bool)


def gpu_decode(gpu, ch, out_type, device=False):
    """device=True: dbhip_pq_chunk_open_device / _decode_device (the payload is decompressed and walked on the GPU)"""
    pc = gpu.ParquetChunk(ch["chunk"], ch["physical"], out_type, ch["type_length"], ch["max_def"], ch.get("max_rep", 0), ch.get("codec", 0),
                          device=device)
    i = pc.info
    col = pc.decode()
    if device:
        assert i.num_nulls in (-1, pc.nulls)
        i.num_nulls = pc.nulls           # (v1 pages of a nullable column: only known once the levels were walked)
    n = i.num_values
    valid = unpack(col.validity.to_numpy(np.uint8, i.validity_bytes).tobytes(), n) if i.has_validity else np.ones(n, dtype=bool)
    raw = col.data.to_numpy(np.uint8, i.out_bytes).tobytes()
    py = PU.decoded_to_python(raw, valid, out_type, n, pc.device_image() if device else pc.image())   # (long String views point into the chunk / the image)
    if i.has_validity and n % 64:
        tail = unpack(col.validity.to_numpy(np.uint8, i.validity_bytes).tobytes(), i.validity_bytes * 8)[n:]
        assert not tail.any()            # padding bits of the bitmap are clear
    # decoding twice from the resident chunk gives the same column (the handle is reusable)
    col2 = pc.decode()
    assert col2.data.to_numpy(np.uint8, i.out_bytes).tobytes() == raw
    pc.close()
    return py, valid, i


@pytest.mark.parametrize("vi", range(len(PC.VARIANTS)))
def test_chunk_decode_matches_pyarrow_and_oracle(gpu, vi):
    import pyarrow as pa
    for name, arr, out_type, wkw in PC.make_cases(seed=vi):
        kw = dict(PC.VARIANTS[vi])
        kw.update(wkw)
        fb = PU.write_parquet(pa.table({"c": arr}), **kw)
        chunks, back = PU.column_chunks(fb)
        ch = chunks[0]
        exp, exp_valid = PU.expected_of(back.column(0), out_type)
        got, valid, info = gpu_decode(gpu, ch, out_type)
        assert info.num_values == len(exp) and info.num_nulls == int((~exp_valid).sum()), name
        assert np.array_equal(valid, exp_valid), name
        assert got == exp, name
        o_got, o_valid, _, _, rc = PU.oracle_decode(ch, out_type)
        assert rc == 0 and o_got == got, name


@pytest.mark.parametrize("cname", ["zstd", "lz4", "snappy"])
@pytest.mark.parametrize("vi", [0, 1, 4])
def test_compressed_chunks_decode_like_pyarrow(gpu, cname, vi):
    """TableCompression Zstd (the reference's default) / LZ4 / Snappy: pages are decompressed on the host inside open(), the
    device decodes the decompressed image; values, validity and String views (which point into the image) equal pyarrow's."""
    import pyarrow as pa
    for name, arr, out_type, wkw in PC.make_cases(seed=vi):
        kw = dict(PC.VARIANTS[vi])
        kw.update(wkw)
        chunks, back = PU.column_chunks(PU.write_parquet(pa.table({"c": arr}), compression=cname, **kw))
        ch = chunks[0]
        exp, exp_valid = PU.expected_of(back.column(0), out_type)
        got, valid, info = gpu_decode(gpu, ch, out_type)
        assert info.num_values == len(exp) and info.num_nulls == int((~exp_valid).sum()), name
        assert info.image_bytes > 0 or len(exp) == 0, name
        assert np.array_equal(valid, exp_valid) and got == exp, name


def test_golden_fixtures_through_the_c_abi(gpu):
    names = sorted(f[:-5] for f in os.listdir(GOLD) if f.endswith(".json"))
    assert len(names) >= 10
    for nm in names:
        meta = json.load(open(os.path.join(GOLD, nm + ".json")))
        chunk = open(os.path.join(GOLD, nm + ".bin"), "rb").read()
        ch = dict(chunk=chunk, physical=meta["physical"], type_length=meta["type_length"], max_def=meta["max_def"])
        got, valid, info = gpu_decode(gpu, ch, meta["out_type"])
        assert info.num_values == meta["rows"] and info.num_nulls == meta["nulls"], nm
        norm = [None if v is None else (v.hex() if isinstance(v, bytes) else (int(v) if not isinstance(v, bool) else v)) for v in got]
        assert norm == meta["values"], nm


def test_open_reports_what_it_does_not_decode(gpu):
    import io
    import pyarrow as pa
    import pyarrow.parquet as pq
    t = pa.table({"c": pa.array(list(range(5000)), pa.int64()), "l": pa.array([[1, 2]] * 5000, pa.list_(pa.int32()))})

    def code(ch, out_type=T.T_I64, **over):
        d = dict(ch)
        d.update(over)
        with pytest.raises(T.DbhipError) as e:
            gpu.ParquetChunk(d["chunk"], d["physical"], out_type, d["type_length"], d["max_def"], d["max_rep"], d["codec"])
        return e.value.code

    buf = io.BytesIO()
    pq.write_table(t, buf, compression="gzip", use_dictionary=False)
    chunks, _ = PU.column_chunks(buf.getvalue())
    assert chunks[0]["codec"] == 2
    assert code(chunks[0]) == T.ERR_UNSUPPORTED                       # a codec the library does not decompress (GZIP)
    assert code(chunks[0], codec=0) == T.ERR_UNSUPPORTED              # compressed pages in a chunk declared UNCOMPRESSED
    buf = io.BytesIO()
    pq.write_table(t, buf, compression="none", use_dictionary=False, column_encoding={"c": "DELTA_BINARY_PACKED", "l.list.element": "PLAIN"})
    chunks, _ = PU.column_chunks(buf.getvalue())
    assert code(chunks[0]) == T.ERR_UNSUPPORTED                       # encoding outside the reference writer's repertoire
    assert code(chunks[1], T.T_I32) == T.ERR_UNSUPPORTED              # nested column (repetition levels)
    fb = PU.write_parquet(t.select(["c"]), dictionary=True)
    chunks, _ = PU.column_chunks(fb)
    assert code(chunks[0], T.T_F64) == T.ERR_UNSUPPORTED              # INT64 cannot become Float64
    assert code(chunks[0], chunk=chunks[0]["chunk"][: len(chunks[0]["chunk"]) // 2]) == T.ERR_INVALID   # truncated
    assert code(chunks[0], chunk=b"\xff" * 64) == T.ERR_INVALID       # not a page header


def test_decoded_columns_feed_the_operators(gpu, oracle):
    """lineitem-shaped chunk -> decode -> the existing operators, no host round trip of the data: l_shipdate <= cutoff on the
    decoded Date column, and a group-by on the decoded (dictionary-encoded, nullable) l_returnflag views."""
    import pyarrow as pa
    n = 300_000
    rng = np.random.default_rng(9)
    ship = rng.integers(8000, 10600, n).astype(np.int32)
    flag = np.array(["A", "R", "N"])[rng.integers(0, 3, n)]
    fmask = rng.random(n) < 0.02
    qty = rng.integers(1, 51, n) * 100
    t = pa.table({"ship": pa.array(ship, pa.int32()).cast(pa.date32()), "flag": pa.array(flag, pa.string(), mask=fmask),
                  "qty": pa.array(qty, pa.int64())})
    chunks, _ = PU.column_chunks(PU.write_parquet(t, dictionary=True))
    cols = []
    for ch, ot in zip(chunks, (T.T_DATE, T.T_STRING, T.T_DEC64)):
        pc = gpu.ParquetChunk(ch["chunk"], ch["physical"], ot, ch["type_length"], ch["max_def"], precision=15, scale=2)
        cols.append(pc.decode())
    d_ship, d_flag, d_qty = cols
    pred = gpu.cmp(T.CMP_LTE, d_ship, gpu.Column.scalar(10471, T.T_DATE), n)
    assert gpu.bitmap_count(pred, n) == int((ship <= 10471).sum())
    g = gpu.GroupBy([T.T_STRING], [(T.AGG_SUM, T.T_DEC64, 15, 2, 0), (T.AGG_COUNT, 0, 0, 0, 0)], key_nullable=[1])
    g.add_block([d_flag], [d_qty, None], n)
    got = {r[0]: (r[1], r[2]) for r in g.result()}
    exp = {}
    for f, m, q in zip(flag.tolist(), fmask.tolist(), qty.tolist()):
        k = None if m else f.encode()
        s = exp.setdefault(k, [0, 0])
        s[0] += q
        s[1] += 1
    assert {k: tuple(v) for k, v in exp.items()} == got


def test_full_size_roundtrip_20m_rows(gpu):
    """BASELINE-sized chunk: 20 M Int64 values (Decimal(15,2) of lineitem) with 3 % NULLs, dictionary off (PLAIN, v1) and a
    low-cardinality dictionary column; decode == the array that was written (size-independent property: write -> decode is
    the identity), NULL slots zero."""
    import pyarrow as pa
    n = 20_000_000
    rng = np.random.default_rng(4)
    price = rng.integers(90000, 10494951, n)
    mask = rng.random(n) < 0.03
    disc = rng.integers(0, 11, n)
    for arr, dictionary, src, m in ((pa.array(price, pa.int64(), mask=mask), False, price, mask), (pa.array(disc, pa.int64()), True, disc, None)):
        chunks, _ = PU.column_chunks(PU.write_parquet(pa.table({"c": arr}), dictionary=dictionary))
        ch = chunks[0]
        pc = gpu.ParquetChunk(ch["chunk"], ch["physical"], T.T_DEC64, ch["type_length"], ch["max_def"], precision=15, scale=2)
        col = pc.decode()
        got = col.data.to_numpy(np.int64, n)
        if m is not None:
            valid = unpack(col.validity.to_numpy(np.uint8, pc.info.validity_bytes).tobytes(), n)
            assert np.array_equal(valid, ~m)
            assert np.array_equal(got, np.where(m, 0, src))
            assert pc.info.num_nulls == int(m.sum())
        else:
            assert np.array_equal(got, src)
        pc.close()


def test_mutated_chunks_never_fault(gpu):
    """Corrupt chunks (bit flips, truncation, overwritten bytes): open() either rejects them (INVALID / UNSUPPORTED) or
    plans a decode that stays inside the chunk, the dictionary and the output buffers — every offset the device uses is
    validated on the host, dictionary indices are clamped. Accepted mutants are decoded; the values may be garbage, the call
    must succeed."""
    import pyarrow as pa
    rng = np.random.default_rng(12)
    seeds = []
    for vi in (0, 1, 4, 6):
        for name, arr, ot, wkw in PC.make_cases(seed=vi):
            kw = dict(PC.VARIANTS[vi])
            kw.update(wkw)
            chunks, _ = PU.column_chunks(PU.write_parquet(pa.table({"c": arr.slice(0, 2500)}), **kw))
            if len(chunks[0]["chunk"]):
                seeds.append((chunks[0], ot))
    opened = rejected = 0
    for it in range(1500):
        ch, ot = seeds[it % len(seeds)]
        b = bytearray(ch["chunk"])
        k = int(rng.integers(0, 3))
        if k == 0:
            for _ in range(int(rng.integers(1, 4))):
                b[int(rng.integers(0, len(b)))] ^= 1 << int(rng.integers(0, 8))
        elif k == 1:
            b = b[: int(rng.integers(1, len(b)))]
        else:
            i = int(rng.integers(0, len(b)))
            b[i:i + 4] = bytes(rng.integers(0, 256, 4).astype(np.uint8))
        try:
            pc = gpu.ParquetChunk(bytes(b), ch["physical"], ot, ch["type_length"], ch["max_def"])
        except T.DbhipError as e:
            assert e.code in (T.ERR_INVALID, T.ERR_UNSUPPORTED)
            rejected += 1
            continue
        col = pc.decode()
        col.data.to_numpy(np.uint8, pc.info.out_bytes)     # forces completion: a device fault would surface here
        pc.close()
        opened += 1
    assert opened > 100 and rejected > 100


def test_reference_held_parquet_files_decode_to_what_the_reference_tests_print(gpu):
    """The scan-side decode PINNED ON THE REFERENCE (not on pyarrow's reader): column chunks of the Parquet files under the reference's
    tests/data — written by parquet-cpp, parquet-mr and parquet-rs 58.1.0, the crate the reference links — through dbhip_pq_chunk_open /
    _decode give exactly what the reference's sqllogictests print for those files (select_parquet.test:6-16,69-72,
    parquet_field_types.test:214-219, timestamp.test:1-36, on_time.test:1-12,54-61): SNAPPY v1 pages, dictionary-encoded INT32 / INT64 /
    FLOAT / DOUBLE / BYTE_ARRAY, PLAIN Booleans, an uncompressed BYTE_ARRAY with a value longer than 12 bytes, 8-row-group files."""
    from tests import parquet_ref as PR

    def decode(ch, out_type):
        py, valid, info = gpu_decode(gpu, ch, out_type)
        assert info.num_values == ch["num_values"]
        return py, valid
    assert PR.check_all(decode) == 21
