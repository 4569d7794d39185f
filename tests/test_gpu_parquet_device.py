"""GPU parity of the DEVICE mode of the scan-side decode: dbhip_pq_chunk_open_device / _decode_device. The host reads the
thrift page headers only; Snappy / LZ4 page decompression, the run headers of the RLE / bit-packed hybrid streams, the BYTE_ARRAY length
chain and DELTA_BINARY_PACKED blocks are walked on the GPU. Same fixtures as the host-planned mode (tests/test_gpu_parquet.py): pyarrow's
reader, the CPU oracle, the committed golden chunks and the Parquet files the reference keeps under tests/data with the values its own
sqllogictests print for them."""
import ctypes as C
import io
import json
import os
import time

import numpy as np
import pytest

from databend_amd import _lib as T
from tests import parquet_cases as PC
from tests import parquet_util as PU
from tests.test_gpu_parquet import GOLD, gpu_decode, unpack

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("vi", range(len(PC.VARIANTS)))
def test_device_mode_matches_pyarrow_and_oracle(gpu, vi):
    import pyarrow as pa
    for name, arr, out_type, wkw in PC.make_cases(seed=vi):
        kw = dict(PC.VARIANTS[vi])
        kw.update(wkw)
        chunks, back = PU.column_chunks(PU.write_parquet(pa.table({"c": arr}), **kw))
        ch = chunks[0]
        exp, exp_valid = PU.expected_of(back.column(0), out_type)
        got, valid, info = gpu_decode(gpu, ch, out_type, device=True)
        assert info.num_values == len(exp) and info.num_nulls == int((~exp_valid).sum()), name
        assert np.array_equal(valid, exp_valid), name
        assert got == exp, name
        o_got, _, _, _, rc = PU.oracle_decode(ch, out_type)
        assert rc == 0 and o_got == got, name


@pytest.mark.parametrize("cname", ["zstd", "lz4", "snappy"])
@pytest.mark.parametrize("vi", [0, 1, 2, 4, 5])
def test_device_side_decompression_matches_pyarrow(gpu, cname, vi):
    """TableCompression Zstd (the default) / LZ4 / Snappy (table_compression.rs:27-58): one wave per page decompresses into the HBM image;
    values, validity and String views (which point into that image) equal pyarrow's, and the image equals the page payloads of the
    uncompressed twin."""
    import pyarrow as pa
    for name, arr, out_type, wkw in PC.make_cases(seed=vi):
        kw = dict(PC.VARIANTS[vi])
        kw.update(wkw)
        chunks, back = PU.column_chunks(PU.write_parquet(pa.table({"c": arr}), compression=cname, **kw))
        ch = chunks[0]
        exp, exp_valid = PU.expected_of(back.column(0), out_type)
        got, valid, info = gpu_decode(gpu, ch, out_type, device=True)
        assert info.num_values == len(exp) and info.num_nulls == int((~exp_valid).sum()), name
        assert info.image_bytes > 0 or len(exp) == 0, name
        assert np.array_equal(valid, exp_valid) and got == exp, name


def _zstd():
    import ctypes
    Z = ctypes.CDLL("libzstd.so.1")
    Z.ZSTD_compress.restype = ctypes.c_size_t
    Z.ZSTD_compressBound.restype = ctypes.c_size_t
    Z.ZSTD_compressBound.argtypes = [ctypes.c_size_t]

    def compress(data, level):
        cap = Z.ZSTD_compressBound(len(data))
        buf = ctypes.create_string_buffer(cap)
        n = Z.ZSTD_compress(buf, ctypes.c_size_t(cap), data, ctypes.c_size_t(len(data)), ctypes.c_int(level))
        assert n <= cap
        return buf.raw[:n]
    return compress


def _decode_frames_as_page(gpu, frames, plain):
    """a payload compressed elsewhere, wrapped as the one v1 PLAIN INT64 page of a required column -> (values, decompressed image)"""
    n = len(plain) // 8
    chunk = PU.raw_page_chunk(frames, len(plain), n)
    pc = gpu.ParquetChunk(chunk, PU.PHYS["INT64"], T.T_I64, 0, 0, 0, 6, device=True)
    col = pc.decode()
    vals = col.data.to_numpy(np.int64, n)
    img = pc.device_image()
    pc.close()
    return vals, bytes(img[:len(plain)])


def _odd_snappy(rng, total):
    """(stream, plaintext): a legal Snappy raw stream no encoder writes — literal headers longer than they need to be (1..4 length bytes in
    front of a few bytes), runs of one-byte literals with five-byte headers, copies of every kind (1-, 2- and 4-byte offsets), copies that
    overlap their own output (offset < length), offsets far behind the decompressor's LDS ring"""
    s, out = bytearray(), bytearray()

    def literal(data):
        n = len(data)
        forms = [f for f, lim in ((0, 60), (1, 256), (2, 65536), (3, 1 << 24), (4, 1 << 32)) if n <= lim]
        f = int(rng.choice(forms))
        if f == 0:
            s.append((n - 1) << 2)
        else:
            s.append((59 + f) << 2)
            s.extend((n - 1).to_bytes(f, "little"))
        s.extend(data)
        out.extend(data)

    def copy(off, ln):
        kinds = [3] + ([2] if off < 65536 else []) + ([1] if 4 <= ln <= 11 and off < 2048 else [])
        k = int(rng.choice(kinds))
        if k == 1:
            s.extend([1 | ((ln - 4) << 2) | ((off >> 8) << 5), off & 0xFF])
        elif k == 2:
            s.append(2 | ((ln - 1) << 2))
            s.extend(off.to_bytes(2, "little"))
        else:
            s.append(3 | ((ln - 1) << 2))
            s.extend(off.to_bytes(4, "little"))
        for _ in range(ln):
            out.append(out[-off])

    while len(out) < total:
        r = rng.random()
        if len(out) == 0 or r < 0.25:
            literal(rng.integers(0, 256, int(rng.choice([1, 1, 2, 3, 4, 5, 17, 60, 61, 100, 300, 5000])), dtype=np.uint8).tobytes())
        elif r < 0.28:
            for _ in range(int(rng.integers(70, 200))):      # one-byte literals, five header bytes each
                s.extend([63 << 2, 0, 0, 0, 0])
                b = int(rng.integers(0, 256))
                s.append(b)
                out.append(b)
        else:
            far = min(len(out), 70000)
            off = int(rng.choice([1, 2, 3, 7, int(rng.integers(1, min(len(out), 64) + 1)), int(rng.integers(1, far + 1))]))
            copy(min(off, len(out)), int(rng.integers(1, 65)))
    if len(out) % 8:
        literal(bytes(8 - len(out) % 8))
    n, pre = len(out), bytearray()
    while True:
        pre.append((n & 0x7F) | (0x80 if n > 0x7F else 0))
        n >>= 7
        if not n:
            break
    return bytes(pre + s), bytes(out)


def _odd_lz4(rng, total):
    """(block, plaintext): an LZ4 block with every length form (15 exactly: one extension byte 0; 15 + 255 k: a chain of 255s closed by 0;
    long literals and matches), overlapping matches and offsets up to 65535"""
    s, out = bytearray(), bytearray()

    def ext(v):
        while v >= 255:
            s.append(255)
            v -= 255
        s.append(v)

    def seq(lit, ml, off):
        s.append((min(lit, 15) << 4) | (min(ml - 4, 15) if ml else 0))
        if lit >= 15:
            ext(lit - 15)
        data = rng.integers(0, 256, lit, dtype=np.uint8).tobytes()
        s.extend(data)
        out.extend(data)
        if ml:
            s.extend(off.to_bytes(2, "little"))
            if ml - 4 >= 15:
                ext(ml - 4 - 15)
            for _ in range(ml):
                out.append(out[-off])

    while len(out) < total:
        lit = int(rng.choice([0, 0, 1, 2, 3, 4, 5, 6, 14, 15, 16, 15 + 255, 15 + 255 + 3, 15 + 510, 700]))
        if len(out) == 0 and lit == 0:
            lit = 3
        ml = int(rng.choice([4, 4, 5, 8, 18, 19, 20, 19 + 255, 19 + 255 + 9, 19 + 510, 1000]))
        far = min(len(out) + lit, 65535)
        off = int(rng.choice([1, 2, 3, 7, int(rng.integers(1, min(far, 64) + 1)), int(rng.integers(1, far + 1))]))
        seq(lit, ml, min(off, far))
    tail = 16 + (-(len(out) + 16)) % 8
    s.append(min(tail, 15) << 4)              # the last sequence: literals only
    if tail >= 15:
        ext(tail - 15)
    s.extend(bytes(tail))
    out.extend(bytes(tail))
    return bytes(s), bytes(out)


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_legal_streams_no_encoder_writes(gpu, seed):
    """Snappy (format_description.txt) and LZ4 (lz4_Block_format.md) streams built element by element from the format documents — forms that
    are legal but that the encoders behind the parquet crate never emit — decompress on the GPU to the plaintext they were built from; the
    system's own decoders (through pyarrow) agree that the streams are legal."""
    import pyarrow as pa
    rng = np.random.default_rng(seed)
    for codec, name, make in ((1, "snappy", _odd_snappy), (7, "lz4_raw", _odd_lz4)):
        for total in (700, 40_000, 400_000):
            z, plain = make(rng, total)
            assert pa.Codec(name).decompress(z, len(plain)).to_pybytes() == plain, (name, total)
            chunk = PU.raw_page_chunk(z, len(plain), len(plain) // 8)
            pc = gpu.ParquetChunk(chunk, PU.PHYS["INT64"], T.T_I64, 0, 0, 0, codec, device=True)
            pc.decode()
            img = bytes(pc.device_image()[:len(plain)])
            pc.close()
            assert img == plain, (name, total, seed)


def test_zstd_frames_the_reference_holds(gpu):
    """Known-answer vectors for the ZSTD path: the frames under the reference's tests/data (tests/golden/zstd_ref; ontime_200.csv.zst with the
    plaintext the reference keeps beside it, a 2.9 MB wasm module compressed at a high level: many blocks, treeless literals, repeat offsets,
    references far beyond the 8 KiB ring) decompressed on the GPU."""
    import hashlib
    gold = os.path.join(os.path.dirname(GOLD), "zstd_ref")
    index = json.load(open(os.path.join(gold, "index.json")))
    for name, meta in index.items():
        z = open(os.path.join(gold, name + ".zst"), "rb").read()
        chunk = PU.raw_page_chunk(z, meta["decoded_bytes"], meta["decoded_bytes"] // 8)
        pc = gpu.ParquetChunk(chunk, PU.PHYS["INT64"], T.T_I64, 0, 0, 0, 6, device=True)
        pc.decode()
        img = bytes(pc.device_image()[:meta["decoded_bytes"]])
        pc.close()
        assert hashlib.sha256(img).hexdigest() == meta["sha256"], name


@pytest.mark.parametrize("level", [1, 3, 9, 19, -3])
def test_zstd_levels_and_block_shapes(gpu, level):
    """What libzstd (the library behind the reference's zstd crate) emits at several levels over the shapes pages take: raw / RLE /
    compressed blocks, single- and four-stream Huffman literals, predefined / RLE / FSE / repeat sequence tables, several frames in one
    page; the GPU's bytes are the input's."""
    compress = _zstd()
    rng = np.random.default_rng(100 + level)
    n = 40_000
    cases = {
        "random": rng.integers(0, 256, n * 8, dtype=np.uint8).tobytes(),
        "zeros": bytes(n * 8),
        "prices": rng.integers(90000, 10494951, n).astype(np.int64).tobytes(),
        "skewed": np.minimum(rng.geometric(0.3, n * 8), 255).astype(np.uint8).tobytes(),
        "packed2bit": rng.integers(0, 4, n * 8, dtype=np.uint8).tobytes(),
        "period": np.tile(rng.integers(0, 2**40, 3001), n // 3001 + 1)[:n].astype(np.int64).tobytes(),
        "runs": np.repeat(rng.integers(0, 1000, n // 500 + 1), 500)[:n].astype(np.int64).tobytes(),
        "tiny": b"abcdefgh" * 3,
        "text": (b"lineitem|orders|DELIVER IN PERSON|TRUCK|furiously final packages|1996-03-13|" * 4000)[: n * 8],
    }
    for name, plain in cases.items():
        vals, img = _decode_frames_as_page(gpu, compress(plain, level), plain)
        assert img == plain, (name, level)
        assert np.array_equal(vals, np.frombuffer(plain, np.int64)), name
    # two frames in one page (the format allows concatenation; back-references do not cross the frame boundary)
    a, b = cases["prices"][:100_000], cases["text"][:160_000]
    vals, img = _decode_frames_as_page(gpu, compress(a, level) + compress(b, level), a + b)
    assert img == a + b


def test_zstd_dictionary_frames_and_nested_chunks_are_refused(gpu):
    import pyarrow as pa
    compress = _zstd()
    plain = bytes(range(256)) * 64
    z = bytearray(compress(plain, 3))
    # Frame_Header_Descriptor: set Dictionary_ID_flag = 1 and insert a non-zero one-byte dictionary id behind the (optional) window byte
    single = (z[4] >> 5) & 1
    z[4] |= 1
    z.insert(5 if single else 6, 7)
    chunk = PU.raw_page_chunk(bytes(z), len(plain), len(plain) // 8)
    pc = gpu.ParquetChunk(chunk, PU.PHYS["INT64"], T.T_I64, 0, 0, 0, 6, device=True)
    with pytest.raises(T.DbhipError) as e:
        pc.decode()
    assert e.value.code == T.ERR_UNSUPPORTED
    pc.close()
    # a wrong uncompressed_page_size in the page header is a malformed chunk, not an overrun
    z = compress(plain, 3)
    for wrong in (len(plain) - 8, len(plain) + 8):
        pc = gpu.ParquetChunk(PU.raw_page_chunk(z, wrong, wrong // 8), PU.PHYS["INT64"], T.T_I64, 0, 0, 0, 6, device=True)
        with pytest.raises(T.DbhipError) as e:
            pc.decode()
        assert e.value.code == T.ERR_INVALID
        pc.close()
    t = pa.table({"c": pa.array(list(range(5000)), pa.int64())})
    # a handle of one mode is refused by the other mode's decode
    chunks, _ = PU.column_chunks(PU.write_parquet(t))
    ch = chunks[0]
    pc = gpu.ParquetChunk(ch["chunk"], ch["physical"], T.T_I64, ch["type_length"], ch["max_def"], device=True)
    pc.device = False
    with pytest.raises(T.DbhipError):
        pc.decode()
    pc.close()


@pytest.mark.parametrize("cname", ["none", "snappy", "lz4", "zstd"])
def test_delta_binary_packed(gpu, cname):
    """DELTA_BINARY_PACKED INT32 / INT64 (not in the reference writer's repertoire, but in files it reads): several pages, NULLs, runs of
    equal deltas (bit width 0), full-width deltas, against pyarrow and the oracle's statement of Encodings.md."""
    import pyarrow as pa
    import pyarrow.parquet as pq
    rng = np.random.default_rng(5)
    for typ, ot, hi in ((pa.int64(), T.T_I64, 2**62), (pa.int32(), T.T_I32, 2**30), (pa.int64(), T.T_DEC128, 2**40), (pa.int32(), T.T_I64, 2**20)):
        for n, frac in ((1, 0.0), (129, 0.0), (70_000, 0.07), (5000, 1.0)):
            a = rng.integers(-hi, hi, n)
            a[n // 4: n // 2] = np.arange(n // 2 - n // 4) * 3 + 7      # constant delta
            a[n // 2: n // 2 + n // 8] = 42                             # zero delta
            mask = rng.random(n) < frac
            t = pa.table({"c": pa.array(a, typ, mask=mask)})
            buf = io.BytesIO()
            pq.write_table(t, buf, compression=cname, use_dictionary=False, column_encoding={"c": "DELTA_BINARY_PACKED"}, data_page_size=20_000,
                           write_statistics=False, row_group_size=n)
            chunks, back = PU.column_chunks(buf.getvalue())
            ch = chunks[0]
            assert "DELTA_BINARY_PACKED" in ch["encodings"]
            exp, exp_valid = PU.expected_of(back.column(0), ot)
            got, valid, info = gpu_decode(gpu, ch, ot, device=True)
            assert np.array_equal(valid, exp_valid) and got == exp, (str(typ), n)
            if cname == "none":
                o_got, _, _, _, rc = PU.oracle_decode(ch, ot)
                assert rc == 0 and o_got == got


@pytest.mark.parametrize("cname", ["snappy", "lz4", "zstd"])
def test_large_compressible_pages(gpu, cname):
    """Pages far larger than the 64 KiB LDS window, with the back-reference shapes real data produces: long runs (overlapping matches at
    distance 1..8), repeated rows at 16-bit distances, incompressible stretches (long literals), Booleans, repetitive strings longer than 12
    bytes (views into the decompressed image)."""
    import pyarrow as pa
    rng = np.random.default_rng(8)
    n = 600_000
    runs = np.repeat(rng.integers(0, 1000, n // 500 + 1), 500)[:n].astype(np.int64)
    mixed = np.where(rng.random(n) < 0.5, rng.integers(0, 2**60, n), 7).astype(np.int64)
    period = np.tile(rng.integers(0, 2**40, 3001), n // 3001 + 1)[:n].astype(np.int64)
    words = np.array([b"lineitem-comment-%06d-abcdefghijklmnop" % i for i in range(997)], dtype=object)
    strs = words[rng.integers(0, 997, 200_000)]
    cases = [("runs", pa.array(runs, pa.int64()), T.T_I64), ("mixed", pa.array(mixed, pa.int64(), mask=rng.random(n) < 0.02), T.T_I64),
             ("period", pa.array(period, pa.int64()), T.T_I64), ("bool", pa.array(rng.random(n) < 0.01, pa.bool_()), T.T_BOOL),
             ("strings", pa.array(list(strs), pa.binary()), T.T_STRING), ("f64", pa.array(rng.random(n // 4)), T.T_F64)]
    for name, arr, ot in cases:
        for dictionary in (False, True):
            chunks, back = PU.column_chunks(PU.write_parquet(pa.table({"c": arr}), dictionary=dictionary, compression=cname, page_size=1 << 20))
            ch = chunks[0]
            exp, exp_valid = PU.expected_of(back.column(0), ot)
            got, valid, info = gpu_decode(gpu, ch, ot, device=True)
            assert np.array_equal(valid, exp_valid), name
            assert got == exp, (name, dictionary)


def test_golden_and_reference_held_chunks_in_device_mode(gpu):
    from tests import parquet_ref as PR
    names = sorted(f[:-5] for f in os.listdir(GOLD) if f.endswith(".json"))
    for nm in names:
        meta = json.load(open(os.path.join(GOLD, nm + ".json")))
        chunk = open(os.path.join(GOLD, nm + ".bin"), "rb").read()
        ch = dict(chunk=chunk, physical=meta["physical"], type_length=meta["type_length"], max_def=meta["max_def"])
        got, valid, info = gpu_decode(gpu, ch, meta["out_type"], device=True)
        assert info.num_values == meta["rows"] and info.num_nulls == meta["nulls"], nm
        norm = [None if v is None else (v.hex() if isinstance(v, bytes) else (int(v) if not isinstance(v, bool) else v)) for v in got]
        assert norm == meta["values"], nm

    # the Parquet files under the reference's tests/data (parquet-cpp, parquet-mr and parquet-rs 58.1.0 writers; SNAPPY v1 pages) against
    # what the reference's sqllogictests print for them
    def decode(ch, out_type):
        py, valid, info = gpu_decode(gpu, ch, out_type, device=True)
        assert info.num_values == ch["num_values"]
        return py, valid
    assert PR.check_all(decode) == 21
    assert PR.check_tuple(decode) == 3     # (round 6) the members of a NOT NULL Tuple column are flat leaves


def test_mutated_chunks_never_fault_in_device_mode(gpu):
    """Nothing but the page table is validated on the host here, so the kernels themselves must stay inside the page, the window, the
    dictionary and the output for ANY bytes: bit flips, truncation and overwritten words in plain, Snappy and LZ4 chunks. A mutant is either
    refused (open: header damage; decode: a device-side check) or decodes to (possibly garbage) values; the device must stay healthy."""
    import pyarrow as pa
    rng = np.random.default_rng(13)
    seeds = []
    for vi, cname in ((0, "none"), (1, "none"), (4, "snappy"), (1, "snappy"), (0, "lz4"), (5, "lz4"), (0, "zstd"), (4, "zstd"), (5, "zstd")):
        for name, arr, ot, wkw in PC.make_cases(seed=vi):
            kw = dict(PC.VARIANTS[vi])
            kw.update(wkw)
            chunks, _ = PU.column_chunks(PU.write_parquet(pa.table({"c": arr.slice(0, 2500)}), compression=cname, **kw))
            if len(chunks[0]["chunk"]):
                seeds.append((chunks[0], ot))
    opened = rejected = failed = 0
    for it in range(2200):
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
            pc = gpu.ParquetChunk(bytes(b), ch["physical"], ot, ch["type_length"], ch["max_def"], 0, ch["codec"], device=True)
        except T.DbhipError as e:
            assert e.code in (T.ERR_INVALID, T.ERR_UNSUPPORTED)
            rejected += 1
            continue
        try:
            col = pc.decode()
            col.data.to_numpy(np.uint8, pc.info.out_bytes)     # forces completion: a device fault would surface here
            opened += 1
        except T.DbhipError as e:
            assert e.code in (T.ERR_INVALID, T.ERR_UNSUPPORTED)
            failed += 1
        pc.close()
    assert opened > 100 and rejected > 100 and failed > 50
    # the device is still healthy: an intact chunk decodes right after
    ch, ot = seeds[0]
    gpu_decode(gpu, ch, ot, device=True)


def test_many_chunks_one_launch_set(gpu):
    """dbhip_pq_chunks_decode_device: the column chunks of several blocks (all four codecs, every encoding of make_cases, NULLs, strings,
    Booleans) decoded by one launch set equal their one-by-one decodes; a corrupt chunk in the batch gets its own status and the others
    still decode."""
    import pyarrow as pa
    opened, singles = [], []
    for vi, cname in ((0, "zstd"), (1, "snappy"), (2, "none"), (4, "lz4"), (5, "zstd")):
        for name, arr, ot, wkw in PC.make_cases(seed=vi):
            kw = dict(PC.VARIANTS[vi])
            kw.update(wkw)
            chunks, back = PU.column_chunks(PU.write_parquet(pa.table({"c": arr}), compression=cname, **kw))
            ch = chunks[0]
            if not len(ch["chunk"]):
                continue
            exp, exp_valid = PU.expected_of(back.column(0), ot)
            opened.append((gpu.ParquetChunk(ch["chunk"], ch["physical"], ot, ch["type_length"], ch["max_def"], 0, ch["codec"], device=True), ot, exp,
                           exp_valid, name))
    assert len(opened) > 40
    cols = gpu.ParquetChunk.decode_many([o[0] for o in opened])
    for (pc, ot, exp, exp_valid, name), col in zip(opened, cols):
        n = pc.info.num_values
        valid = unpack(col.validity.to_numpy(np.uint8, pc.info.validity_bytes).tobytes(), n) if col.validity is not None else np.ones(n, bool)
        raw = col.data.to_numpy(np.uint8, pc.info.out_bytes).tobytes()
        got = PU.decoded_to_python(raw, valid, ot, n, chunk=pc.device_image())
        assert np.array_equal(valid, exp_valid) and got == exp, name
        assert pc.nulls == int((~exp_valid).sum()), name
    # one corrupt chunk (a ZSTD frame that loses its last bytes) among intact ones
    compress = _zstd()
    plain = np.arange(50_000, dtype=np.int64).tobytes()
    z = compress(plain, 3)[:-5]
    bad = gpu.ParquetChunk(PU.raw_page_chunk(z, len(plain), len(plain) // 8), PU.PHYS["INT64"], T.T_I64, 0, 0, 0, 6, device=True)
    good = gpu.ParquetChunk(PU.raw_page_chunk(compress(plain, 3), len(plain), len(plain) // 8), PU.PHYS["INT64"], T.T_I64, 0, 0, 0, 6, device=True)
    st = []
    cols = gpu.ParquetChunk.decode_many([good, bad, opened[0][0]], statuses=st)
    assert st[0] == 0 and st[2] == 0 and st[1] == T.ERR_INVALID and cols[1] is None
    assert np.array_equal(cols[0].data.to_numpy(np.int64, 50_000), np.arange(50_000))
    with pytest.raises(T.DbhipError):
        gpu.ParquetChunk.decode_many([good, bad])
    for o in opened:
        o[0].close()
    bad.close()
    good.close()


def test_full_size_device_mode_20m_rows(gpu):
    """BASELINE-sized chunk (20 M Decimal(15,2) values of lineitem, 3 % NULLs, PLAIN v1 pages of 1 MiB) stored uncompressed, with Snappy,
    LZ4 and ZSTD: write -> decode is the identity; the rates (bytes of the chunk as stored per second of the decode call) go to
    gpurun_out/pq_device_rates.json."""
    import pyarrow as pa
    n = 20_000_000
    rng = np.random.default_rng(4)
    price = rng.integers(90000, 10494951, n)
    mask = rng.random(n) < 0.03
    disc = rng.integers(0, 11, n)
    rates = {}
    for cname in ("none", "snappy", "lz4", "zstd"):
        for label, arr, dictionary, src, m in (("price_plain_nullable", pa.array(price, pa.int64(), mask=mask), False, price, mask),
                                               ("discount_dictionary", pa.array(disc, pa.int64()), True, disc, None)):
            chunks, _ = PU.column_chunks(PU.write_parquet(pa.table({"c": arr}), dictionary=dictionary, compression=cname))
            ch = chunks[0]
            pc = gpu.ParquetChunk(ch["chunk"], ch["physical"], T.T_DEC64, ch["type_length"], ch["max_def"], 0, ch["codec"], precision=15, scale=2, device=True)
            col = pc.decode()
            got = col.data.to_numpy(np.int64, n)
            if m is not None:
                valid = unpack(col.validity.to_numpy(np.uint8, pc.info.validity_bytes).tobytes(), n)
                assert np.array_equal(valid, ~m) and np.array_equal(got, np.where(m, 0, src)) and pc.nulls == int(m.sum())
            else:
                assert np.array_equal(got, src)
            best = 1e9
            for _ in range(3):
                t0 = time.perf_counter()
                pc.decode()                                   # (decode_device synchronises its stream)
                best = min(best, time.perf_counter() - t0)
            rates[f"{label}/{cname}"] = dict(chunk_bytes=len(ch["chunk"]), image_bytes=int(pc.info.image_bytes), pages=int(pc.info.n_pages),
                                             decode_ms=round(best * 1e3, 3), stored_GBps=round(len(ch["chunk"]) / best / 1e9, 2),
                                             out_GBps=round(n * 8 / best / 1e9, 2))
            pc.close()
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(rates, open("gpurun_out/pq_device_rates.json", "w"), indent=1)
    print(json.dumps(rates))


# ---- List<primitive> (round 5) --------------------------------------------------------------------------------------------------------
def _list_cases(rng, n):
    """name -> (pyarrow array of lists, list_nullable, element_nullable, out type, element -> python)"""
    import pyarrow as pa
    lens = rng.integers(0, 6, n)
    lens[rng.random(n) < 0.1] = 0                                           # empty lists
    null_list = rng.random(n) < 0.08
    def lists(make, elem_null):
        out = []
        for i in range(n):
            if null_list[i]:
                out.append(None)
            else:
                out.append([None if (elem_null and rng.random() < 0.2) else make() for _ in range(lens[i])])
        return out
    req = lambda ls: [[] if x is None else x for x in ls]
    cases = {}
    ints = lists(lambda: int(rng.integers(-10**12, 10**12)), True)
    cases["list<int64> both nullable"] = (pa.array(ints, pa.list_(pa.int64())), 1, 1, T.T_I64)
    ints2 = lists(lambda: int(rng.integers(0, 50)), False)
    cases["list<int32 not null> nullable (dictionary)"] = (pa.array(ints2, pa.list_(pa.field("item", pa.int32(), nullable=False))), 1, 0, T.T_I32)
    f = req(lists(lambda: float(rng.integers(-1000, 1000)) / 8, True))
    cases["required list<double>"] = (pa.array(f, pa.list_(pa.float64())), 0, 1, T.T_F64)
    u = req(lists(lambda: int(rng.integers(0, 2**31)), False))
    cases["required list<int64 not null>"] = (pa.array(u, pa.list_(pa.field("item", pa.int64(), nullable=False))), 0, 0, T.T_I64)
    strs = lists(lambda: (b"s%d" % rng.integers(0, 10**6)) * int(rng.integers(1, 4)), True)
    cases["list<binary> both nullable"] = (pa.array(strs, pa.list_(pa.binary())), 1, 1, T.T_STRING)
    bools = lists(lambda: bool(rng.integers(0, 2)), True)
    cases["list<bool> both nullable"] = (pa.array(bools, pa.list_(pa.bool_())), 1, 1, T.T_BOOL)
    return cases


LIST_NAMES = list(_list_cases(np.random.default_rng(0), 4).keys())


@pytest.mark.parametrize("name", LIST_NAMES)
@pytest.mark.parametrize("codec,v2,dictionary,n", [("none", False, False, 1), ("none", True, True, 3000), ("zstd", True, True, 40_000), ("snappy", False, False, 40_000),
                                                   ("lz4", True, False, 250_000)])
def test_list_columns_decode_on_the_device(gpu, name, codec, v2, dictionary, n):
    """List<primitive> leaves (max_rep 1): repetition / definition levels, offsets, both validities and the element values against what
    pyarrow reads back from the same file (arrow C++ — the reference reads these columns through arrow-rs, deserialize.rs:33-81): V1 and V2
    pages, several pages per chunk (rows spanning pages), dictionary and PLAIN values, every codec of the device path."""
    import pyarrow as pa
    import pyarrow.parquet as pq
    import io
    rng = np.random.default_rng(n + len(name))
    arr, ln, en, ot = _list_cases(rng, n)[name]
    field = pa.field("c", arr.type, nullable=bool(ln))
    table = pa.Table.from_arrays([arr], schema=pa.schema([field]))
    data = PU.write_parquet(table, dictionary=dictionary, v2=v2, compression=codec, page_size=4096)
    ch = PU.column_chunks(data)[0][0]
    assert ch["max_rep"] == 1 and ch["max_def"] == ln + 1 + en
    back = pq.read_table(io.BytesIO(data)).column(0).to_pylist()
    pc = gpu.ParquetChunk(ch["chunk"], ch["physical"], ot, ch["type_length"], codec=ch["codec"], list_of=(ln, en))
    offs, lv, col = pc.decode_list()
    assert pc.rows == n and len(offs) == n + 1 and int(offs[0]) == 0 and int(offs[-1]) == pc.elems == col.n
    assert pc.null_lists == sum(1 for x in back if x is None)
    ev = col.validity_numpy() if en else np.ones(col.n, bool)
    vals = col.to_strings() if ot == T.T_STRING else col.to_numpy().tolist()
    got = []
    for r in range(n):
        if lv is not None and not lv[r]:
            assert offs[r] == offs[r + 1]
            got.append(None)
        else:
            got.append([vals[x] if ev[x] else None for x in range(int(offs[r]), int(offs[r + 1]))])
    assert got == back
    pc.close()


def test_list_chunks_malformed_or_deeper_nesting(gpu):
    import pyarrow as pa
    rng = np.random.default_rng(5)
    arr, ln, en, ot = _list_cases(rng, 20_000)["list<int64> both nullable"]
    table = pa.Table.from_arrays([arr], schema=pa.schema([pa.field("c", arr.type)]))
    ch = PU.column_chunks(PU.write_parquet(table, dictionary=False, v2=False, page_size=4096))[0][0]
    # the flat open refuses it and names the List entry point; List<List<..>> has max_rep 2: refused by the binding's own check of max_rep
    with pytest.raises(T.DbhipError, match="open_device_list"):
        gpu.ParquetChunk(ch["chunk"], ch["physical"], ot, ch["type_length"], ch["max_def"], ch["max_rep"], device=True)
    # corrupted level bytes: a status or the right answer, never a crash (the level streams sit right behind the 4-byte lengths of page 1)
    base = bytearray(ch["chunk"])
    for k in range(60):
        bad = bytearray(base)
        pos = int(rng.integers(20, min(len(bad), 4000)))
        bad[pos] ^= 1 << int(rng.integers(0, 8))
        try:
            pc = gpu.ParquetChunk(bytes(bad), ch["physical"], ot, ch["type_length"], codec=ch["codec"], list_of=(ln, en))
        except T.DbhipError:
            continue
        try:
            pc.decode_list()
        except T.DbhipError as e:
            assert e.code in (T.ERR_INVALID, T.ERR_UNSUPPORTED)
        pc.close()


@pytest.mark.parametrize("name", LIST_NAMES)
def test_list_columns_equal_the_oracle(gpu, oracle, name):
    """the device's List decode against oracle/parquet_oracle.c::orc_pq_decode_list (itself pinned on pyarrow in the CPU suite): offsets,
    list validity, element validity and the element bytes, bit for bit (uncompressed v1 chunk of several pages, PLAIN values)"""
    import pyarrow as pa
    n = 30_000
    rng = np.random.default_rng(len(name))
    arr, ln, en, ot = _list_cases(rng, n)[name]
    table = pa.Table.from_arrays([arr], schema=pa.schema([pa.field("c", arr.type, nullable=bool(ln))]))
    ch = PU.column_chunks(PU.write_parquet(table, dictionary=False, v2=False, page_size=4096))[0][0]
    ent = ch["num_values"]
    chunk = np.frombuffer(ch["chunk"], dtype=np.uint8)
    es = 1 if ot == T.T_BOOL else PU.ESIZE[ot]
    offs = np.zeros(ent + 2, np.uint64)
    lval = np.zeros(ent + 1, np.uint8)
    vals = np.zeros(max(ent, 1) * es + 16, np.uint8)
    ev = np.zeros(ent + 1, np.uint8)
    rows, elems = C.c_int64(), C.c_int64()
    oracle.orc_pq_decode_list.restype = C.c_int
    rc = oracle.orc_pq_decode_list(chunk.ctypes.data_as(C.c_void_p), C.c_int64(len(chunk)), ch["physical"], ch["type_length"], ln, en, ot, C.c_int64(ent),
                                   offs.ctypes.data_as(C.c_void_p), lval.ctypes.data_as(C.c_void_p), vals.ctypes.data_as(C.c_void_p), ev.ctypes.data_as(C.c_void_p),
                                   C.byref(rows), C.byref(elems))
    assert rc == 0 and rows.value == n
    pc = gpu.ParquetChunk(ch["chunk"], ch["physical"], ot, ch["type_length"], codec=ch["codec"], list_of=(ln, en))
    goffs, glv, col = pc.decode_list()
    m = elems.value
    assert pc.rows == n and pc.elems == m and np.array_equal(goffs, offs[: n + 1])
    if ln:
        assert np.array_equal(glv, lval[:n].astype(bool))
    if en:
        assert np.array_equal(col.validity_numpy(), ev[:m].astype(bool))
    if ot == T.T_BOOL:
        assert np.array_equal(col.to_numpy(), vals[:m].astype(bool))
    elif ot == T.T_STRING:      # (views point into different copies of the chunk: compare the strings)
        ovalid = ev[:m].astype(bool)
        assert [s if ok else None for s, ok in zip(col.to_strings(), ovalid)] == PU.decoded_to_python(vals.tobytes(), ovalid, ot, m, chunk)
    else:
        assert col.data.to_numpy(np.uint8, m * es).tobytes() == vals[: m * es].tobytes()
    pc.close()


def test_list_column_of_the_reference_held_multi_page_files(gpu):
    """col_arr of tests/data/parquet/multi_page/multi_page_{1..4}.parquet through dbhip_pq_chunk_open_device_list / decode_device_list:
    [[1], [1, 2]] * num_row (gen.py:12), SNAPPY dictionary-encoded v1 pages, rows spanning pages — the reference-held pin of the List
    decode (the CPU suite runs the same fixtures through the oracle)."""
    from tests import parquet_ref as PR

    def decode_list(ch, ln, en, ot):
        pc = gpu.ParquetChunk(ch["chunk"], ch["physical"], ot, ch["type_length"], codec=ch["codec"], list_of=(ln, en))
        offs, lv, col = pc.decode_list()
        assert int(offs[-1]) == pc.elems == col.n and pc.null_lists == 0
        ev = col.validity_numpy() if en else np.ones(col.n, bool)
        vals = col.to_numpy().tolist()
        out = [None if (lv is not None and not lv[r]) else [vals[x] if ev[x] else None for x in range(int(offs[r]), int(offs[r + 1]))] for r in range(pc.rows)]
        pc.close()
        return out
    assert PR.check_lists(decode_list) == 4


def test_map_column_of_the_reference_held_no_stats_file_as_two_list_decodes(gpu):
    """Map(String, String) `product` of tests/data/parquet/no-stats.parquet (parquet-mr, SNAPPY, dictionary-encoded v1 pages): key and value
    leaves through dbhip_pq_chunk_open_device_list / decode_device_list with (list_nullable, element_nullable) = (1, 0) and (1, 1); the two
    decodes agree on offsets and list validity and zip to pyarrow's reading of all 25,825 rows / 188,558 entries (VERDICT r05 missing #4:
    Map members — a binding composes a Map from two List decodes, nothing new on the device)."""
    from tests import parquet_ref as PR

    def decode_list(ch, ln, en, ot):
        pc = gpu.ParquetChunk(ch["chunk"], ch["physical"], ot, ch["type_length"], codec=ch["codec"], list_of=(ln, en))
        offs, lv, col = pc.decode_list()
        ev = col.validity_numpy() if en else np.ones(col.n, bool)
        vals = col.to_strings()
        out = [None if (lv is not None and not lv[r]) else [vals[x] if ev[x] else None for x in range(int(offs[r]), int(offs[r + 1]))] for r in range(pc.rows)]
        pc.close()
        return out
    assert PR.check_map(decode_list) == 1
