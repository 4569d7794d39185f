"""The stored form of the reference's HNSW index (databend_amd/hnsw_format.py): known answers from the reference's own tests,
writer <-> reader agreement over random graphs, the file layout of graph_links/header.rs, and (GPU) save -> open -> identical
search results."""
import struct

import numpy as np
import pytest

from databend_amd import hnsw_format as F


def test_bitwriter_known_answers_of_the_reference():
    """bitpacking.rs:268-294 (test_simple) and :357-373 (test_packed_bits_simple)"""
    out = bytearray()
    w = F.BitWriter(out)
    for v, b in ((0b01010, 5), (0b10110, 5), (0b10100, 5), (0b010110010, 9), (0b101100001, 9), (0b001001101, 9), (0x12345678, 32)):
        w.write(v, b)
    w.finish()
    assert len(out) == 10
    r = F.BitReader(out)
    r.set_bits(5)
    assert [r.read() for _ in range(3)] == [0b01010, 0b10110, 0b10100]
    r.set_bits(9)
    assert [r.read() for _ in range(3)] == [0b010110010, 0b101100001, 0b001001101]
    r.set_bits(32)
    assert r.read() == 0x12345678
    assert [F.packed_bits(x) for x in (0, 1, 2, 3, 4, 7, 0x7FFFFFFF, 0x80000000, 0xFFFFFFFF)] == [0, 1, 2, 2, 3, 3, 31, 32, 32]


@pytest.mark.parametrize("seed", range(6))
def test_pack_links_round_trip(seed):
    """pack_links / iterate_packed_links (bitpacking_links.rs): the sorted prefix comes back ASCENDING (the file does not keep
    the order of the first m / m0 links), the rest in order"""
    rng = np.random.default_rng(seed)
    for _ in range(200):
        npoints = int(rng.choice([5, 300, 70_000, 3_000_000]))
        bpu = max(F.packed_bits(npoints - 1), 8)
        k = int(rng.integers(0, 40))
        links = rng.choice(npoints, size=min(k, npoints), replace=False).tolist()
        sc = int(rng.choice([0, 8, 16, 32, 64]))
        out = bytearray()
        F.pack_links(out, links, bpu, sc)
        s = min(len(links), sc)
        assert F.unpack_links(out, bpu, sc) == sorted(links[:s]) + links[s:]
        if not links:
            assert len(out) == 0


@pytest.mark.parametrize("seed", range(4))
def test_offsets_compress_round_trip_and_parameters(seed):
    """bitpacking_ordered.rs: compress picks the smallest of the 8 chunk lengths; every value reads back; the tail is 7 x 0xFF"""
    rng = np.random.default_rng(10 + seed)
    for n in (1, 2, 127, 128, 129, 1000, 5000):
        steps = rng.integers(0, [3, 40, 300, 5000][seed], n)
        vals = np.cumsum(steps).tolist()
        comp, p = F.compress_offsets(vals)
        assert p.valid() and p.length == n and comp[-7:] == b"\xff" * 7 and len(comp) == p.total_chunks_size_bytes() + 7
        for log2 in range(8):   # no other chunk length is smaller
            q = F.OffsetParameters(n, p.base_bits, 1, log2)
            step = 1 << log2
            q.delta_bits = max([1] + [F.packed_bits(vals[min(i + step, n) - 1] - vals[i]) for i in range(0, n, step)])
            assert q.total_chunks_size_bytes() >= p.total_chunks_size_bytes()
        r = F.OffsetReader(F.OffsetParameters.from_bytes(p.to_bytes()), comp + b"trailing bytes are ignored")
        assert [r.get(i) for i in range(n)] == vals
        with pytest.raises(IndexError):
            r.get(n)


def random_graph(rng, n, m):
    u = rng.random(n)
    levels = np.round(-np.log(np.maximum(u, 1e-12)) / np.log(max(m, 2))).astype(np.int32)
    lists = []
    for p in range(n):
        for lv in range(levels[p] + 1):
            cand = np.nonzero(levels >= lv)[0]
            cand = cand[cand != p]
            k = int(rng.integers(0, (2 * m if lv == 0 else m) + 1))
            lists.append(rng.choice(cand, size=min(k, len(cand)), replace=False).astype(np.uint32))
    return levels, lists


@pytest.mark.parametrize("n,m", [(1, 4), (2, 4), (300, 4), (3000, 10), (70_000, 16)])
def test_graph_links_file_layout_and_round_trip(n, m):
    rng = np.random.default_rng(n)
    levels, lists = random_graph(rng, n, m)
    data = F.write_graph_links(levels, lists, m, 2 * m)
    # HeaderCompressed (graph_links/header.rs:36-50): 64 bytes, little endian
    pc, ver, lc, tlb = struct.unpack("<QQQQ", data[:32])
    assert (pc, ver, lc) == (n, 0xFFFFFFFFFFFFFF01, int(levels.max()) + 1)
    length, base_bits, delta_bits, log2 = struct.unpack("<QBBB", data[32:43])
    assert struct.unpack("<QQ", data[43:59]) == (m, 2 * m) and data[59:64] == b"\0" * 5
    nlists = int((levels + 1).sum())
    assert length == nlists + 1                                   # one offset per list, plus the leading 0
    lo = struct.unpack(f"<{lc}Q", data[64:64 + 8 * lc])
    assert lo[0] == 0 and all(lo[k + 1] - lo[k] == int((levels >= k).sum()) for k in range(lc - 1))
    reindex = np.frombuffer(data, dtype=np.uint32, count=n, offset=64 + 8 * lc)
    assert sorted(reindex.tolist()) == list(range(n))             # a permutation: position of the point in the by-level order
    assert all(levels[a] >= levels[b] for a, b in zip(np.argsort(reindex)[:-1], np.argsort(reindex)[1:]))
    assert len(data) == 64 + 8 * lc + 4 * n + tlb + F.OffsetParameters(length, base_bits, delta_bits, log2).total_chunks_size_bytes() + 7
    got_levels, got_lists, gm, gm0 = F.read_graph_links(data)
    assert (gm, gm0) == (m, 2 * m) and np.array_equal(got_levels, levels) and len(got_lists) == len(lists)
    for a, b in zip(got_lists, lists):
        assert np.array_equal(a, np.sort(b))                      # lists of <= m / m0 links are stored sorted
    with pytest.raises(ValueError):
        F.read_graph_links(data[:40])
    with pytest.raises(ValueError):
        F.read_graph_links(data[:64 + 8 * lc + 4 * n + tlb - 1] if tlb else data[:63])


def test_graph_data_is_bincode_standard_varints():
    """bincode 2 standard configuration: u < 251 one byte, 251 + u16, 252 + u32, 253 + u64 (little endian)"""
    b = F.write_graph_data(10, 20, 40, [(70000, 3)])
    assert b == bytes([10, 20, 40, 1, 252]) + struct.pack("<I", 70000) + bytes([3, 0, 2])
    assert F.read_graph_data(b) == {"m": 10, "m0": 20, "ef_construct": 40, "entry_points": [(70000, 3)], "extra_entry_points": [], "extra_length": 2}
    b = F.write_graph_data(16, 32, 300, [(5, 1)], [(250, 0), (251, 1), (2 ** 40, 2)])
    d = F.read_graph_data(b)
    assert d["ef_construct"] == 300 and d["extra_entry_points"] == [(250, 0), (251, 1), (2 ** 40, 2)] and F.entry_point_of(d) == (5, 1)
    assert F.entry_point_of({"entry_points": [], "extra_entry_points": [(1, 2), (7, 4), (9, 4)]}) == (9, 4)


def test_encoded_meta_is_the_references_json():
    b = F.write_encoded_meta(32, np.float32(0.1), np.float32(-1.5), np.float32(0.01), 24, 1000, "cosine")
    d = F.read_encoded_meta(b)
    assert list(d) == ["actual_dim", "alpha", "offset", "multiplier", "vector_parameters"]
    assert d["vector_parameters"] == {"dim": 24, "count": 1000, "distance_type": "Dot", "invert": False}
    assert d["alpha"] == np.float32(0.1) and d["offset"] == np.float32(-1.5) and d["multiplier"] == np.float32(0.01)
    assert F.read_encoded_meta(F.write_encoded_meta(16, 1, 2, 3, 9, 5, "l2"))["vector_parameters"]["invert"] is True


@pytest.mark.gpu
@pytest.mark.parametrize("distance", ["cosine", "l2", "l1"])
def test_saved_index_opens_and_searches_identically(gpu, distance):
    """build on the device -> the four Binary columns -> HNSWIndex::open (dbhip_hnsw_open) -> the same answers. The opened
    index has no original vectors; its lists are the stored (sorted) ones, so the comparison index is made of the same sorted
    lists through dbhip_hnsw_from_graph (the order of a list only decides which of two equally scored neighbours is seen first)."""
    from databend_amd import _lib as T
    METRIC = {"cosine": T.VEC_COSINE, "l2": T.VEC_L2, "l1": T.VEC_L1}
    rng = np.random.default_rng(3)
    n, dim, m = 4000, 40, 10
    raw = rng.standard_normal((n, dim)).astype(np.float32)
    base = gpu.VectorColumn(raw)
    idx = gpu.HnswIndex.build(METRIC[distance], base, m=m, ef_construct=40, seed=7)
    cols = F.save_index(idx, distance, m, 40)
    assert [type(c) for c in cols] == [bytes] * 4 and len(cols[3]) == n * (4 + idx.meta()[3])
    opened = F.open_index(METRIC[distance], distance, dim, n, cols)
    levels, lists, ep, el = idx.export_graph()
    same = gpu.HnswIndex.from_graph(METRIC[distance], base, m, levels, [np.sort(x) for x in lists], ep, el)
    l2, s2, ep2, el2 = opened.export_graph()
    assert np.array_equal(l2, levels) and (ep2, el2) == (ep, el) and all(np.array_equal(a, np.sort(b)) for a, b in zip(s2, lists))
    assert opened.meta() == idx.meta() and np.array_equal(opened.encoded(), idx.encoded())
    q = gpu.VectorColumn(rng.standard_normal((200, dim)).astype(np.float32))
    i1, d1 = same.search(q, 10)
    i2, d2 = opened.search(q, 10)
    assert np.array_equal(i1, i2) and np.array_equal(d1, d2, equal_nan=True)
    # an empty index round-trips too
    e = gpu.HnswIndex.build(METRIC[distance], gpu.VectorColumn(np.zeros((0, dim), np.float32)), m=m, ef_construct=40, seed=1)
    ecols = F.save_index(e, distance, m, 40)
    assert F.read_graph_links(ecols[0])[0].shape == (0,)
    for x in (idx, opened, same, e):
        x.destroy()


def test_graph_links_bytes_of_a_hand_worked_graph():
    """Three points, m = 2 (m0 = 4), levels [0, 1, 0]; lists p0/L0 = [1, 2], p1/L0 = [0, 2], p1/L1 = [], p2/L0 = [1]. Every byte below was
    derived BY HAND from serializer.rs / bitpacking_links.rs / bitpacking_ordered.rs (not by running the writer): back_index = [1, 0, 2]
    (most levels first), reindex = [1, 0, 2]; level offsets [0, 3]; bits_per_unsorted = 8; each non-empty list = 5-bit header 0
    (bits_per_sorted = 8) + 8-bit deltas, LSB first, padded to a byte; offsets [0, 3, 6, 8, 8] -> base_bits 4, delta_bits 2,
    chunk_len_log2 1 (3 bytes, the smallest of the eight candidates) + the 7-byte 0xFF tail."""
    levels = [0, 1, 0]
    lists = [np.array([1, 2], np.uint32), np.array([0, 2], np.uint32), np.array([], np.uint32), np.array([1], np.uint32)]
    header = struct.pack("<QQQQ", 3, 0xFFFFFFFFFFFFFF01, 2, 8) + struct.pack("<QBBB", 5, 4, 2, 1) + struct.pack("<QQ", 2, 4) + b"\0" * 5
    expected = (header + struct.pack("<QQ", 0, 3) + struct.pack("<III", 1, 0, 2)
                + bytes([0x20, 0x20, 0x00]) + bytes([0x00, 0x40, 0x00]) + bytes([0x20, 0x00])
                + bytes([0x30, 0x26, 0x38]) + b"\xff" * 7)
    got = F.write_graph_links(levels, lists, 2, 4)
    assert got == expected, (got.hex(), expected.hex())
    lv, ls, m, m0 = F.read_graph_links(expected)
    assert lv.tolist() == levels and [x.tolist() for x in ls] == [[1, 2], [0, 2], [], [1]] and (m, m0) == (2, 4)
