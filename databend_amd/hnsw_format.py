"""The STORED form of the reference's HNSW index: the four Binary columns HNSWIndex::build emits and HNSWIndex::open reads
(src/query/storages/common/index/src/hnsw_index/hnsw.rs:62-98, 237-300):

    <col>-<distance>_graph_links       GraphLinks, Compressed format   graph_links/{header,serializer,view}.rs
    <col>-<distance>_graph_data        bincode 2 (standard config) GraphLayerData { m, m0, ef_construct, entry_points }
                                       graph_layers.rs:46-52, graph_layers_builder.rs:100-121, entry_points.rs:24-48
    <col>-<distance>_encoded_u8_meta   serde_json Metadata { actual_dim, alpha, offset, multiplier, vector_parameters }
                                       quantization/encoded_vectors_u8.rs:45-52, 306-310
    <col>-<distance>_encoded_u8_data   per vector: f32 offset + actual_dim codes (quantization/encoded_storage.rs)

Host-side reader and writer (byte work of a few MB per block: the graph walk and the scoring stay on the device). The reader
follows view.rs / bitpacking_links.rs::iterate_packed_links / bitpacking_ordered.rs::Reader, the writer follows serializer.rs /
pack_links / compress — two separately restated paths that must agree with each other (tests/test_hnsw_format.py) and with the
few known answers the reference's tests hold (bitpacking.rs:268-294, 357-373). PARITY UNPINNED beyond that: no index file written
by the reference is available here, and the image has no Rust toolchain to produce one.
One documented difference in the WRITER: GraphLinksSerializer orders points of equal top level with `sort_unstable_by_key`
(serializer.rs:62), whose order of equal keys is an implementation detail of the Rust standard library; this writer orders them
by point id. Both are valid files (the order is stored in `reindex`), they are just not byte-identical.
"""
import json
import struct

import numpy as np

HEADER_VERSION_COMPRESSED = 0xFFFF_FFFF_FFFF_FF01   # graph_links/header.rs:52
MIN_BITS_PER_VALUE = 8                              # bitpacking_links.rs:21
HEADER_BITS = 5                                     # bitpacking_links.rs:24
TAIL_SIZE = 7                                       # bitpacking_ordered.rs:29
MAX_CHUNK_LEN_LOG2 = 7                              # bitpacking_ordered.rs:39
HEADER_COMPRESSED_BYTES = 64
DISTANCE_NAME = {"cosine": "Dot", "dot": "Dot", "l1": "L1", "l2": "L2"}   # encoded_vectors.rs:25-30 (cosine is Dot over a normalised column)


def packed_bits(x):
    """bitpacking.rs:236-238: bits needed for x (0 for 0)"""
    return int(x).bit_length()


class BitWriter:
    """bitpacking.rs:55-106: values appended LSB first; finish() pads to a byte"""

    def __init__(self, out):
        self.out, self.acc, self.nbits = out, 0, 0

    def write(self, value, bits):
        assert 0 <= value < (1 << bits) or bits == 0 and value == 0
        self.acc |= int(value) << self.nbits
        self.nbits += bits

    def finish(self):
        self.out += self.acc.to_bytes((self.nbits + 7) // 8, "little")


class BitReader:
    """bitpacking.rs:108-190 (reads past the end give zeros, as read_buf_and_advance does)"""

    def __init__(self, data):
        self.acc, self.pos, self.bits = int.from_bytes(bytes(data), "little"), 0, 0

    def set_bits(self, bits):
        self.bits = bits

    def read(self):
        v = (self.acc >> self.pos) & ((1 << self.bits) - 1)
        self.pos += self.bits
        return v


def pack_links(out, raw_links, bits_per_unsorted, sorted_count):
    """bitpacking_links.rs:40-73: the first `sorted_count` links sorted and delta coded at their own width (5-bit header =
    width - 8), the rest at `bits_per_unsorted`; nothing at all for an empty list"""
    raw = [int(x) for x in raw_links]
    if not raw:
        return
    sc = min(len(raw), sorted_count)
    head = sorted(raw[:sc])
    deltas = [head[0]] + [head[i] - head[i - 1] for i in range(1, sc)] if sc else []
    w = BitWriter(out)
    if sc:
        bps = max(packed_bits(max(deltas)), MIN_BITS_PER_VALUE)
        w.write(bps - MIN_BITS_PER_VALUE, HEADER_BITS)
        for d in deltas:
            w.write(d, bps)
    for v in raw[sc:]:
        w.write(v, bits_per_unsorted)
    w.finish()


def unpack_links(data, bits_per_unsorted, sorted_count):
    """bitpacking_links.rs:76-160 (iterate_packed_links + PackedLinksIterator::fold)"""
    data = bytes(data)
    r = BitReader(data)
    remaining = len(data) * 8
    target = remaining
    out = []
    if sorted_count != 0 and data:
        r.set_bits(HEADER_BITS)
        bps = r.read() + MIN_BITS_PER_VALUE
        remaining -= HEADER_BITS
        r.set_bits(bps)
        target -= min(sorted_count, remaining // bps) * bps
    else:
        r.set_bits(bits_per_unsorted)
    cur = 0
    while remaining > target:
        cur = (cur + r.read()) & 0xFFFFFFFF
        remaining -= r.bits
        out.append(cur)
    r.set_bits(bits_per_unsorted)
    while remaining >= r.bits:
        remaining -= r.bits
        out.append(r.read())
    return out


class OffsetParameters:
    """bitpacking_ordered.rs:228-236 (11 bytes: length u64 LE, base_bits, delta_bits, chunk_len_log2)"""

    def __init__(self, length, base_bits, delta_bits, chunk_len_log2):
        self.length, self.base_bits, self.delta_bits, self.chunk_len_log2 = length, base_bits, delta_bits, chunk_len_log2

    def valid(self):
        return self.base_bits <= 64 and 1 <= self.delta_bits <= 56 and self.chunk_len_log2 <= MAX_CHUNK_LEN_LOG2

    def chunk_size_bytes(self):
        return (self.base_bits + self.delta_bits * ((1 << self.chunk_len_log2) - 1) + 7) // 8

    def total_chunks_size_bytes(self):
        return -(-self.length // (1 << self.chunk_len_log2)) * self.chunk_size_bytes()

    def to_bytes(self):
        return struct.pack("<QBBB", self.length, self.base_bits, self.delta_bits, self.chunk_len_log2)

    @classmethod
    def from_bytes(cls, b):
        return cls(*struct.unpack("<QBBB", bytes(b[:11])))

    @classmethod
    def find_best(cls, values):
        """:259-285: every chunk length 2^0 .. 2^7, the smallest total (the first of equals: Iterator::min_by_key)"""
        last = values[-1] if len(values) else 0
        best = None
        for log2 in range(MAX_CHUNK_LEN_LOG2 + 1):
            step = 1 << log2
            delta_bits = 1
            for i in range(0, len(values), step):
                chunk_last = values[min(i + step, len(values)) - 1]
                delta_bits = max(delta_bits, packed_bits(chunk_last - values[i]))
            p = cls(len(values), max(packed_bits(last), 1), delta_bits, log2)
            if not 1 <= delta_bits <= 56:
                continue
            if best is None or p.total_chunks_size_bytes() < best.total_chunks_size_bytes():
                best = p
        return best


def compress_offsets(values):
    """bitpacking_ordered.rs:41-72: chunks of 2^k ascending values as (first, deltas to the first ...), missing slots of the
    last chunk filled with all-ones deltas, every chunk padded to a byte, 7 bytes of 0xFF behind the last one"""
    values = [int(v) for v in values]
    p = OffsetParameters.find_best(values)
    out = bytearray()
    step = 1 << p.chunk_len_log2
    for i in range(0, len(values), step):
        chunk = values[i:i + step]
        w = BitWriter(out)
        w.write(chunk[0], p.base_bits)
        for v in chunk[1:]:
            w.write(v - chunk[0], p.delta_bits)
        for _ in range(step - len(chunk)):
            w.write((1 << p.delta_bits) - 1, p.delta_bits)
        w.finish()
    out += b"\xff" * TAIL_SIZE
    assert len(out) == p.total_chunks_size_bytes() + TAIL_SIZE
    return bytes(out), p


class OffsetReader:
    """bitpacking_ordered.rs:74-215 (Reader::new / get)"""

    def __init__(self, p, data):
        if not p.valid():
            raise ValueError("graph_links: invalid offset parameters")
        total = p.total_chunks_size_bytes() + TAIL_SIZE
        if len(data) < total:
            raise ValueError(f"graph_links: insufficient length (compressed offsets, expected {total} bytes, got {len(data)})")
        self.p, self.data = p, bytes(data[:total]) + b"\0" * 8
        self.csb = p.chunk_size_bytes()

    def __len__(self):
        return self.p.length

    def get(self, index):
        p = self.p
        if index >= p.length:
            raise IndexError(index)
        co = (index >> p.chunk_len_log2) * self.csb
        vi = index & ((1 << p.chunk_len_log2) - 1)
        base = int.from_bytes(self.data[co:co + 8], "little") & ((1 << p.base_bits) - 1 if p.base_bits < 64 else (1 << 64) - 1)
        if vi == 0:
            return base
        bit = p.base_bits + (vi - 1) * p.delta_bits
        word = int.from_bytes(self.data[co + bit // 8:co + bit // 8 + 8], "little")
        return base + ((word >> (bit % 8)) & ((1 << p.delta_bits) - 1))


# ---- graph_links (Compressed) ------------------------------------------------------------------------------------------
def write_graph_links(levels, lists, m, m0):
    """GraphLinksSerializer::new + serialize_to_writer (serializer.rs:52-233). `lists`: the link lists in point-major,
    level-minor order (dbhip_hnsw_export_graph's layout); a point of level L has L + 1 lists."""
    levels = np.asarray(levels, dtype=np.int64)
    n = len(levels)
    first = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(levels + 1, out=first[1:])
    # back_index: points by descending number of levels (ties: by id — see the module docstring); reindex = its inverse
    back_index = sorted(range(n), key=lambda i: (-int(levels[i]), i))
    reindex = np.zeros(n, dtype=np.uint32)
    for pos, pid in enumerate(back_index):
        reindex[pid] = pos
    levels_count = int(levels.max()) + 1 if n else 0
    count_by_level = np.bincount(levels, minlength=levels_count) if n else np.zeros(0, np.int64)
    level_offsets, total, suffix = [], 0, int(count_by_level.sum())
    for v in count_by_level.tolist():
        level_offsets.append(total)
        total += suffix
        suffix -= v
    links = bytearray()
    offsets = [0]
    bits_per_unsorted = max(packed_bits(max(n - 1, 0)), MIN_BITS_PER_VALUE)
    for level in range(levels_count):
        count = int(count_by_level[level:].sum())
        ids = range(count) if level == 0 else back_index[:count]
        for pid in ids:
            pack_links(links, lists[int(first[pid]) + level], bits_per_unsorted, m0 if level == 0 else m)
            offsets.append(len(links))
    comp, params = compress_offsets(offsets)
    header = (struct.pack("<QQQQ", n, HEADER_VERSION_COMPRESSED, levels_count, len(links)) + params.to_bytes() + struct.pack("<QQ", m, m0) + b"\0" * 5)
    assert len(header) == HEADER_COMPRESSED_BYTES
    return header + struct.pack(f"<{levels_count}Q", *level_offsets) + reindex.tobytes() + bytes(links) + comp


def read_graph_links(data):
    """GraphLinksView::load_compressed + links + point_level (view.rs:92-166) -> (levels, lists, m, m0); lists in
    point-major, level-minor order"""
    data = bytes(data)
    if len(data) < HEADER_COMPRESSED_BYTES:
        raise ValueError("Unsufficent file size for GraphLinks file")
    n, version, levels_count, total_links_bytes = struct.unpack("<QQQQ", data[:32])
    if version != HEADER_VERSION_COMPRESSED:
        raise ValueError(f"GraphLinks file: not the compressed format (version {version:#x})")
    params = OffsetParameters.from_bytes(data[32:43])
    m, m0 = struct.unpack("<QQ", data[43:59])
    pos = HEADER_COMPRESSED_BYTES
    need = pos + 8 * levels_count + 4 * n + total_links_bytes
    if len(data) < need or params.length < 1:
        raise ValueError("Unsufficent file size for GraphLinks file")
    level_offsets = list(struct.unpack(f"<{levels_count}Q", data[pos:pos + 8 * levels_count])) + [params.length - 1]
    pos += 8 * levels_count
    reindex = np.frombuffer(data, dtype=np.uint32, count=n, offset=pos)
    pos += 4 * n
    links = data[pos:pos + total_links_bytes]
    pos += total_links_bytes
    offsets = OffsetReader(params, data[pos:])
    bits_per_unsorted = max(MIN_BITS_PER_VALUE, packed_bits(max(n - 1, 0)))

    def point_level(pid):   # view.rs:150-166
        r = int(reindex[pid])
        for level in range(len(level_offsets) - 2):
            a, b = level_offsets[level + 1], level_offsets[level + 2]
            if r >= b - a:
                return level
        return len(level_offsets) - 2

    levels = np.zeros(n, dtype=np.int32)
    lists = []
    for pid in range(n):
        lv = point_level(pid)
        levels[pid] = lv
        for level in range(lv + 1):
            idx = pid if level == 0 else level_offsets[level] + int(reindex[pid])
            lo, hi = offsets.get(idx), offsets.get(idx + 1)
            lists.append(np.array(unpack_links(links[lo:hi], bits_per_unsorted, m0 if level == 0 else m), dtype=np.uint32))
    return levels, lists, int(m), int(m0)


# ---- graph_data: bincode 2, standard configuration (little endian, variable-length integers) -------------------------------
def _varint(x):
    x = int(x)
    if x < 251:
        return bytes([x])
    if x < 1 << 16:
        return b"\xfb" + struct.pack("<H", x)
    if x < 1 << 32:
        return b"\xfc" + struct.pack("<I", x)
    return b"\xfd" + struct.pack("<Q", x)


def _read_varint(b, pos):
    t = b[pos]
    if t < 251:
        return t, pos + 1
    if t == 251:
        return struct.unpack_from("<H", b, pos + 1)[0], pos + 3
    if t == 252:
        return struct.unpack_from("<I", b, pos + 1)[0], pos + 5
    if t == 253:
        return struct.unpack_from("<Q", b, pos + 1)[0], pos + 9
    raise ValueError("graph_data: integer wider than 64 bits")


def write_graph_data(m, m0, ef_construct, entry_points, extra_entry_points=(), extra_length=2):
    """GraphLayerData (graph_layers.rs:46-52) = m, m0, ef_construct, EntryPoints { entry_points: Vec<EntryPoint { point_id,
    level }>, extra_entry_points: FixedLengthPriorityQueue { heap: BinaryHeap<Reverse<EntryPoint>> (a sequence, in the heap
    array's order), length: NonZeroUsize } } — HNSWIndex::build creates the queue with length 2 (hnsw.rs:150)."""
    out = _varint(m) + _varint(m0) + _varint(ef_construct) + _varint(len(entry_points))
    for pid, level in entry_points:
        out += _varint(pid) + _varint(level)
    out += _varint(len(extra_entry_points))
    for pid, level in extra_entry_points:
        out += _varint(pid) + _varint(level)
    return out + _varint(extra_length)


def read_graph_data(data):
    b = bytes(data)
    pos = 0
    vals = []
    for _ in range(3):
        v, pos = _read_varint(b, pos)
        vals.append(v)
    out = {"m": vals[0], "m0": vals[1], "ef_construct": vals[2], "entry_points": [], "extra_entry_points": []}
    for key in ("entry_points", "extra_entry_points"):
        cnt, pos = _read_varint(b, pos)
        for _ in range(cnt):
            pid, pos = _read_varint(b, pos)
            lv, pos = _read_varint(b, pos)
            out[key].append((pid, lv))
    out["extra_length"], pos = _read_varint(b, pos)
    return out


def entry_point_of(graph_data):
    """EntryPoints::get_entry_point without a filter (entry_points.rs:112-127): the first main entry point, else the extra one
    of the highest level (Iterator::max_by_key: the last of equals)"""
    if graph_data["entry_points"]:
        return graph_data["entry_points"][0]
    best = None
    for ep in graph_data["extra_entry_points"]:
        if best is None or ep[1] >= best[1]:
            best = ep
    return best


# ---- encoded_u8_meta --------------------------------------------------------------------------------------------------------
def write_encoded_meta(actual_dim, alpha, offset, multiplier, dim, count, distance):
    """serde_json of Metadata (encoded_vectors_u8.rs:45-52); `invert` as HNSWIndex::build sets it (hnsw.rs:266-275). The f32
    fields are written as the shortest decimal of their exact value (serde_json reads a number as f64 and narrows it)."""
    name = DISTANCE_NAME[distance]
    f = lambda x: float(np.float32(x))  # noqa: E731
    return json.dumps({"actual_dim": int(actual_dim), "alpha": f(alpha), "offset": f(offset), "multiplier": f(multiplier),
                       "vector_parameters": {"dim": int(dim), "count": int(count), "distance_type": name, "invert": name != "Dot"}},
                      separators=(",", ":")).encode()


def read_encoded_meta(data):
    d = json.loads(bytes(data).decode("utf-8"))
    for k in ("alpha", "offset", "multiplier"):
        d[k] = np.float32(d[k])
    return d


# ---- the index as a whole ------------------------------------------------------------------------------------------------------
def save_index(idx, distance, m, ef_construct):
    """A device index (databend_amd.device.HnswIndex) -> the four Binary columns, in the order HNSWIndex::open takes them
    (hnsw.rs:69-72): graph_links, graph_data, encoded_u8_meta, encoded_u8_data."""
    levels, lists, ep, el = idx.export_graph()
    alpha, offset, mult, adim = idx.meta()
    links = write_graph_links(levels, lists, m, 2 * m)
    gdata = write_graph_data(m, 2 * m, ef_construct, [(int(ep), int(el))] if idx.n else [])
    meta = write_encoded_meta(adim, alpha, offset, mult, idx.dim, idx.n, distance)
    return links, gdata, meta, idx.encoded().tobytes()


def open_index(metric, distance, dim, count, columns):
    """HNSWIndex::open (hnsw.rs:62-98): the four Binary columns -> a device index (dbhip_hnsw_open). `metric`: the C-ABI's
    metric code for `distance`."""
    from . import device as D
    links, gdata, meta, edata = columns
    levels, lists, m, m0 = read_graph_links(links)
    g = read_graph_data(gdata)
    md = read_encoded_meta(meta)
    if len(levels) != count or md["vector_parameters"]["count"] != count or md["vector_parameters"]["dim"] != dim:
        raise ValueError("hnsw index: the columns disagree about the number of vectors / the dimension")
    if m0 != 2 * m or g["m"] != m:
        raise ValueError("hnsw index: m / m0 of the link file and the graph data disagree (the device index needs m0 = 2 m)")
    rec = 4 + md["actual_dim"]
    if len(edata) != rec * count:
        raise ValueError("hnsw index: encoded data has the wrong size")
    ep = entry_point_of(g) if count else (0, 0)
    return D.HnswIndex.open(metric, np.frombuffer(bytes(edata), dtype=np.uint8), md["alpha"], md["offset"], md["multiplier"], count, dim, m,
                            levels, lists, ep[0], ep[1])
