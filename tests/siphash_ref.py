"""An independent Python statement of siphash64 and the scatter indices (test infrastructure): SipHash-1-3 with zero keys over the
bytes DFHash feeds (scalars/hash.rs:436-545; decimals scalars/decimal/src/hash.rs:144-160), and flight_scatter_hash.rs:133-233."""
import struct

M = (1 << 64) - 1


def _rotl(x, b):
    return ((x << b) | (x >> (64 - b))) & M


def siphash13(data):
    v = [0x736f6d6570736575, 0x646f72616e646f6d, 0x6c7967656e657261, 0x7465646279746573]

    def rnd():
        v[0] = (v[0] + v[1]) & M; v[1] = _rotl(v[1], 13); v[1] ^= v[0]; v[0] = _rotl(v[0], 32)
        v[2] = (v[2] + v[3]) & M; v[3] = _rotl(v[3], 16); v[3] ^= v[2]
        v[0] = (v[0] + v[3]) & M; v[3] = _rotl(v[3], 21); v[3] ^= v[0]
        v[2] = (v[2] + v[1]) & M; v[1] = _rotl(v[1], 17); v[1] ^= v[2]; v[2] = _rotl(v[2], 32)

    n = len(data)
    for i in range(0, n - n % 8, 8):
        m = struct.unpack_from("<Q", data, i)[0]
        v[3] ^= m; rnd(); v[0] ^= m
    b = (n & 0xFF) << 56
    for i, c in enumerate(data[n - n % 8:]):
        b |= c << (8 * i)
    v[3] ^= b; rnd(); v[0] ^= b
    v[2] ^= 0xFF
    rnd(); rnd(); rnd()
    return v[0] ^ v[1] ^ v[2] ^ v[3]


def value_bytes(kind, value, scale=0):
    """the bytes DFHash writes for one value"""
    if kind == "bytes":
        return bytes(value)
    if kind == "string":
        return value if isinstance(value, bytes) else value.encode("utf-8")
    if kind == "bool":
        return b"\x01" if value else b"\x00"
    if kind.startswith("decimal"):
        return bytes([scale]) + int(value).to_bytes(16, "little", signed=True)
    fmt = {"i8": "<b", "u8": "<B", "i16": "<h", "u16": "<H", "i32": "<i", "u32": "<I", "date": "<i", "f32": "<f", "i64": "<q", "u64": "<Q",
           "timestamp": "<q", "f64": "<d"}[kind]
    return struct.pack(fmt, value)


def siphash64(kind, value, scale=0):
    return siphash13(value_bytes(kind, value, scale))


def scatter_index(hashes, scatter_size, default_index=0):
    """hashes: one entry per key, None = NULL key"""
    if len(hashes) == 1:
        return default_index if hashes[0] is None else hashes[0] % scatter_size
    return siphash13(b"".join(struct.pack("<Q", 0 if h is None else h) for h in hashes)) % scatter_size
