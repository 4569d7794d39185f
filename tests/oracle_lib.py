"""ctypes binding of oracle/liboracle.so — the CPU checker (tests only)."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ODIR = os.path.join(ROOT, "oracle")
_LIB = None


class OCol(C.Structure):
    _fields_ = [("type", C.c_int32), ("is_scalar", C.c_int32), ("data", C.c_void_p), ("validity", C.c_void_p),
                ("validity_offset", C.c_int64), ("buffers", C.c_void_p), ("n_buffers", C.c_int32),
                ("precision", C.c_uint8), ("scale", C.c_uint8), ("_pad", C.c_uint8 * 2)]


class OAgg(C.Structure):
    _fields_ = [("kind", C.c_int32), ("arg_type", C.c_int32), ("arg_precision", C.c_uint8), ("arg_scale", C.c_uint8),
                ("arg_nullable", C.c_uint8), ("_pad", C.c_uint8)]


class Q1Result(C.Structure):
    _fields_ = [("returnflag", (C.c_uint8 * 16) * 64), ("linestatus", (C.c_uint8 * 16) * 64),
                ("sum_qty", C.c_int64 * 64), ("sum_price", C.c_int64 * 64), ("sum_disc", C.c_int64 * 64),
                ("sum_disc_price", (C.c_uint64 * 2) * 64), ("sum_charge", (C.c_uint64 * 2) * 64),
                ("count", C.c_uint64 * 64)]


def load():
    global _LIB
    if _LIB is not None:
        return _LIB
    so = os.path.join(ODIR, "liboracle.so")
    srcs = [os.path.join(ODIR, f) for f in ("oracle.c", "decimal256.c", "kmeans_oracle.c", "hnsw_oracle.c", "parquet_oracle.c", "q1_typed.c")]
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(f) for f in srcs):
        subprocess.check_call(["make", "-C", ODIR, "liboracle.so"], stdout=subprocess.DEVNULL)
    L = C.CDLL(so)
    L.orc_agg_hash_bytes.restype = C.c_uint64
    L.orc_agg_hash_u64.restype = C.c_uint64
    L.orc_agg_hash_u64.argtypes = [C.c_uint64]
    L.orc_filter_select.restype = C.c_int64
    L.orc_sum_a_plus_b_mul_c_i64.restype = C.c_int64
    L.orc_hashagg_create.restype = C.c_void_p
    L.orc_hashagg_num_groups.restype = C.c_int64
    L.orc_join_inner_u64.restype = C.c_int64
    _LIB = L
    return L


class HostCol:
    """numpy-backed column for the oracle."""

    def __init__(self, dtype, arr, validity=None, precision=0, scale=0, is_scalar=False, buffers=None):
        self.dtype = dtype
        self.arr = np.ascontiguousarray(arr)
        self.validity = None
        if validity is not None:
            v = np.packbits(np.asarray(validity, dtype=bool), bitorder="little")
            self.validity = np.concatenate([v, np.zeros(8, np.uint8)])
        self.precision, self.scale, self.is_scalar = precision, scale, is_scalar
        self.buffers = buffers  # list of numpy uint8 arrays
        self._bufptr = None
        if buffers:
            self._bufptr = (C.c_void_p * len(buffers))(*[b.ctypes.data for b in buffers])

    def c(self):
        o = OCol()
        o.type = self.dtype
        o.is_scalar = 1 if self.is_scalar else 0
        o.data = self.arr.ctypes.data
        o.validity = self.validity.ctypes.data if self.validity is not None else None
        o.validity_offset = 0
        o.buffers = C.cast(self._bufptr, C.c_void_p) if self._bufptr is not None else None
        o.n_buffers = len(self.buffers) if self.buffers else 0
        o.precision, o.scale = self.precision, self.scale
        return o


def cols(hcols):
    return (OCol * len(hcols))(*[h.c() for h in hcols])


def i128_array(ints):
    out = np.zeros((len(ints), 2), dtype=np.uint64)
    for i, v in enumerate(ints):
        v = int(v) & ((1 << 128) - 1)
        out[i, 0], out[i, 1] = v & 0xFFFFFFFFFFFFFFFF, v >> 64
    return out.reshape(-1)


def i128_list(raw):
    w = np.ascontiguousarray(raw).view(np.uint64).reshape(-1, 2)
    out = []
    for lo, hi in w:
        v = (int(hi) << 64) | int(lo)
        out.append(v - (1 << 128) if v >> 127 else v)
    return out


def q1_run(host, cutoff, threads=1, block_rows=65536, n=None, typed=False):
    """Runs the reference-shaped CPU Q1 (filter -> take -> maps -> partial/final hash-agg): through the generic restatements
    (the checker), or typed=True through the type-specialised twin of the same pipeline (oracle/q1_typed.c: bench.py's baseline)."""
    L = load()
    n = len(host["l_quantity"]) if n is None else n
    res = Q1Result()
    g = (L.orc_q1_run_typed if typed else L.orc_q1_run)(host["l_quantity"].ctypes.data_as(C.c_void_p), host["l_extendedprice"].ctypes.data_as(C.c_void_p),
                     host["l_discount"].ctypes.data_as(C.c_void_p), host["l_tax"].ctypes.data_as(C.c_void_p),
                     host["l_returnflag"].ctypes.data_as(C.c_void_p), host["l_linestatus"].ctypes.data_as(C.c_void_p),
                     host["l_shipdate"].ctypes.data_as(C.c_void_p), C.c_int32(cutoff), C.c_int64(n), C.c_int(threads),
                     C.c_int64(block_rows), C.byref(res))
    if g == -105:
        raise OverflowError("Decimal overflow in the reference-shaped CPU path (aggregate_sum.rs:203-216)")
    assert g >= 0, f"oracle q1 failed: {g}"
    out = {}
    for i in range(g):
        rf = bytes(res.returnflag[i])
        ls = bytes(res.linestatus[i])
        rfk = rf[4:4 + int.from_bytes(rf[0:4], "little")]
        lsk = ls[4:4 + int.from_bytes(ls[0:4], "little")]

        def i128(pair):
            v = (int(pair[1]) << 64) | int(pair[0])
            return v - (1 << 128) if v >> 127 else v
        out[(rfk, lsk)] = dict(sum_qty=res.sum_qty[i], sum_base_price=res.sum_price[i], sum_disc_price=i128(res.sum_disc_price[i]),
                               sum_charge=i128(res.sum_charge[i]), sum_disc=res.sum_disc[i], count=res.count[i])
    return out


def q3_run(host, segment, date, limit=10, threads=1, block_rows=65536, stages=None):
    """Reference-shaped CPU Q3 -> [(l_orderkey, revenue, o_orderdate, o_shippriority)] in output order."""
    L = load()
    L.orc_q3_run.restype = C.c_int64
    c, o, li = host["customer"], host["orders"], host["lineitem"]
    nc, no, nl = len(c["c_custkey"]), len(o["o_orderkey"]), len(li["l_orderkey"])
    cap = max(limit if limit > 0 else no, 1)
    ok, rev = np.zeros(cap, np.int64), np.zeros(2 * cap, np.uint64)
    od, sp = np.zeros(cap, np.int32), np.zeros(cap, np.int32)
    ng, st = C.c_int64(), (C.c_int64 * 4)()
    p = lambda a: np.ascontiguousarray(a).ctypes.data_as(C.c_void_p)  # noqa: E731
    keep = [np.ascontiguousarray(x) for x in (c["c_custkey"], c["c_mktsegment"], o["o_orderkey"], o["o_custkey"], o["o_orderdate"],
                                              o["o_shippriority"], li["l_orderkey"], li["l_extendedprice"], li["l_discount"], li["l_shipdate"])]
    m = L.orc_q3_run(p(keep[0]), p(keep[1]), C.c_int64(nc), p(keep[2]), p(keep[3]), p(keep[4]), p(keep[5]), C.c_int64(no),
                     p(keep[6]), p(keep[7]), p(keep[8]), p(keep[9]), C.c_int64(nl), segment.encode(), C.c_int32(date),
                     C.c_int64(limit), C.c_int(threads), C.c_int64(block_rows), p(ok), p(rev), p(od), p(sp), C.byref(ng), st)
    assert m >= 0, f"oracle q3 failed: {m}"
    if stages is not None:
        stages.update(customers_kept=st[0], orders_kept=st[1], orders_joined=st[2], groups=st[3])
    r = i128_list(rev[:2 * m])
    return [(int(ok[i]), r[i], int(od[i]), int(sp[i])) for i in range(m)]
