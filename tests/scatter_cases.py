"""Test infrastructure shared by the CPU and GPU scatter / concat tests: the reference's Scatter / Concat fixtures
(kernel-pass.txt, tests/golden/kernel.json) as host columns, and the oracle's statement of DataBlock::scatter / DataBlock::concat
(kernels/scatter.rs:20-66, kernels/concat.rs:62-340) over numpy columns."""
import ctypes as C

import numpy as np


def cells_to_columns(header, rows):
    """text cells of a kernel-pass.txt table -> per column (kind, values, validity): kind 'int' (Int32 values) or 'str' (bytes)"""
    cols = []
    for c in range(len(header)):
        cells = [r[c] for r in rows]
        valid = np.array([x != "NULL" for x in cells], bool)
        if any(x.startswith("'") for x in cells):
            cols.append(("str", [x.strip("'").encode() if x != "NULL" else b"" for x in cells], valid))
        else:
            cols.append(("int", np.array([int(x) if x != "NULL" else 0 for x in cells], np.int32), valid))
    return cols


def render(kind, values, validity):
    """(values, validity) -> the reference's rendered cells"""
    out = []
    for v, ok in zip(values, validity):
        if not ok:
            out.append("NULL")
        elif kind == "str":
            out.append("'" + (v.decode() if isinstance(v, bytes) else str(v)) + "'")
        else:
            out.append(str(int(v)))
    return out


def pack(bits):
    b = np.packbits(np.asarray(bits, bool), bitorder="little")
    return np.concatenate([b, np.zeros(16 - len(b) % 8, np.uint8)])


def unpack(by, n, off=0):
    return np.unpackbits(np.asarray(by, np.uint8), bitorder="little")[off:off + n].astype(bool)


def oracle_scatter(L, index, scatter_size, columns):
    """columns: list of (elem_size, values array [n, ...] | None for a Bitmap-only column, bitmaps list) -> per destination the
    taken arrays. `columns` entries: dict(values=np array or None, bits=[bool arrays])"""
    index = np.ascontiguousarray(index, np.uint32)
    n = len(index)
    rows = np.zeros(max(n, 1), np.uint32)
    starts = np.zeros(scatter_size + 1, np.int64)
    L.orc_divide_indices(index.ctypes.data_as(C.c_void_p), C.c_int64(n), C.c_int(scatter_size), rows.ctypes.data_as(C.c_void_p),
                         starts.ctypes.data_as(C.c_void_p))
    out = []
    for d in range(scatter_size):
        sel = np.ascontiguousarray(rows[starts[d]:starts[d + 1]])
        k = len(sel)
        blk = []
        for col in columns:
            o = {}
            if col.get("values") is not None:
                v = np.ascontiguousarray(col["values"])
                es = v.dtype.itemsize * (int(np.prod(v.shape[1:])) if v.ndim > 1 else 1)
                res = np.zeros((max(k, 1),) + v.shape[1:], v.dtype)
                L.orc_take(v.ctypes.data_as(C.c_void_p), C.c_int(es), sel.ctypes.data_as(C.c_void_p), C.c_int64(k), res.ctypes.data_as(C.c_void_p))
                o["values"] = res[:k]
            o["bits"] = []
            for bits in col.get("bits", []):
                src = pack(bits)
                dst = np.zeros((k + 7) // 8 + 8, np.uint8)
                L.orc_take_bitmap(src.ctypes.data_as(C.c_void_p), C.c_int64(0), sel.ctypes.data_as(C.c_void_p), C.c_int64(k),
                                  dst.ctypes.data_as(C.c_void_p), C.c_int64(0))
                o["bits"].append(unpack(dst, k))
            blk.append(o)
        out.append(blk)
    return out, starts.tolist(), rows[:n]


def oracle_concat_fixed(L, blocks):
    blocks = [np.ascontiguousarray(b) for b in blocks]
    es = blocks[0].dtype.itemsize * (int(np.prod(blocks[0].shape[1:])) if blocks[0].ndim > 1 else 1)
    rows = np.array([len(b) for b in blocks], np.int64)
    out = np.zeros((max(int(rows.sum()), 1),) + blocks[0].shape[1:], blocks[0].dtype)
    ptrs = (C.c_void_p * len(blocks))(*[b.ctypes.data for b in blocks])
    L.orc_concat_fixed(ptrs, rows.ctypes.data_as(C.c_void_p), len(blocks), es, out.ctypes.data_as(C.c_void_p))
    return out[:int(rows.sum())]


def oracle_concat_bits(L, blocks, rows):
    """blocks: bool arrays or None (no validity: all valid)"""
    packed = [pack(b) if b is not None else None for b in blocks]
    rows = np.array(rows, np.int64)
    total = int(rows.sum())
    out = np.zeros((total + 7) // 8 + 8, np.uint8)
    ptrs = (C.c_void_p * len(blocks))(*[p.ctypes.data if p is not None else None for p in packed])
    L.orc_concat_bitmap(ptrs, None, rows.ctypes.data_as(C.c_void_p), len(blocks), out.ctypes.data_as(C.c_void_p))
    return unpack(out, total)
