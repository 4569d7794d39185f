"""Python host mirror of the reference's column vocabulary on top of the C-ABI.

Column / DataBlock here are HBM-resident (the device analogue of
src/query/expression/src/values.rs:192 `Column` and block.rs:49-59 `DataBlock`);
every operation goes through libdbhip.so — there is no numpy compute path.
"""
import ctypes as C

import numpy as np

from . import _lib as L
from ._lib import AggDesc, Col, check, lib

NP_OF = {
    L.T_I8: np.int8, L.T_I16: np.int16, L.T_I32: np.int32, L.T_I64: np.int64,
    L.T_U8: np.uint8, L.T_U16: np.uint16, L.T_U32: np.uint32, L.T_U64: np.uint64,
    L.T_F32: np.float32, L.T_F64: np.float64, L.T_DATE: np.int32, L.T_TIMESTAMP: np.int64,
    L.T_DEC64: np.int64,
}
TYPE_OF_NP = {np.dtype(v): k for k, v in NP_OF.items() if k not in (L.T_DATE, L.T_TIMESTAMP, L.T_DEC64)}
ELEM_SIZE = {L.T_DEC128: 16, L.T_STRING: 16, L.T_DEC256: 32}
for _t, _d in NP_OF.items():
    ELEM_SIZE[_t] = np.dtype(_d).itemsize

_initialised = False


def init(device=0):
    global _initialised
    check(lib().dbhip_init(int(device)))
    _initialised = True


def _ensure():
    if not _initialised:
        init(0)


class DeviceBuffer:
    """Owned HBM allocation (dbhip_alloc / dbhip_free)."""

    def __init__(self, nbytes):
        _ensure()
        self.nbytes = int(nbytes)
        p = C.c_void_p()
        check(lib().dbhip_alloc(C.c_size_t(max(self.nbytes, 16)), C.byref(p)))
        self.ptr = p.value

    @classmethod
    def from_numpy(cls, arr):
        arr = np.ascontiguousarray(arr)
        b = cls(arr.nbytes)
        if arr.nbytes:
            check(lib().dbhip_memcpy_h2d(C.c_void_p(b.ptr), arr.ctypes.data_as(C.c_void_p), C.c_size_t(arr.nbytes), None))
        return b

    def to_numpy(self, dtype, count=None):
        dtype = np.dtype(dtype)
        n = self.nbytes // dtype.itemsize if count is None else int(count)
        out = np.empty(n, dtype=dtype)
        if n:
            check(lib().dbhip_memcpy_d2h(out.ctypes.data_as(C.c_void_p), C.c_void_p(self.ptr), C.c_size_t(n * dtype.itemsize), None))
        return out

    def zero(self):
        check(lib().dbhip_memset(C.c_void_p(self.ptr), 0, C.c_size_t(max(self.nbytes, 1)), None))
        return self

    def free(self):
        if getattr(self, "ptr", None):
            lib().dbhip_free(C.c_void_p(self.ptr))
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class BorrowedBuffer:
    """Non-owning view of HBM some other allocator of the same HIP context owns (e.g. a torch tensor's storage): the
    DeviceBuffer surface (`ptr`, `nbytes`, `to_numpy`) without dbhip_alloc / dbhip_free. `keep` pins the owner."""

    def __init__(self, ptr, nbytes, keep=None):
        self.ptr, self.nbytes, self._keep = int(ptr), int(nbytes), keep

    @classmethod
    def of_tensor(cls, t):
        return cls(t.data_ptr(), t.numel() * t.element_size(), keep=t)

    to_numpy = DeviceBuffer.to_numpy

    def free(self):
        self._keep = None


def pack_bits(bools):
    """bool array -> LSB-first Bitmap bytes (padded to a multiple of 8 bytes)."""
    bools = np.asarray(bools, dtype=bool)
    by = np.packbits(bools, bitorder="little")
    pad = (-len(by)) % 8
    if pad or len(by) == 0:
        by = np.concatenate([by, np.zeros(pad if len(by) else 8, np.uint8)])
    return by


def unpack_bits(by, n):
    return np.unpackbits(np.asarray(by, dtype=np.uint8), bitorder="little")[:n].astype(bool)


def make_views(strings):
    """list of bytes (each <= 12 bytes) -> (n,16) uint8 inline BinaryView array
    (src/common/column/src/binview/view.rs:30-42)."""
    n = len(strings)
    v = np.zeros((n, 16), dtype=np.uint8)
    for i, s in enumerate(strings):
        assert len(s) <= 12, "only inline views here"
        v[i, 0:4] = np.frombuffer(np.uint32(len(s)).tobytes(), np.uint8)
        v[i, 4:4 + len(s)] = np.frombuffer(s, np.uint8)
    return v


def make_views_general(strings):
    """Any-length strings -> (views (n,16) u8, data buffer u8): long strings go to buffer 0."""
    n = len(strings)
    v = np.zeros((n, 16), dtype=np.uint8)
    buf = bytearray()
    for i, s in enumerate(strings):
        v[i, 0:4] = np.frombuffer(np.uint32(len(s)).tobytes(), np.uint8)
        if len(s) <= 12:
            v[i, 4:4 + len(s)] = np.frombuffer(s, np.uint8)
        else:
            v[i, 4:8] = np.frombuffer(s[:4], np.uint8)
            v[i, 8:12] = np.frombuffer(np.uint32(0).tobytes(), np.uint8)
            v[i, 12:16] = np.frombuffer(np.uint32(len(buf)).tobytes(), np.uint8)
            buf += s
    return v, np.frombuffer(bytes(buf) + b"\0" * 16, dtype=np.uint8).copy()


def view_strings(views, buffer0=None):
    """16-byte views -> bytes; long views ({len, prefix, buffer 0, offset}) read from `buffer0` (numpy uint8)"""
    views = np.asarray(views, dtype=np.uint8).reshape(-1, 16)
    out = []
    for r in views:
        ln = int(np.frombuffer(r[0:4].tobytes(), np.uint32)[0])
        if ln <= 12:
            out.append(bytes(r[4:4 + ln]))
        else:
            off = int(np.frombuffer(r[12:16].tobytes(), np.uint32)[0])
            out.append(bytes(buffer0[off:off + ln]))
    return out


class Column:
    """HBM-resident column: values + optional validity Bitmap (NullableColumn)."""

    def __init__(self, dtype, n, data, validity=None, precision=0, scale=0, is_scalar=False, buffers=None, keep=()):
        self.dtype, self.n, self.data, self.validity = dtype, int(n), data, validity
        self.precision, self.scale, self.is_scalar = int(precision), int(scale), bool(is_scalar)
        self.buffers = buffers  # DeviceBuffer holding an array of device pointers (strings)
        self.n_buffers = 1 if buffers is not None else 0
        self.voff = 0           # bit offset of the validity Bitmap (a sliced Bitmap, bitmap/immutable.rs:78-85)
        self.boff = 0           # bit offset of a Boolean column's VALUES (only dbhip_concat_columns reads sliced Boolean values)
        self._keep = keep

    # ---- constructors -----------------------------------------------------------------
    @classmethod
    def from_numpy(cls, arr, dtype=None, validity=None, precision=0, scale=0):
        arr = np.ascontiguousarray(arr)
        if dtype is None:
            dtype = TYPE_OF_NP[arr.dtype]
        n = len(arr)
        vb = DeviceBuffer.from_numpy(pack_bits(validity)) if validity is not None else None
        return cls(dtype, n, DeviceBuffer.from_numpy(arr), vb, precision, scale)

    @classmethod
    def scalar(cls, value, dtype, precision=0, scale=0):
        if dtype == L.T_DEC128:
            arr = i128_to_bytes([int(value)])
        elif dtype == L.T_DEC256:
            arr = ints_to_limbs([int(value)], 256)
        else:
            arr = np.array([value], dtype=NP_OF[dtype])
        return cls(dtype, 1, DeviceBuffer.from_numpy(arr), None, precision, scale, is_scalar=True)

    @classmethod
    def decimal128(cls, ints, precision, scale, validity=None):
        vb = DeviceBuffer.from_numpy(pack_bits(validity)) if validity is not None else None
        return cls(L.T_DEC128, len(ints), DeviceBuffer.from_numpy(i128_to_bytes(ints)), vb, precision, scale)

    @classmethod
    def decimal256(cls, ints, precision, scale, validity=None):
        vb = DeviceBuffer.from_numpy(pack_bits(validity)) if validity is not None else None
        return cls(L.T_DEC256, len(ints), DeviceBuffer.from_numpy(ints_to_limbs(ints, 256)), vb, precision, scale)

    @classmethod
    def decimal(cls, ints, precision, scale, validity=None, bits=None):
        """a decimal column in the storage class of its precision (or an explicitly wider one: legacy columns)"""
        bits = bits or (64 if precision <= 18 else (128 if precision <= 38 else 256))
        if bits == 64:
            return cls.from_numpy(np.array(ints, dtype=np.int64), L.T_DEC64, validity, precision, scale)
        return cls.decimal128(ints, precision, scale, validity) if bits == 128 else cls.decimal256(ints, precision, scale, validity)

    @classmethod
    def boolean(cls, bools, validity=None):
        vb = DeviceBuffer.from_numpy(pack_bits(validity)) if validity is not None else None
        return cls(L.T_BOOL, len(bools), DeviceBuffer.from_numpy(pack_bits(bools)), vb)

    @classmethod
    def strings(cls, strings, validity=None):
        views, buf = make_views_general(strings)
        dbuf = DeviceBuffer.from_numpy(buf)
        ptrs = DeviceBuffer.from_numpy(np.array([dbuf.ptr], dtype=np.uint64))
        vb = DeviceBuffer.from_numpy(pack_bits(validity)) if validity is not None else None
        return cls(L.T_STRING, len(strings), DeviceBuffer.from_numpy(views), vb, buffers=ptrs, keep=(dbuf,))

    @classmethod
    def from_views(cls, views):
        views = np.ascontiguousarray(views, dtype=np.uint8).reshape(-1, 16)
        return cls(L.T_STRING, len(views), DeviceBuffer.from_numpy(views))

    # ---- access ---------------------------------------------------------------------------
    def c(self):
        col = Col()
        col.type = self.dtype
        col.is_scalar = 1 if self.is_scalar else 0
        col.data = self.data.ptr
        col.validity = self.validity.ptr if self.validity is not None else None
        col.validity_offset = self.voff
        col.buffers = self.buffers.ptr if self.buffers is not None else None
        col.n_buffers = self.n_buffers
        col.precision, col.scale = self.precision, self.scale
        return col

    def to_numpy(self):
        if self.dtype == L.T_DEC128:
            return bytes_to_i128(self.data.to_numpy(np.uint8, 16 * self.n))
        if self.dtype == L.T_DEC256:
            return limbs_to_ints(self.data.to_numpy(np.uint64, 4 * self.n), 256)
        if self.dtype == L.T_BOOL:
            return unpack_bits(self.data.to_numpy(np.uint8, (self.boff + self.n + 7) // 8), self.boff + self.n)[self.boff:]
        if self.dtype == L.T_STRING:
            return self.data.to_numpy(np.uint8, 16 * self.n).reshape(-1, 16)
        return self.data.to_numpy(NP_OF[self.dtype], self.n)

    def to_strings(self):
        """the values of a String column as bytes (test / debugging aid): inline views as they are, long views through buffer 0 — the
        DeviceBuffer this Column keeps alive (Column.strings, the received columns of an exchange)"""
        assert self.dtype == L.T_STRING
        buf0 = None
        for k in self._keep:
            if isinstance(k, DeviceBuffer):
                buf0 = k.to_numpy(np.uint8, k.nbytes)
                break
        return view_strings(self.to_numpy(), buf0)

    def validity_numpy(self):
        if self.validity is None:
            return np.ones(self.n, dtype=bool)
        return unpack_bits(self.validity.to_numpy(np.uint8, (self.voff + self.n + 7) // 8), self.voff + self.n)[self.voff:]

    def slice(self, lo, hi):
        """Column::slice (values.rs): rows [lo, hi) as a view — value buffers by address, Bitmaps by bit offset"""
        assert 0 <= lo <= hi <= self.n and not self.is_scalar
        c = Column(self.dtype, hi - lo, self.data, self.validity, self.precision, self.scale, buffers=self.buffers, keep=(self,))
        c.n_buffers = self.n_buffers
        c.voff = self.voff + lo
        if self.dtype == L.T_BOOL:
            c.boff = self.boff + lo
        else:
            es = ELEM_SIZE[self.dtype]
            c.data = BorrowedBuffer(self.data.ptr + lo * es, (hi - lo) * es, keep=self.data)
        return c

    def string_values(self):
        """the values of a String column as bytes (views resolved through the column's buffer table)"""
        views = self.data.to_numpy(np.uint8, 16 * self.n).reshape(-1, 16)
        ptrs = self.buffers.to_numpy(np.uint64, self.n_buffers) if self.buffers is not None and self.n_buffers else []
        out = []
        for r in views:
            ln, bi, off = (int(x) for x in np.frombuffer(r.tobytes(), np.uint32)[[0, 2, 3]])
            if ln <= 12:
                out.append(bytes(r[4:4 + ln]))
            else:
                out.append(BorrowedBuffer(int(ptrs[bi]) + off, ln).to_numpy(np.uint8, ln).tobytes())
        return out


def ints_to_limbs(ints, bits):
    """python ints -> little-endian two's complement u64 limbs, bits / 64 per value (flat)"""
    k = bits // 64
    out = np.zeros((len(ints), k), dtype=np.uint64)
    for i, v in enumerate(ints):
        v = int(v) & ((1 << bits) - 1)
        for j in range(k):
            out[i, j] = (v >> (64 * j)) & 0xFFFFFFFFFFFFFFFF
    return out.reshape(-1)


def limbs_to_ints(raw, bits):
    k = bits // 64
    out = []
    for row in np.ascontiguousarray(raw).view(np.uint64).reshape(-1, k):
        v = sum(int(x) << (64 * j) for j, x in enumerate(row))
        out.append(v - (1 << bits) if v >> (bits - 1) else v)
    return out


def i128_to_bytes(ints):
    out = np.zeros((len(ints), 2), dtype=np.uint64)
    for i, v in enumerate(ints):
        v = int(v) & ((1 << 128) - 1)
        out[i, 0] = v & 0xFFFFFFFFFFFFFFFF
        out[i, 1] = v >> 64
    return out.reshape(-1)


def bytes_to_i128(raw):
    w = np.frombuffer(np.ascontiguousarray(raw).tobytes(), dtype=np.uint64).reshape(-1, 2)
    out = []
    for lo, hi in w:
        v = (int(hi) << 64) | int(lo)
        if v >> 127:
            v -= 1 << 128
        out.append(v)
    return out


def _cols(cols):
    arr = (Col * len(cols))(*[c.c() for c in cols])
    return arr


class RowErrors:
    """Per-row error bitmap + count (EvalContext errors, function.rs:534-556)."""

    def __init__(self, n):
        self.n = n
        self.bitmap = DeviceBuffer(((n + 31) // 32) * 4 + 8)
        self.count = DeviceBuffer(8).zero()

    def error_rows(self):
        ok = unpack_bits(self.bitmap.to_numpy(np.uint8, ((self.n + 31) // 32) * 4), self.n)
        return np.nonzero(~ok)[0]

    def num_errors(self):
        return int(self.count.to_numpy(np.uint64, 1)[0])


def _merged_validity(a, b, n):
    if a.validity is None and b.validity is None:
        return None
    if a.validity is None or (a.is_scalar and a.validity_numpy()[0]):
        va = None
    else:
        va = a
    # validity = AND of inputs (passthrough_nullable, register_vectorize.rs:447-471), on device
    bits = []
    for x in (a, b):
        if x.validity is None:
            continue
        if x.is_scalar:
            bits.append(DeviceBuffer.from_numpy(pack_bits(np.repeat(x.validity_numpy()[:1], n))))
        else:
            bits.append(x.validity)
    if len(bits) == 1:
        return bits[0]
    out = DeviceBuffer(((n + 63) // 64) * 8)
    check(lib().dbhip_bitmap_binary(0, C.c_void_p(bits[0].ptr), C.c_void_p(bits[1].ptr), C.c_int64(n), C.c_void_p(out.ptr), None))
    return out


def arith(op, a, b, n=None, errors=None):
    """plus/minus/multiply/divide/div/modulo on numeric columns -> Column."""
    n = n if n is not None else max(a.n if not a.is_scalar else 0, b.n if not b.is_scalar else 0)
    out_t = lib().dbhip_arith_result_type(op, a.dtype, b.dtype)
    if out_t < 0:
        raise L.DbhipError(L.ERR_INVALID, f"no numeric overload for op {op} on ({a.dtype},{b.dtype})")
    out = DeviceBuffer(max(n, 1) * ELEM_SIZE[out_t] + 64)
    ca, cb = a.c(), b.c()
    eb = C.c_void_p(errors.bitmap.ptr) if errors else None
    ec = C.c_void_p(errors.count.ptr) if errors else None
    check(lib().dbhip_arith(op, C.byref(ca), C.byref(cb), C.c_int64(n), out_t, C.c_void_p(out.ptr), eb, ec, None))
    return Column(out_t, n, out, _merged_validity(a, b, n))


def cast(col, dst_type, is_try=False, rounding_mode=True, n=None):
    """to_<number> / try_to_<number> (dbhip_cast). -> (Column, bool[n] ok-rows, error count): for CAST `ok` marks the rows that did
    not raise "number overflowed"; for TRY_CAST it is the result's validity (also attached to the Column)."""
    n = n if n is not None else col.n
    out = DeviceBuffer(max(n, 1) * ELEM_SIZE[dst_type] + 64)
    bm = DeviceBuffer(((max(n, 1) + 63) // 64) * 8 + 8)
    cnt = DeviceBuffer(8)
    cnt.zero()
    cc = col.c()
    check(lib().dbhip_cast(C.byref(cc), dst_type, int(is_try), int(rounding_mode), C.c_int64(n), C.c_void_p(out.ptr), C.c_void_p(bm.ptr),
                           C.c_void_p(cnt.ptr), None))
    ok = unpack_bits(bm.to_numpy(np.uint8, (n + 7) // 8), n) if n else np.zeros(0, dtype=bool)
    validity = bm if is_try else col.validity
    return Column(dst_type, n, out, validity, keep=(col,)), ok, int(cnt.to_numpy(np.uint64, 1)[0])


def decimal_result_size(op, a, b):
    props = {L.T_I8: (3, 0), L.T_U8: (3, 0), L.T_I16: (5, 0), L.T_U16: (5, 0), L.T_I32: (10, 0), L.T_U32: (10, 0),
             L.T_I64: (19, 0), L.T_U64: (20, 0)}
    ap = (a.precision, a.scale) if a.dtype in (L.T_DEC64, L.T_DEC128, L.T_DEC256) else props[a.dtype]
    bp = (b.precision, b.scale) if b.dtype in (L.T_DEC64, L.T_DEC128, L.T_DEC256) else props[b.dtype]
    p, s = C.c_uint8(), C.c_uint8()
    check(lib().dbhip_decimal_result_size(op, ap[0], ap[1], bp[0], bp[1], C.byref(p), C.byref(s)))
    return p.value, s.value


def decimal_arith(op, a, b, n=None, errors=None):
    n = n if n is not None else max(a.n if not a.is_scalar else 0, b.n if not b.is_scalar else 0)
    p, s = decimal_result_size(op, a, b)
    out_t = L.T_DEC64 if p <= 18 else (L.T_DEC128 if p <= 38 else L.T_DEC256)
    out = DeviceBuffer(max(n, 1) * ELEM_SIZE[out_t] + 64)
    ca, cb = a.c(), b.c()
    eb = C.c_void_p(errors.bitmap.ptr) if errors else None
    ec = C.c_void_p(errors.count.ptr) if errors else None
    check(lib().dbhip_decimal_arith(op, C.byref(ca), C.byref(cb), C.c_int64(n), out_t, p, s, C.c_void_p(out.ptr), eb, ec, None))
    return Column(out_t, n, out, _merged_validity(a, b, n), p, s)


def decimal_neg(col, n=None):
    """unary minus on a decimal column (dbhip_decimal_neg): same DecimalSize and storage class"""
    n = n if n is not None else col.n
    out = DeviceBuffer(max(n, 1) * ELEM_SIZE[col.dtype] + 64)
    cc = col.c()
    check(lib().dbhip_decimal_neg(C.byref(cc), C.c_int64(n), C.c_void_p(out.ptr), None))
    return Column(col.dtype, n, out, col.validity, col.precision, col.scale, keep=(col,))


def decimal_cast(col, precision, scale, is_try=False, rounding_mode=False, n=None):
    """to_decimal(p, s) / try_to_decimal(p, s) for decimal and integer columns (dbhip_decimal_cast) -> (Column, ok rows, error count)"""
    n = n if n is not None else col.n
    dst_type = L.T_DEC64 if precision <= 18 else (L.T_DEC128 if precision <= 38 else L.T_DEC256)
    out = DeviceBuffer(max(n, 1) * ELEM_SIZE[dst_type] + 64)
    bm = DeviceBuffer(((max(n, 1) + 63) // 64) * 8 + 8)
    cnt = DeviceBuffer(8)
    cnt.zero()
    cc = col.c()
    check(lib().dbhip_decimal_cast(C.byref(cc), dst_type, precision, scale, int(is_try), int(rounding_mode), C.c_int64(n), C.c_void_p(out.ptr),
                                   C.c_void_p(bm.ptr), C.c_void_p(cnt.ptr), None))
    ok = unpack_bits(bm.to_numpy(np.uint8, (n + 7) // 8), n) if n else np.zeros(0, dtype=bool)
    return Column(dst_type, n, out, bm if is_try else col.validity, precision, scale, keep=(col,)), ok, int(cnt.to_numpy(np.uint64, 1)[0])


def cmp(op, a, b, n=None):
    n = n if n is not None else max(a.n if not a.is_scalar else 0, b.n if not b.is_scalar else 0)
    out = DeviceBuffer(((n + 63) // 64) * 8 + 8)
    ca, cb = a.c(), b.c()
    check(lib().dbhip_cmp(op, C.byref(ca), C.byref(cb), C.c_int64(n), C.c_void_p(out.ptr), None))
    return Column(L.T_BOOL, n, out, _merged_validity(a, b, n))


def filter_select(pred):
    """Boolean column -> ascending u32 selection vector (device) and its length."""
    sel = DeviceBuffer(max(pred.n, 1) * 4 + 64)
    cnt = DeviceBuffer(8)
    check(lib().dbhip_filter_select(C.c_void_p(pred.data.ptr), C.c_int64(0), C.c_int64(pred.n), C.c_void_p(sel.ptr), C.c_void_p(cnt.ptr), None))
    k = int(cnt.to_numpy(np.uint64, 1)[0])
    return sel, k


def bitmap_count(pred, n):
    """number of set bits of a Boolean column (Bitmap::true_count)"""
    out = DeviceBuffer(8)
    check(lib().dbhip_bitmap_count(C.c_void_p(pred.data.ptr), C.c_int64(0), C.c_int64(n), C.c_void_p(out.ptr), None))
    return int(out.to_numpy(np.uint64, 1)[0])


def select_cmp(op, a, b, sel=None, n=None, want_false=False):
    """Selector leaf (filter/select_value): `a op b` evaluated on the rows of `sel` (a DeviceBuffer of u32 row ids, `n` of them) or on
    all n rows; -> (true list DeviceBuffer, n_true, false list DeviceBuffer | None). NULL rows do not pass."""
    if n is None:
        n = a.n if not a.is_scalar else b.n
    t = DeviceBuffer(max(n, 1) * 4 + 64)
    f = DeviceBuffer(max(n, 1) * 4 + 64) if want_false else None
    cnt = DeviceBuffer(8)
    ca, cb = a.c(), b.c()
    check(lib().dbhip_select_cmp(op, C.byref(ca), C.byref(cb), C.c_void_p(sel.ptr) if sel is not None else None, C.c_int64(n), C.c_void_p(t.ptr),
                                 C.c_void_p(f.ptr) if f is not None else None, C.c_void_p(cnt.ptr), None))
    return t, int(cnt.to_numpy(np.uint64, 1)[0]), f


def select_bool(pred, sel=None, n=None, want_false=False):
    """Selector leaf for a Boolean column (select_boolean_column)"""
    n = pred.n if n is None else n
    t = DeviceBuffer(max(n, 1) * 4 + 64)
    f = DeviceBuffer(max(n, 1) * 4 + 64) if want_false else None
    cnt = DeviceBuffer(8)
    cp = pred.c()
    check(lib().dbhip_select_bool(C.byref(cp), C.c_void_p(sel.ptr) if sel is not None else None, C.c_int64(n), C.c_void_p(t.ptr),
                                  C.c_void_p(f.ptr) if f is not None else None, C.c_void_p(cnt.ptr), None))
    return t, int(cnt.to_numpy(np.uint64, 1)[0]), f


def select_tree(expr, n):
    """Selector::select over a tree of ("and", [..]) / ("or", [..]) / ("cmp", op, a, b) / ("bool", column) nodes
    (filter/selector.rs:64-330: process_and narrows the TRUE list conjunct by conjunct, process_or feeds the FALSE list of one
    disjunct to the next and concatenates the true lists): -> (row ids of the rows that pass, ascending within each OR branch's
    contribution exactly like the reference's true_selection, count). Later predicates are evaluated only on the rows that are
    still undecided."""
    def run(e, sel, cnt, want_false):
        # -> (true_ids ndarray-free: DeviceBuffer, n_true, false DeviceBuffer | None)
        kind = e[0]
        if kind == "cmp":
            return select_cmp(e[1], e[2], e[3], sel, cnt, want_false)
        if kind == "bool":
            return select_bool(e[1], sel, cnt, want_false)
        if kind == "and":
            cur, k = sel, cnt
            fparts = []
            for child in e[1]:
                t, kt, f = run(child, cur, k, want_false)
                if want_false and k - kt:
                    fparts.append((f, k - kt))
                cur, k = t, kt
                if k == 0:
                    break
            if cur is None:   # no conjunct at all: everything passes
                cur = DeviceBuffer.from_numpy(np.arange(cnt, dtype=np.uint32))
            return cur, k, (_concat_u32(fparts) if want_false else None)
        if kind == "or":
            cur, k = sel, cnt
            tparts = []
            f = None
            for child in e[1]:
                t, kt, f = run(child, cur, k, True)
                if kt:
                    tparts.append((t, kt))
                cur, k = f, k - kt
                if k == 0:
                    break
            tt = _concat_u32(tparts)
            return tt, sum(c for _, c in tparts), (cur if want_false else None)
        raise ValueError(kind)

    t, k, _ = run(expr, None, n, False)
    return t, k


def _concat_u32(parts):
    total = sum(c for _, c in parts)
    out = DeviceBuffer(max(total, 1) * 4 + 64)
    off = 0
    for buf, c in parts:
        if c:
            check(lib().dbhip_memcpy_d2d(C.c_void_p(out.ptr + off * 4), C.c_void_p(buf.ptr), C.c_size_t(c * 4), None))
        off += c
    return out


def take(col, sel, k):
    """DataBlock::take for one column (kernels/take.rs:43)."""
    if col.dtype == L.T_BOOL:
        out = DeviceBuffer(((k + 63) // 64) * 8 + 8)
        check(lib().dbhip_take_bitmap(C.c_void_p(col.data.ptr), C.c_int64(0), C.c_void_p(sel.ptr), C.c_int64(k), C.c_void_p(out.ptr), None))
    else:
        es = ELEM_SIZE[col.dtype]
        out = DeviceBuffer(max(k, 1) * es + 64)
        check(lib().dbhip_take(C.c_void_p(col.data.ptr), es, C.c_void_p(sel.ptr), C.c_int64(k), C.c_void_p(out.ptr), None))
    vb = None
    if col.validity is not None:
        vb = DeviceBuffer(((k + 63) // 64) * 8 + 8)
        check(lib().dbhip_take_bitmap(C.c_void_p(col.validity.ptr), C.c_int64(0), C.c_void_p(sel.ptr), C.c_int64(k), C.c_void_p(vb.ptr), None))
    return Column(col.dtype, k, out, vb, col.precision, col.scale, buffers=col.buffers, keep=(col,))


def take_block(cols, sel, k):
    """DataBlock::take over several columns with one selection: ONE launch for the value buffers (dbhip_take_block); Bitmap
    columns and validities go through dbhip_take_bitmap. -> list of Columns"""
    plain = [c for c in cols if c.dtype != L.T_BOOL]
    outs = {}
    for g0 in range(0, len(plain), 8):
        grp = plain[g0:g0 + 8]
        bufs = [DeviceBuffer(max(k, 1) * ELEM_SIZE[c.dtype] + 64) for c in grp]
        srcs = (C.c_void_p * len(grp))(*[c.data.ptr for c in grp])
        dsts = (C.c_void_p * len(grp))(*[b.ptr for b in bufs])
        es = (C.c_int32 * len(grp))(*[ELEM_SIZE[c.dtype] for c in grp])
        check(lib().dbhip_take_block(srcs, es, len(grp), C.c_void_p(sel.ptr), C.c_int64(k), dsts, None))
        for c, b in zip(grp, bufs):
            outs[id(c)] = b
    res = []
    for c in cols:
        if c.dtype == L.T_BOOL:
            res.append(take(c, sel, k))
            continue
        vb = None
        if c.validity is not None:
            vb = DeviceBuffer(((k + 63) // 64) * 8 + 8)
            check(lib().dbhip_take_bitmap(C.c_void_p(c.validity.ptr), C.c_int64(0), C.c_void_p(sel.ptr), C.c_int64(k), C.c_void_p(vb.ptr), None))
        res.append(Column(c.dtype, k, outs[id(c)], vb, c.precision, c.scale, buffers=c.buffers, keep=(c,)))
    return res


def sel_from_ranges(ranges, num_rows):
    """DataBlock::take_ranges as a device selection vector: ranges = [(start, end), ...]"""
    r = np.ascontiguousarray(np.array(ranges, dtype=np.uint32).reshape(-1, 2))
    sel = DeviceBuffer(max(num_rows, 1) * 4 + 64)
    check(lib().dbhip_sel_from_ranges(r.ctypes.data_as(C.c_void_p), C.c_int32(len(r)), C.c_void_p(sel.ptr), C.c_int64(num_rows), None))
    return sel


def sel_from_repeats(repeats, num_rows):
    """DataBlock::take_compacted_indices as a device selection vector: repeats = [(row, count), ...]"""
    r = np.ascontiguousarray(np.array(repeats, dtype=np.uint32).reshape(-1, 2))
    sel = DeviceBuffer(max(num_rows, 1) * 4 + 64)
    check(lib().dbhip_sel_from_repeats(r.ctypes.data_as(C.c_void_p), C.c_int32(len(r)), C.c_void_p(sel.ptr), C.c_int64(num_rows), None))
    return sel


def take_chunks(cols, pairs):
    """DataBlock::take_blocks for one column of several blocks: pairs = [(block, row), ...] -> Column"""
    pr = np.ascontiguousarray(np.array(pairs, dtype=np.uint32).reshape(-1, 2))
    n = len(pr)
    dp = DeviceBuffer.from_numpy(pr.reshape(-1)) if n else DeviceBuffer(16)
    c0 = cols[0]
    ptrs = (C.c_void_p * len(cols))(*[c.data.ptr for c in cols])
    if c0.dtype == L.T_BOOL:
        out = DeviceBuffer(((n + 63) // 64) * 8 + 8)
        check(lib().dbhip_take_chunks(ptrs, len(cols), 0, C.c_void_p(dp.ptr), C.c_int64(n), C.c_void_p(out.ptr), None))
    else:
        es = ELEM_SIZE[c0.dtype]
        out = DeviceBuffer(max(n, 1) * es + 64)
        check(lib().dbhip_take_chunks(ptrs, len(cols), es, C.c_void_p(dp.ptr), C.c_int64(n), C.c_void_p(out.ptr), None))
    vb = None
    if any(c.validity is not None for c in cols):
        vptrs = (C.c_void_p * len(cols))(*[(c.validity.ptr if c.validity is not None else None) for c in cols])
        vb = DeviceBuffer(((n + 63) // 64) * 8 + 8)
        check(lib().dbhip_take_chunks(vptrs, len(cols), 0, C.c_void_p(dp.ptr), C.c_int64(n), C.c_void_p(vb.ptr), None))
    return Column(c0.dtype, n, out, vb, c0.precision, c0.scale, buffers=c0.buffers, keep=tuple(cols))


def group_hash(cols, n):
    out = DeviceBuffer(max(n, 1) * 8)
    arr = _cols(cols)
    check(lib().dbhip_group_hash(arr, len(cols), C.c_int64(n), C.c_void_p(out.ptr), None))
    return out.to_numpy(np.uint64, n)


def sum_a_plus_b_mul_c(a, b, c):
    out = DeviceBuffer(8).zero()
    check(lib().dbhip_sum_a_plus_b_mul_c_i64(C.c_void_p(a.data.ptr), C.c_void_p(b.data.ptr), C.c_void_p(c.data.ptr), C.c_int64(a.n), C.c_void_p(out.ptr), None))
    return int(out.to_numpy(np.int64, 1)[0])


def column_sum(col):
    out = DeviceBuffer(8).zero()
    cc = col.c()
    check(lib().dbhip_sum(C.byref(cc), C.c_int64(col.n), C.c_void_p(out.ptr), None))
    if col.dtype in (L.T_F32, L.T_F64):
        return float(out.to_numpy(np.float64, 1)[0])
    if col.dtype in (L.T_U8, L.T_U16, L.T_U32, L.T_U64):
        return int(out.to_numpy(np.uint64, 1)[0])
    return int(out.to_numpy(np.int64, 1)[0])


class ExprProgram:
    """Post-order flattening of an Expr tree into the register program of dbhip_expr_eval (the job of the binding's
    Evaluator::run replacement). Methods return the register that holds the node's value. Decimal nodes carry their
    DecimalSize (ArithmeticOp::result_size, decimal/src/arithmetic.rs:80-139)."""

    ARITH = {L.EX_PLUS: L.OP_PLUS, L.EX_MINUS: L.OP_MINUS, L.EX_MULTIPLY: L.OP_MULTIPLY, L.EX_DIVIDE: L.OP_DIVIDE}
    INT_PROPS = {L.T_I8: (3, 0), L.T_U8: (3, 0), L.T_I16: (5, 0), L.T_U16: (5, 0), L.T_I32: (10, 0), L.T_U32: (10, 0),
                 L.T_I64: (19, 0), L.T_U64: (20, 0)}

    def __init__(self, inputs):
        self.inputs = list(inputs)
        self.ins = []
        self.types = {}
        self.size = {}   # register -> (precision, scale) of decimal values
        self.free = list(range(16))

    def _emit(self, op, a, b, typ, imm=0, release=(), size=(0, 0)):
        for r in release:
            if r not in self.free:
                self.free.append(r)
        self.free.sort()
        dst = self.free.pop(0)
        self.ins.append((op, dst, a, b, typ, imm, size))
        self.types[dst] = typ
        self.size[dst] = size
        return dst

    def load(self, i):
        c = self.inputs[i]
        return self._emit(L.EX_LOAD, i, 0, c.dtype, size=(c.precision, c.scale))

    def const(self, value, typ, precision=0, scale=0):
        if typ in (L.T_F32, L.T_F64):
            imm = int(np.float64(np.float32(value) if typ == L.T_F32 else value).view(np.uint64))
        else:
            imm = int(value) & ((1 << 64) - 1)
        return self._emit(L.EX_CONST, 0, 0, typ, imm, size=(precision, scale))

    def _props(self, r):
        t = self.types[r]
        return self.size[r] if t in (L.T_DEC64, L.T_DEC128) else self.INT_PROPS[t]

    def arith(self, op, a, b, keep=()):
        """`keep`: operand registers that are read again later (not released)"""
        ta, tb = self.types[a], self.types[b]
        rel = tuple(r for r in (a, b) if r not in keep)
        if ta in (L.T_DEC64, L.T_DEC128) or tb in (L.T_DEC64, L.T_DEC128):
            (ap, as_), (bp, bs) = self._props(a), self._props(b)
            p, sc = C.c_uint8(), C.c_uint8()
            check(lib().dbhip_decimal_result_size(self.ARITH[op], ap, as_, bp, bs, C.byref(p), C.byref(sc)))
            typ = L.T_DEC64 if p.value <= 18 else L.T_DEC128
            return self._emit(op, a, b, typ, release=rel, size=(p.value, sc.value))
        typ = lib().dbhip_arith_result_type(self.ARITH[op], ta, tb)
        return self._emit(op, a, b, typ, release=rel)

    def cmp(self, op, a, b, keep=()):
        return self._emit(op, a, b, L.T_BOOL, release=tuple(r for r in (a, b) if r not in keep))

    def logic(self, op, a, b=0):
        return self._emit(op, a, b, L.T_BOOL, release=(a,) if op == L.EX_NOT else (a, b))

    def is_true(self, a):
        """decode_predicate: NULL / FALSE -> FALSE, never NULL (utils/filter_helper.rs)"""
        return self._emit(L.EX_IS_TRUE, a, 0, L.T_BOOL, release=(a,))

    def or_filters(self, *regs):
        """or_filters / and_filters (evaluator.rs:1802-1880): every argument decoded (NULL -> FALSE), then OR-ed / AND-ed"""
        acc = self.is_true(regs[0])
        for r in regs[1:]:
            acc = self.logic(L.EX_OR, acc, self.is_true(r))
        return acc

    def and_filters(self, *regs):
        acc = self.is_true(regs[0])
        for r in regs[1:]:
            acc = self.logic(L.EX_AND, acc, self.is_true(r))
        return acc

    def cast(self, a, typ, precision=0, scale=0):
        size = (precision, scale) if typ in (L.T_DEC64, L.T_DEC128) else (0, 0)
        if typ == L.T_DEC128 and not precision:
            size = (38, self.size[a][1])
        return self._emit(L.EX_CAST, a, 0, typ, release=(a,), size=size)

    def if_(self, cond, then, other):
        return self._emit(L.EX_IF, cond, then, self.types[then], imm=other, release=(cond, then, other), size=self.size[then])

    def c_program(self):
        prog = (L.ExprIns * max(len(self.ins), 1))()
        for k, ins in enumerate(self.ins):
            op, dst, a, b, typ, imm = ins[:6]
            prog[k].op, prog[k].dst, prog[k].a, prog[k].b, prog[k].type, prog[k].imm = op, dst, a, b, typ, imm
            prog[k].precision, prog[k].scale = ins[6] if len(ins) > 6 else (0, 0)
        return prog

    def run(self, out_reg, n=None, want_values=True, want_sum=False, errors=None):
        """-> dict(values=np array | bool array, validity=bool array | None, sum=int/float | None)"""
        n = self.inputs[0].n if n is None else n
        prog = self.c_program()
        ot = self.types[out_reg]
        words = (max(n, 1) + 63) // 64
        nullable = any(c.validity is not None for c in self.inputs)
        vals = None
        if want_values:
            vals = DeviceBuffer(words * 8 if ot == L.T_BOOL else max(n, 1) * ELEM_SIZE[ot] + 64)
        vbuf = DeviceBuffer(words * 8) if nullable else None
        sbuf = None
        if want_sum:
            sbuf = DeviceBuffer(8)
            sbuf.zero()
        check(lib().dbhip_expr_eval(prog, len(self.ins), _cols(self.inputs), len(self.inputs), C.c_int64(n), out_reg,
                                    C.c_void_p(vals.ptr) if vals else None, C.c_void_p(vbuf.ptr) if vbuf else None,
                                    C.c_void_p(errors.bitmap.ptr) if errors else None, C.c_void_p(errors.count.ptr) if errors else None,
                                    C.c_void_p(sbuf.ptr) if sbuf else None, None))
        out = dict(values=None, validity=None, sum=None, type=ot)
        out["size"] = self.size.get(out_reg, (0, 0))
        if vals is not None:
            if ot == L.T_BOOL:
                out["values"] = unpack_bits(vals.to_numpy(np.uint8, words * 8), n)
            elif ot == L.T_DEC128:
                out["values"] = bytes_to_i128(vals.to_numpy(np.uint8, 16 * n))
            else:
                out["values"] = vals.to_numpy(NP_OF[ot], n)
        if vbuf is not None:
            out["validity"] = unpack_bits(vbuf.to_numpy(np.uint8, words * 8), n)
        if sbuf is not None:
            cls_float = ot in (L.T_F32, L.T_F64)
            out["sum"] = float(sbuf.to_numpy(np.float64, 1)[0]) if cls_float else int(sbuf.to_numpy(np.int64 if ot in (L.T_I8, L.T_I16, L.T_I32, L.T_I64, L.T_DATE, L.T_TIMESTAMP, L.T_DEC64) else np.uint64, 1)[0])
        return out


def _filter_bits(pred, n):
    """The Bitmap a pushed-down predicate hands to the aggregate: a NULL predicate row is dropped like FALSE (the reference's
    filter treats NULL as false, filter_executor.rs), so a nullable Boolean column contributes data AND validity."""
    if pred.validity is None:
        return pred.data
    out = DeviceBuffer(((n + 63) // 64) * 8 + 8)
    check(lib().dbhip_bitmap_binary(0, C.c_void_p(pred.data.ptr), C.c_void_p(pred.validity.ptr), C.c_int64(n), C.c_void_p(out.ptr), None))
    return out


class GroupBy:
    """Device AggregateHashTable (dbhip_groupby_*)."""

    def __init__(self, key_types, aggs, key_nullable=None, capacity=1024, handle=None):
        _ensure()
        self.key_types = list(key_types)
        self.aggs = list(aggs)  # (kind, arg_type, precision, scale, nullable)
        self.key_nullable = list(key_nullable) if key_nullable else [0] * len(key_types)
        if handle is not None:
            self.h = handle
        else:
            kt = (C.c_int32 * len(key_types))(*key_types)
            kn = (C.c_uint8 * len(key_types))(*self.key_nullable)
            ad = (AggDesc * max(len(aggs), 1))()
            for i, a in enumerate(aggs):
                ad[i].kind, ad[i].arg_type, ad[i].arg_precision, ad[i].arg_scale, ad[i].arg_nullable = a
            self.h = C.c_void_p()
            check(lib().dbhip_groupby_create(kt, kn, len(key_types), ad, len(aggs), C.c_int64(capacity), C.byref(self.h)))

    @classmethod
    def q1(cls):
        _ensure()
        h = C.c_void_p()
        check(lib().dbhip_q1_create_groupby(C.byref(h)))
        aggs = [(L.AGG_SUM, L.T_DEC64, 15, 2, 0), (L.AGG_SUM, L.T_DEC64, 15, 2, 0), (L.AGG_SUM, L.T_DEC128, 31, 4, 0),
                (L.AGG_SUM, L.T_DEC128, 38, 6, 0), (L.AGG_SUM, L.T_DEC64, 15, 2, 0), (L.AGG_COUNT, 0, 0, 0, 0)]
        return cls([L.T_STRING, L.T_STRING], aggs, handle=h)

    def add_block(self, keys, args, n, filter=None, stream=None):
        """AggregateHashTable::add_groups. `filter`: Boolean Column over the same (unfiltered) rows — the pushed-down
        TransformFilter predicate (dbhip_groupby_add_block_filtered)."""
        ka = _cols(keys)
        aa = (Col * max(len(self.aggs), 1))()
        for i, a in enumerate(args):
            if a is not None:
                aa[i] = a.c()
        if filter is None:
            check(lib().dbhip_groupby_add_block(self.h, ka, aa, C.c_int64(n), stream))
        else:
            check(lib().dbhip_groupby_add_block_filtered(self.h, ka, aa, C.c_int64(n), C.c_void_p(_filter_bits(filter, n).ptr), C.c_int64(0), stream))

    def add_block_program(self, keys, program, arg_regs, n, filter_reg=-1, filter=None, stream=None, prepare=False):
        """Fused TransformFilter -> maps -> partial aggregate (dbhip_groupby_add_block_program). `program`: ExprProgram;
        `arg_regs[i]`: register of aggregate i's argument, ("input", c) for input column c as it is, None for count(*)."""
        regs = (C.c_int32 * max(len(self.aggs), 1))()
        for i, r in enumerate(arg_regs):
            regs[i] = -(2 ** 31) if r is None else (-(1 + r[1]) if isinstance(r, tuple) else r)
        ap = L.AggProgram()
        cprog = program.c_program()
        cin = _cols(program.inputs)
        ap.prog, ap.n_ins = C.cast(cprog, C.c_void_p), len(program.ins)
        ap.inputs, ap.n_inputs = C.cast(cin, C.c_void_p), len(program.inputs)
        ap.filter_reg, ap.arg_regs = filter_reg, C.cast(regs, C.c_void_p)
        fb = C.c_void_p(_filter_bits(filter, n).ptr) if filter is not None else None
        if prepare:
            check(lib().dbhip_groupby_prepare_program(self.h, _cols(keys), C.byref(ap)))
            return
        check(lib().dbhip_groupby_add_block_program(self.h, _cols(keys), C.byref(ap), C.c_int64(n), fb, C.c_int64(0), stream))

    def set_pipelined(self, on=True, stream=None):
        """dbhip_groupby_set_pipelined: add_block_program calls queue one launch each and return; `checkpoint` reports."""
        check(lib().dbhip_groupby_set_pipelined(self.h, C.c_int32(1 if on else 0), stream))

    def checkpoint(self, stream=None, raise_on_error=True):
        """dbhip_groupby_checkpoint -> blocks merged since the previous checkpoint (raises the deferred error unless
        raise_on_error=False: then -> (rc, committed))."""
        n = C.c_int64()
        rc = lib().dbhip_groupby_checkpoint(self.h, C.byref(n), stream)
        if not raise_on_error:
            return rc, n.value
        check(rc)
        return n.value

    def prepare_program(self, keys, program, arg_regs, filter_reg=-1):
        """dbhip_groupby_prepare_program: compile the run-time specialised kernel of this query shape now (blocking, cached)."""
        self.add_block_program(keys, program, arg_regs, 1, filter_reg=filter_reg, prepare=True)

    def state_fields(self):
        """-> [(dbhip_type, aggregate index)] of the serialized-state block (dbhip_groupby_state_fields)."""
        t, a, n = (C.c_int32 * 96)(), (C.c_int32 * 96)(), C.c_int32()
        check(lib().dbhip_groupby_state_fields(self.h, t, a, 96, C.byref(n)))
        return [(t[i], a[i]) for i in range(n.value)]

    def merge_state_block(self, keys, states, n):
        """batch_merge of a serialized-state block [state fields..., group columns...] (payload_flush.rs:151-181)."""
        aa = (Col * max(len(states), 1))()
        for i, a in enumerate(states):
            aa[i] = a.c()
        check(lib().dbhip_groupby_merge_state_block(self.h, _cols(keys), aa, C.c_int64(n), None))

    def flush_state_block(self):
        """Payload::aggregate_flush as HBM-resident Columns -> (key Columns, state field Columns)."""
        g = self.num_groups()
        cap = max(g, 1)
        fields = self.state_fields()
        key_bufs = [DeviceBuffer(cap * ELEM_SIZE.get(t, 1) + 64) for t in self.key_types]
        key_val = [DeviceBuffer(((cap + 63) // 64) * 8 + 8) for _ in self.key_types]
        fbufs = [DeviceBuffer((((cap + 63) // 64) * 8 + 8) if t == L.T_BOOL else cap * ELEM_SIZE[t] + 64) for t, _ in fields]
        kp = (C.c_void_p * len(key_bufs))(*[b.ptr for b in key_bufs])
        kv = (C.c_void_p * len(key_val))(*[b.ptr for b in key_val])
        fp = (C.c_void_p * max(len(fbufs), 1))(*[b.ptr for b in fbufs])
        n = C.c_int64()
        check(lib().dbhip_groupby_flush_state_block(self.h, kp, kv, fp, None, C.c_int64(cap), C.byref(n), None))
        n = n.value
        keys = [Column(t, n, b, v if nul else None) for t, b, v, nul in zip(self.key_types, key_bufs, key_val, self.key_nullable)]
        states = []
        arena_ptrs = None
        for (t, a), b in zip(fields, fbufs):
            kind, at, p, sc, _nul = self.aggs[a]
            prec, scale = (0, 0)
            if t in (L.T_DEC64, L.T_DEC128, L.T_DEC256):
                prec, scale = ({L.T_DEC64: 18, L.T_DEC128: 38, L.T_DEC256: 76}[t], sc) if kind == L.AGG_SUM else (p, sc)
            if t == L.T_STRING:
                # the value buffer of a Nullable(String) state (min / max over String): long views point into the table's arena
                # (buffer 0; valid until the table is reset, destroyed or takes more rows)
                if arena_ptrs is None:
                    arena_ptrs = DeviceBuffer.from_numpy(np.array([self.arena()[0] or 0], dtype=np.uint64))
                states.append(Column(t, n, b, None, buffers=arena_ptrs))
            else:
                states.append(Column(t, n, b, None, prec, scale))
        return keys, states

    def arena(self):
        """-> (device pointer, bytes in use) of the long string keys' bytes (dbhip_groupby_arena)"""
        p, n = C.c_void_p(), C.c_int64()
        check(lib().dbhip_groupby_arena(self.h, C.byref(p), C.byref(n), None))
        return p.value, n.value

    def arena_numpy(self):
        p, n = self.arena()
        if not n:
            return np.zeros(0, np.uint8)
        return BorrowedBuffer(p, n).to_numpy(np.uint8, n)

    def merge_serialized_arena(self, rows, arena):
        """rows (uint64 [n, W]) of another table + that table's arena bytes (numpy uint8)"""
        rows = np.ascontiguousarray(rows, dtype=np.uint64)
        if rows.size == 0:
            return
        buf = DeviceBuffer.from_numpy(rows)
        ab = DeviceBuffer.from_numpy(np.concatenate([np.ascontiguousarray(arena, dtype=np.uint8), np.zeros(8, np.uint8)]))
        check(lib().dbhip_groupby_merge_serialized_arena(self.h, C.c_void_p(buf.ptr), C.c_int64(rows.shape[0]), C.c_void_p(ab.ptr), None))

    def partition_blocks(self, blocks_ptr, n_buckets, max_rows, stream=None):
        """dbhip_groupby_partition_blocks: rows routed to bucket hash % n_buckets, one fixed-size block per bucket."""
        check(lib().dbhip_groupby_partition_blocks(self.h, C.c_int32(n_buckets), C.c_void_p(blocks_ptr), C.c_int64(max_rows), stream))

    def replace_with_blocks(self, blocks_ptr, n_blocks, max_rows, stream=None):
        check(lib().dbhip_groupby_replace_with_blocks(self.h, C.c_void_p(blocks_ptr), C.c_int32(n_blocks), C.c_int64(max_rows), stream))

    def flush_partitioned(self, n_buckets, rows_ptr, max_rows, stream=None):
        """-> counts[n_buckets]; the rows (grouped by bucket) are written to the device buffer at rows_ptr."""
        cnt = (C.c_int64 * n_buckets)()
        check(lib().dbhip_groupby_flush_partitioned(self.h, C.c_int32(n_buckets), C.c_void_p(rows_ptr), C.c_int64(max_rows), cnt, stream))
        return [int(x) for x in cnt]

    def merge_serialized_device(self, rows_ptr, n_rows, stream=None):
        if n_rows:
            check(lib().dbhip_groupby_merge_serialized(self.h, C.c_void_p(rows_ptr), C.c_int64(n_rows), stream))

    def num_groups(self):
        n = C.c_int64()
        check(lib().dbhip_groupby_num_groups(self.h, C.byref(n), None))
        return n.value

    def row_bytes(self):
        n = C.c_int64()
        check(lib().dbhip_groupby_row_bytes(self.h, C.byref(n)))
        return n.value

    def flush_serialized(self):
        """-> (rows as uint64 ndarray [n, W])"""
        g = self.num_groups()
        rb = self.row_bytes()
        buf = DeviceBuffer(max(g, 1) * rb)
        n = C.c_int64()
        check(lib().dbhip_groupby_flush_serialized(self.h, C.c_void_p(buf.ptr), C.c_int64(g), C.byref(n), None))
        return buf.to_numpy(np.uint64, n.value * rb // 8).reshape(n.value, rb // 8)

    def merge_serialized(self, rows):
        rows = np.ascontiguousarray(rows, dtype=np.uint64)
        if rows.size == 0:
            return
        buf = DeviceBuffer.from_numpy(rows)
        check(lib().dbhip_groupby_merge_serialized(self.h, C.c_void_p(buf.ptr), C.c_int64(rows.shape[0]), None))

    def flush_block(self, block_ptr, max_rows, stream=None):
        """dbhip_groupby_flush_block: header row + serialized rows into a device block, no host synchronisation."""
        check(lib().dbhip_groupby_flush_block(self.h, C.c_void_p(block_ptr), C.c_int64(max_rows), stream))

    def merge_blocks(self, blocks_ptr, n_blocks, max_rows, skip_block=-1, stream=None):
        check(lib().dbhip_groupby_merge_blocks(self.h, C.c_void_p(blocks_ptr), C.c_int32(n_blocks), C.c_int64(max_rows),
                                               C.c_int32(skip_block), stream))

    def result(self):
        """-> list of rows [(key values..., agg values...)] (order unspecified)."""
        g = self.num_groups()
        cap = max(g, 1)
        key_bufs = [DeviceBuffer(cap * ELEM_SIZE.get(t, 1) + 64) for t in self.key_types]
        key_val = [DeviceBuffer(((cap + 31) // 32) * 4 + 8) for _ in self.key_types]
        agg_t = []
        for a in self.aggs:
            d = AggDesc()
            d.kind, d.arg_type, d.arg_precision, d.arg_scale, d.arg_nullable = a
            t, p, s = C.c_int32(), C.c_uint8(), C.c_uint8()
            check(lib().dbhip_groupby_result_type(C.byref(d), C.byref(t), C.byref(p), C.byref(s)))
            agg_t.append(t.value)
        agg_bufs = [DeviceBuffer(cap * ELEM_SIZE[t] + 64) for t in agg_t]
        hashes = DeviceBuffer(cap * 8)
        kp = (C.c_void_p * len(key_bufs))(*[b.ptr for b in key_bufs])
        kv = (C.c_void_p * len(key_val))(*[b.ptr for b in key_val])
        ap = (C.c_void_p * max(len(agg_bufs), 1))(*[b.ptr for b in agg_bufs])
        n = C.c_int64()
        agg_val = [DeviceBuffer(((cap + 63) // 64) * 8 + 8) for _ in self.aggs]
        av = (C.c_void_p * max(len(agg_val), 1))(*[b.ptr for b in agg_val])
        check(lib().dbhip_groupby_flush_result_nullable(self.h, kp, kv, ap, av, C.c_void_p(hashes.ptr), C.c_int64(cap), C.byref(n), None))
        n = n.value
        cols = []
        for t, b, v in zip(self.key_types, key_bufs, key_val):
            valid = unpack_bits(v.to_numpy(np.uint8, ((cap + 31) // 32) * 4), n)
            if t == L.T_STRING:
                vals = view_strings(b.to_numpy(np.uint8, 16 * n), self.arena_numpy())
            elif t == L.T_DEC128:
                vals = bytes_to_i128(b.to_numpy(np.uint8, 16 * n))
            elif t == L.T_DEC256:
                vals = limbs_to_ints(b.to_numpy(np.uint8, 32 * n), 256)
            elif t == L.T_BOOL:
                vals = [bool(x) for x in b.to_numpy(np.uint8, n)]
            else:
                vals = b.to_numpy(NP_OF[t], n).tolist()
            cols.append([x if ok else None for x, ok in zip(vals, valid)])
        for t, b, a, v in zip(agg_t, agg_bufs, self.aggs, agg_val):
            if t == L.T_DEC128:
                vals = bytes_to_i128(b.to_numpy(np.uint8, 16 * n))
            elif t == L.T_DEC256:
                vals = limbs_to_ints(b.to_numpy(np.uint8, 32 * n), 256)
            elif t == L.T_STRING:   # min / max over String: views whose long form points into the table's arena
                vals = view_strings(b.to_numpy(np.uint8, 16 * n), self.arena_numpy())
            else:
                vals = b.to_numpy(NP_OF[t], n).tolist()
            if a[4] and a[0] != L.AGG_COUNT:  # nullable argument: NULL unless the group saw a non-NULL row
                valid = unpack_bits(v.to_numpy(np.uint8, ((cap + 63) // 64) * 8), n)
                vals = [x if ok else None for x, ok in zip(vals, valid)]
            cols.append(vals)
        self.last_hashes = hashes.to_numpy(np.uint64, n)
        return [tuple(c[i] for c in cols) for i in range(n)]

    def result_columns(self):
        """merge_result as HBM-resident Columns [keys..., aggregate results...] (group order unspecified);
        nothing is copied to the host (the next operator — sort, projection — consumes them in place)."""
        g = self.num_groups()
        cap = max(g, 1)
        key_bufs = [DeviceBuffer(cap * ELEM_SIZE.get(t, 1) + 64) for t in self.key_types]
        key_val = [DeviceBuffer(((cap + 63) // 64) * 8 + 8) for _ in self.key_types]
        agg_meta = []
        for a in self.aggs:
            d = AggDesc()
            d.kind, d.arg_type, d.arg_precision, d.arg_scale, d.arg_nullable = a
            t, p, s = C.c_int32(), C.c_uint8(), C.c_uint8()
            check(lib().dbhip_groupby_result_type(C.byref(d), C.byref(t), C.byref(p), C.byref(s)))
            agg_meta.append((t.value, p.value, s.value))
        agg_bufs = [DeviceBuffer(cap * ELEM_SIZE[t] + 64) for t, _, _ in agg_meta]
        kp = (C.c_void_p * len(key_bufs))(*[b.ptr for b in key_bufs])
        kv = (C.c_void_p * len(key_val))(*[b.ptr for b in key_val])
        ap = (C.c_void_p * max(len(agg_bufs), 1))(*[b.ptr for b in agg_bufs])
        n = C.c_int64()
        check(lib().dbhip_groupby_flush_result(self.h, kp, kv, ap, None, C.c_int64(cap), C.byref(n), None))
        n = n.value
        cols = [Column(t, n, b, v if nul else None) for t, b, v, nul in zip(self.key_types, key_bufs, key_val, self.key_nullable)]
        cols += [Column(t, n, b, None, p, s) for (t, p, s), b in zip(agg_meta, agg_bufs)]
        return cols

    def reset(self, stream=None):
        check(lib().dbhip_groupby_reset(self.h, stream))

    def debug_set_hash_mask(self, mask):
        f = lib().dbhip_groupby_debug_set_hash_mask
        f.restype = C.c_int32
        check(f(self.h, C.c_uint64(mask)))

    def debug_set_partition_bits(self, bits):
        """test hook: force (bits > 0) / forbid (bits < 0) the radix-partitioned pre-aggregation path"""
        f = lib().dbhip_groupby_debug_set_partition_bits
        f.restype = C.c_int32
        check(f(self.h, C.c_int32(bits)))

    def destroy(self):
        if self.h:
            lib().dbhip_groupby_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


def q1_fused(g, qty, price, disc, tax, rf, ls, shipdate, cutoff, n=None, stream=None):
    n = qty.n if n is None else n
    check(lib().dbhip_q1_fused(g.h, C.c_void_p(qty.data.ptr), C.c_void_p(price.data.ptr), C.c_void_p(disc.data.ptr),
                               C.c_void_p(tax.data.ptr), C.c_void_p(rf.data.ptr), C.c_void_p(ls.data.ptr),
                               C.c_void_p(shipdate.data.ptr), C.c_int32(cutoff), C.c_int64(n), stream))


def keys_method(cols):
    """choose_hash_method_with_types (kernels/group_by.rs:40-80): packed key width in bytes, 0 = Serializer."""
    out = C.c_int32()
    check(lib().dbhip_keys_method(_cols(cols), len(cols), C.byref(out)))
    return out.value


class PackedKeys:
    """Result of pack_keys: `key_bytes`-wide integers (HashMethodFixedKeys::build_keys_vec) + the all-valid bitmap."""

    def __init__(self, data, validity, n, key_bytes):
        self.data, self.validity, self.n, self.key_bytes = data, validity, n, key_bytes

    def to_numpy(self):
        return self.data.to_numpy(np.uint8, self.n * self.key_bytes).reshape(self.n, self.key_bytes)


def pack_keys(cols, key_bytes=None, want_validity=True):
    n = cols[0].n
    kb = key_bytes or keys_method(cols)
    if kb == 0:
        raise L.DbhipError(L.ERR_UNSUPPORTED, "HashMethodSerializer keys stay on the CPU")
    out = DeviceBuffer(max(n, 1) * kb + 64)
    val = DeviceBuffer((max(n, 1) + 7) // 8 + 64) if want_validity else None
    check(lib().dbhip_pack_keys(_cols(cols), len(cols), C.c_int64(n), C.c_int32(kb), C.c_void_p(out.ptr),
                                C.c_void_p(val.ptr) if val else None, None))
    return PackedKeys(out, val, n, kb)


def serialize_keys(cols, n=None):
    """HashMethodSerializer::build_keys_state (dbhip_serialize_keys): -> (offsets DeviceBuffer [n + 1] u64, data DeviceBuffer,
    all_valid DeviceBuffer bitmap, total bytes)"""
    n = n if n is not None else max(c.n for c in cols if not c.is_scalar)
    off = DeviceBuffer((n + 1) * 8 + 64)
    allv = DeviceBuffer(((n + 63) // 64) * 8 + 8)
    total = C.c_uint64()
    ca = _cols(cols)
    check(lib().dbhip_serialize_keys_offsets(ca, len(cols), C.c_int64(n), C.c_void_p(off.ptr), C.c_void_p(allv.ptr), C.byref(total), None))
    data = DeviceBuffer(int(total.value) + 64)
    check(lib().dbhip_serialize_keys(ca, len(cols), C.c_int64(n), C.c_void_p(off.ptr), C.c_void_p(data.ptr), None))
    return off, data, allv, int(total.value)


class BinaryHashJoin:
    """Hash join on serialized keys (HashMethodSerializer: string keys of any length, keys wider than 32 bytes):
    dbhip_join_*_binary. add_block / probe_block take the key COLUMNS and serialize them on the device."""

    def __init__(self, expected_build_rows=1024):
        _ensure()
        self.h = C.c_void_p()
        check(lib().dbhip_join_create_binary(C.c_int64(expected_build_rows), C.byref(self.h)))

    def add_block(self, key_cols, n=None, use_validity=True):
        n = n if n is not None else max(c.n for c in key_cols if not c.is_scalar)
        off, data, allv, _ = serialize_keys(key_cols, n)
        nullable = use_validity and any(c.validity is not None for c in key_cols)
        check(lib().dbhip_join_add_build_binary(self.h, C.c_void_p(off.ptr), C.c_void_p(data.ptr), C.c_void_p(allv.ptr) if nullable else None, C.c_int64(n), None))

    def final_build(self):
        check(lib().dbhip_join_finalize_binary(self.h, None))

    def probe_block(self, key_cols, n=None, use_validity=True):
        """-> (probe_idx u32[], build_row u32[], matched bool[n])"""
        n = n if n is not None else max(c.n for c in key_cols if not c.is_scalar)
        off, data, allv, _ = serialize_keys(key_cols, n)
        nullable = use_validity and any(c.validity is not None for c in key_cols)
        vptr = C.c_void_p(allv.ptr) if nullable else None
        cap = C.c_uint64()
        check(lib().dbhip_join_probe_count_binary(self.h, C.c_void_p(off.ptr), C.c_void_p(data.ptr), vptr, C.c_int64(n), C.byref(cap), None))
        m = int(cap.value)
        pi, bi = DeviceBuffer(max(m, 1) * 4 + 64), DeviceBuffer(max(m, 1) * 4 + 64)
        mark = DeviceBuffer(((n + 31) // 32) * 4 + 8)
        got = C.c_uint64()
        check(lib().dbhip_join_probe_binary(self.h, C.c_void_p(off.ptr), C.c_void_p(data.ptr), vptr, C.c_int64(n), C.c_void_p(pi.ptr), C.c_void_p(bi.ptr),
                                            C.c_int64(m), C.byref(got), C.c_void_p(mark.ptr), None))
        k = int(got.value)
        return pi.to_numpy(np.uint32, k), bi.to_numpy(np.uint32, k), unpack_bits(mark.to_numpy(np.uint8, (n + 7) // 8), n)

    def destroy(self):
        if self.h:
            lib().dbhip_join_destroy_binary(self.h)
            self.h = None

    def __del__(self):
        try:
            self.destroy()
        except Exception:  # noqa: BLE001
            pass


def _with_true_validity(col, n):
    """wrap_true_validity: the probe side of a right join becomes Nullable (all valid for matched rows)"""
    if col.validity is not None:
        return col
    ones = DeviceBuffer(((max(n, 1) + 63) // 64) * 8 + 8)
    check(lib().dbhip_memset(C.c_void_p(ones.ptr), 0xFF, C.c_size_t(((max(n, 1) + 63) // 64) * 8), None))
    col.validity = ones
    return col


class HashJoin:
    """Device hash join on packed fixed keys (dbhip_join_*; trait Join, new_hash_join/join.rs:26-53).
    key_bytes 8 = KeysU8..U64 (zero-extended), 16 = KeysU128, 32 = KeysU256."""

    def __init__(self, expected_build_rows=1024, key_bytes=8):
        _ensure()
        self.h = C.c_void_p()
        self.key_bytes = key_bytes
        check(lib().dbhip_join_create_keys(C.c_int64(expected_build_rows), C.c_int32(key_bytes), C.byref(self.h)))

    def probe_mark(self, keys_col):
        """-> bool[n]: probe row has a build match (semi / anti / left-outer joins)."""
        v = C.c_void_p(keys_col.validity.ptr) if keys_col.validity is not None else None
        n = keys_col.n
        bm = DeviceBuffer((max(n, 1) + 7) // 8 + 64)
        total = C.c_uint64()
        check(lib().dbhip_join_probe_mark(self.h, C.c_void_p(keys_col.data.ptr), v, C.c_int64(n), C.c_void_p(bm.ptr), C.byref(total), None))
        bits = unpack_bits(bm.to_numpy(np.uint8, (n + 7) // 8), n)
        assert int(bits.sum()) == total.value
        return bits

    def add_block(self, keys_col):
        """Join::add_block: one build chunk (u64 key column, optional validity)."""
        v = C.c_void_p(keys_col.validity.ptr) if keys_col.validity is not None else None
        check(lib().dbhip_join_add_build(self.h, C.c_void_p(keys_col.data.ptr), v, C.c_int64(keys_col.n), None))

    def final_build(self):
        check(lib().dbhip_join_finalize(self.h, None))

    def probe_block(self, keys_col):
        """-> (probe_idx u32[], build_row u32[]) sorted by (probe_idx, build_row)."""
        v = C.c_void_p(keys_col.validity.ptr) if keys_col.validity is not None else None
        total = C.c_uint64()
        check(lib().dbhip_join_probe_count(self.h, C.c_void_p(keys_col.data.ptr), v, C.c_int64(keys_col.n), C.byref(total), None))
        m = total.value
        op, ob = DeviceBuffer(max(m, 1) * 4), DeviceBuffer(max(m, 1) * 4)
        got = C.c_uint64()
        check(lib().dbhip_join_probe(self.h, C.c_void_p(keys_col.data.ptr), v, C.c_int64(keys_col.n), C.c_void_p(op.ptr),
                                     C.c_void_p(ob.ptr), C.c_int64(m), C.byref(got), None))
        assert got.value == m
        return op.to_numpy(np.uint32, m), ob.to_numpy(np.uint32, m)

    def probe_block_device(self, keys_col):
        """Join::probe_block keeping the pair lists in HBM -> (probe_idx DeviceBuffer, build_row DeviceBuffer, n_pairs)."""
        v = C.c_void_p(keys_col.validity.ptr) if keys_col.validity is not None else None
        total = C.c_uint64()
        check(lib().dbhip_join_probe_count(self.h, C.c_void_p(keys_col.data.ptr), v, C.c_int64(keys_col.n), C.byref(total), None))
        m = total.value
        op, ob = DeviceBuffer(max(m, 1) * 4 + 64), DeviceBuffer(max(m, 1) * 4 + 64)
        got = C.c_uint64()
        check(lib().dbhip_join_probe(self.h, C.c_void_p(keys_col.data.ptr), v, C.c_int64(keys_col.n), C.c_void_p(op.ptr),
                                     C.c_void_p(ob.ptr), C.c_int64(m), C.byref(got), None))
        assert got.value == m
        return op, ob, m

    def _join_conjunct(self, kind, probe_keys, probe_cols, build_cols, conjunct):
        """The same kinds with ANOTHER CONJUNCT on top of the key equality (the `CONJUNCT = true` streams: inner_join.rs:278-310
        InnerHashJoinFilterStream, left_join.rs:262-292, left_join_semi.rs:320-350, left_join_anti.rs:270-300, right_join.rs:256-290):
        the joined rows of every key match go through the conjunct's filter (a NULL result drops the pair like FALSE); a probe row
        all of whose pairs were dropped is UNMATCHED (left: it comes back with a NULL build side; anti: it is kept; semi: it is
        not), and only the surviving pairs mark their build rows (right / full). conjunct(probe Columns, build Columns, m) ->
        Boolean Column over the m joined rows."""
        n = probe_keys.n
        op, ob, m = self.probe_block_device(probe_keys)
        jp, jb = [take(c, op, m) for c in probe_cols], [take(c, ob, m) for c in build_cols]
        if m:
            pred = conjunct(jp, jb, m)
            sel, k = filter_select(Column(L.T_BOOL, m, _filter_bits(pred, m)))
        else:
            sel, k = DeviceBuffer(64), 0
        op2 = take(Column(L.T_U32, m, op), sel, k).data
        ob2 = take(Column(L.T_U32, m, ob), sel, k).data
        if kind == "inner":
            return [take(c, sel, k) for c in jp], [take(c, sel, k) for c in jb], k
        if kind in ("right", "right_semi", "right_anti", "full"):
            check(lib().dbhip_join_mark_build(self.h, C.c_void_p(ob2.ptr), C.c_int64(k), None))
            if kind in ("right_semi", "right_anti"):
                return [], [], 0
            if kind == "right":
                return [_with_true_validity(take(c, sel, k), k) for c in jp], [take(c, sel, k) for c in jb], k
        # the probe rows that kept a pair
        bm = DeviceBuffer(((max(n, 1) + 63) // 64) * 8 + 64)
        bm.zero()
        check(lib().dbhip_bitmap_set_indices(C.c_void_p(op2.ptr), C.c_int64(k), C.c_void_p(bm.ptr), C.c_int64(n), None))
        matched = Column(L.T_BOOL, n, bm)
        if kind == "left_semi":
            s2, k2 = filter_select(matched)
            return [take(c, s2, k2) for c in probe_cols], [], k2
        false_ = Column.boolean(np.zeros(1, dtype=bool))
        false_.is_scalar = True
        usel, uk = filter_select(cmp(L.CMP_EQ, matched, false_, n))
        if kind == "left_anti":
            return [take(c, usel, uk) for c in probe_cols], [], uk
        if kind not in ("left", "full"):
            raise ValueError(kind)
        return self._left_outer_rows(probe_cols, build_cols, op2, ob2, k, usel, uk)

    def _left_outer_rows(self, probe_cols, build_cols, op, ob, m, usel, uk):
        """the m matched pairs followed by the uk unmatched probe rows with a NULL build block (left_join.rs:196-232)"""
        rows = m + uk
        pidx, bidx = DeviceBuffer(max(rows, 1) * 4 + 64), DeviceBuffer(max(rows, 1) * 4 + 64)
        check(lib().dbhip_memcpy_d2d(C.c_void_p(pidx.ptr), C.c_void_p(op.ptr), C.c_size_t(m * 4), None))
        check(lib().dbhip_memcpy_d2d(C.c_void_p(pidx.ptr + m * 4), C.c_void_p(usel.ptr), C.c_size_t(uk * 4), None))
        check(lib().dbhip_memcpy_d2d(C.c_void_p(bidx.ptr), C.c_void_p(ob.ptr), C.c_size_t(m * 4), None))
        if uk:
            check(lib().dbhip_memset(C.c_void_p(bidx.ptr + m * 4), 0xFF, C.c_size_t(uk * 4), None))
        out_b = []
        for c in build_cols:
            es = ELEM_SIZE[c.dtype]
            data = DeviceBuffer(max(rows, 1) * es + 64)
            valid = DeviceBuffer(((max(rows, 1) + 63) // 64) * 8 + 8)
            sv = C.c_void_p(c.validity.ptr) if c.validity is not None else None
            check(lib().dbhip_take_outer(C.c_void_p(c.data.ptr), sv, C.c_int64(0), es, C.c_void_p(bidx.ptr), C.c_int64(rows), C.c_void_p(data.ptr),
                                         C.c_void_p(valid.ptr), None))
            out_b.append(Column(c.dtype, rows, data, valid, c.precision, c.scale, buffers=c.buffers, keep=(c,)))
        return [take(c, pidx, rows) for c in probe_cols], out_b, rows

    def join(self, kind, probe_keys, probe_cols, build_cols, conjunct=None):
        """Output assembly of `kind` in ("inner", "left", "left_semi", "left_anti", "right", "right_semi", "right_anti", "full") for
        ONE probe block against the finished build side (new_hash_join/memory/{inner_join,left_join,left_join_semi,left_join_anti,
        right_join,right_join_semi,right_join_anti,full_join}.rs; no other conjunct); the right / full kinds are completed by
        final_probe() after the last block:
          inner      matched pairs: probe columns taken by probe_idx, build columns by build_row
          left       the same, build columns Nullable with a true validity (wrap_true_validity), FOLLOWED by the unmatched
                     probe rows with a null build block (left_join.rs:196-232)
          left_semi  probe rows that have a match, each once;  left_anti: probe rows without one
        -> (probe Columns, build Columns, n_rows); everything stays in HBM."""
        if conjunct is not None:
            return self._join_conjunct(kind, probe_keys, probe_cols, build_cols, conjunct)
        n = probe_keys.n
        v = C.c_void_p(probe_keys.validity.ptr) if probe_keys.validity is not None else None
        if kind in ("left_semi", "left_anti", "left"):
            words = (max(n, 1) + 63) // 64
            bm = DeviceBuffer(words * 8 + 64)
            bm.zero()
            total = C.c_uint64()
            check(lib().dbhip_join_probe_mark(self.h, C.c_void_p(probe_keys.data.ptr), v, C.c_int64(n), C.c_void_p(bm.ptr), C.byref(total), None))
            matched = Column(L.T_BOOL, n, bm)
            if kind == "left_semi":
                sel, k = filter_select(matched)
                return [take(c, sel, k) for c in probe_cols], [], k
            # NOT matched: a ^ ones has no entry point; a Boolean equality with false is the reference's own `not`
            false_ = Column.boolean(np.zeros(1, dtype=bool))
            false_.is_scalar = True
            unmatched = cmp(L.CMP_EQ, matched, false_, n)
            usel, uk = filter_select(unmatched)
            if kind == "left_anti":
                return [take(c, usel, uk) for c in probe_cols], [], uk
        op, ob, m = self.probe_block_device(probe_keys)
        if kind in ("right", "right_semi", "right_anti", "full"):
            # the build side remembers which of its rows found a partner, across probe blocks (right_join.rs: the scan map)
            check(lib().dbhip_join_mark_build(self.h, C.c_void_p(ob.ptr), C.c_int64(m), None))
            if kind in ("right_semi", "right_anti"):
                return [], [], 0     # everything comes out of final_probe
            if kind == "right":
                return [_with_true_validity(take(c, op, m), m) for c in probe_cols], [take(c, ob, m) for c in build_cols], m
            kind = "left"            # full = left outer per probe block + the unmatched build rows at final_probe
            words = (max(n, 1) + 63) // 64
            bm = DeviceBuffer(words * 8 + 64)
            bm.zero()
            total = C.c_uint64()
            check(lib().dbhip_join_probe_mark(self.h, C.c_void_p(probe_keys.data.ptr), v, C.c_int64(n), C.c_void_p(bm.ptr), C.byref(total), None))
            false_ = Column.boolean(np.zeros(1, dtype=bool))
            false_.is_scalar = True
            usel, uk = filter_select(cmp(L.CMP_EQ, Column(L.T_BOOL, n, bm), false_, n))
        if kind == "inner":
            return [take(c, op, m) for c in probe_cols], [take(c, ob, m) for c in build_cols], m
        if kind != "left":
            raise ValueError(kind)
        return self._left_outer_rows(probe_cols, build_cols, op, ob, m, usel, uk)

    def final_probe(self, kind, build_cols, probe_cols_like=()):
        """After the last probe block of a right / right_semi / right_anti / full join (Join::final_probe,
        new_hash_join/memory/right_join*.rs, full_join.rs): the build rows that never found a partner, with a NULL probe side
        (right, full), or alone (right_anti), or the build rows that did (right_semi). -> (probe Columns, build Columns, n)"""
        nb = C.c_int64()
        check(lib().dbhip_join_build_matched(self.h, None, C.byref(nb), None))
        nb = nb.value
        bm = DeviceBuffer(((max(nb, 1) + 63) // 64) * 8 + 8)
        check(lib().dbhip_join_build_matched(self.h, C.c_void_p(bm.ptr), C.byref(C.c_int64()), None))
        matched = Column(L.T_BOOL, nb, bm)
        if kind == "right_semi":
            sel, k = filter_select(matched)
        else:
            false_ = Column.boolean(np.zeros(1, dtype=bool))
            false_.is_scalar = True
            sel, k = filter_select(cmp(L.CMP_EQ, matched, false_, nb))
        out_b = [take(c, sel, k) for c in build_cols]
        if kind in ("right_semi", "right_anti"):
            return [], out_b, k
        none = DeviceBuffer(max(k, 1) * 4 + 64)
        check(lib().dbhip_memset(C.c_void_p(none.ptr), 0xFF, C.c_size_t(max(k, 1) * 4), None))
        out_p = []
        for c in probe_cols_like:     # all-NULL columns of the probe side's types
            es = ELEM_SIZE[c.dtype]
            data = DeviceBuffer(max(k, 1) * es + 64)
            valid = DeviceBuffer(((max(k, 1) + 63) // 64) * 8 + 8)
            check(lib().dbhip_take_outer(C.c_void_p(c.data.ptr), None, C.c_int64(0), es, C.c_void_p(none.ptr), C.c_int64(k), C.c_void_p(data.ptr),
                                         C.c_void_p(valid.ptr), None))
            out_p.append(Column(c.dtype, k, data, valid, c.precision, c.scale))
        return out_p, out_b, k

    def destroy(self):
        if self.h:
            lib().dbhip_join_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


def vec_distance_rows(metric, lhs, rhs, n, dim, elem=L.T_F32, lhs_scalar=False, rhs_scalar=False):
    """the vector scalar functions row by row (dbhip_vec_distance_rows; scalars/vector.rs:59-260,490-560): lhs / rhs = numpy arrays
    [n, dim] (or [dim] with *_scalar) of f32 / f64 / i8 -> numpy f32[n] (f64[n] for f64 rows)"""
    _ensure()
    a = DeviceBuffer.from_numpy(np.ascontiguousarray(lhs))
    b = DeviceBuffer.from_numpy(np.ascontiguousarray(rhs)) if rhs is not None else None
    odt = np.float64 if elem == L.T_F64 else np.float32
    out = DeviceBuffer(max(n, 1) * np.dtype(odt).itemsize)
    check(lib().dbhip_vec_distance_rows(metric, elem, C.c_void_p(a.ptr), 1 if lhs_scalar else 0, C.c_void_p(b.ptr if b is not None else None),
                                        1 if rhs_scalar else 0, C.c_int64(n), dim, C.c_void_p(out.ptr), None))
    return out.to_numpy(odt, n)


def siphash64(col):
    """the `siphash64` scalar function (scalars/hash.rs:323-328: SipHash-1-3, zero keys, over the value's bytes) -> numpy u64[n]
    (0 under NULL rows; the column's validity passes through)."""
    _ensure()
    out = DeviceBuffer(max(col.n, 1) * 8)
    cc = col.c()
    check(lib().dbhip_siphash64(C.byref(cc), C.c_int64(col.n), C.c_void_p(out.ptr), None))
    return out.to_numpy(np.uint64, col.n)


def scatter_indices(cols, scatter_size, default_index=0):
    """HashFlightScatter::scatter_indices (flight_scatter_hash.rs:57-330): the destination of every row of a hash-shuffle exchange
    -> (DeviceBuffer u32[n], numpy u64 rows per destination)."""
    _ensure()
    n = cols[0].n
    idx = DeviceBuffer(max(n, 1) * 4)
    counts = DeviceBuffer(int(scatter_size) * 8)
    check(lib().dbhip_scatter_indices(_cols(cols), len(cols), C.c_int64(n), C.c_uint32(scatter_size), C.c_uint64(default_index),
                                      C.c_void_p(idx.ptr), C.c_void_p(counts.ptr), None))
    return idx, counts.to_numpy(np.uint64, int(scatter_size))


def scatter_block(cols, index, scatter_size):
    """DataBlock::scatter (kernels/scatter.rs:20-66) for the value buffers of `cols` (no Bitmap columns): rows grouped by
    index[i], in order inside a destination -> list of Columns (validities are not carried: scatter them as u8 columns)."""
    n = cols[0].n
    bufs = [DeviceBuffer(max(n, 1) * ELEM_SIZE[c.dtype] + 64) for c in cols]
    srcs = (C.c_void_p * len(cols))(*[c.data.ptr for c in cols])
    dsts = (C.c_void_p * len(cols))(*[b.ptr for b in bufs])
    es = (C.c_int32 * len(cols))(*[ELEM_SIZE[c.dtype] for c in cols])
    check(lib().dbhip_scatter_block(srcs, es, len(cols), C.c_void_p(index.ptr), C.c_int64(n), C.c_uint32(scatter_size), dsts, None))
    return [Column(c.dtype, n, b, None, c.precision, c.scale, buffers=c.buffers, keep=(c,)) for c, b in zip(cols, bufs)]


def scatter_columns(cols, index, scatter_size):
    """DataBlock::scatter (kernels/scatter.rs:20-66) over whole columns — values, validities, Boolean / String / Decimal256 columns
    (dbhip_scatter_columns) -> (blocks, row_starts): blocks[d] = the Columns of destination d (views into one output buffer per
    column; every destination's Bitmaps are stand-alone, offset-0 Bitmaps)."""
    _ensure()
    n, S = cols[0].n, int(scatter_size)
    bm_bytes = 8 * (n // 64 + S + 1)
    dbufs = [DeviceBuffer(bm_bytes) if c.dtype == L.T_BOOL else DeviceBuffer(max(n, 1) * ELEM_SIZE[c.dtype] + 64) for c in cols]
    vbufs = [DeviceBuffer(bm_bytes) if c.validity is not None else None for c in cols]
    dsts = (C.c_void_p * len(cols))(*[b.ptr for b in dbufs])
    vdsts = (C.c_void_p * len(cols))(*[b.ptr if b is not None else None for b in vbufs])
    starts = (C.c_int64 * (S + 1))()
    check(lib().dbhip_scatter_columns(_cols(cols), len(cols), C.c_void_p(index.ptr if index is not None else None), C.c_int64(n), C.c_uint32(S),
                                      dsts, vdsts, starts, None))
    starts = list(starts)
    blocks = []
    for d in range(S):
        lo, hi = starts[d], starts[d + 1]
        bm_at = 8 * (lo // 64 + d)
        blk = []
        for c, db, vb in zip(cols, dbufs, vbufs):
            if c.dtype == L.T_BOOL:
                data = BorrowedBuffer(db.ptr + bm_at, bm_bytes - bm_at, keep=db)
            else:
                data = BorrowedBuffer(db.ptr + lo * ELEM_SIZE[c.dtype], (hi - lo) * ELEM_SIZE[c.dtype], keep=db)
            v = BorrowedBuffer(vb.ptr + bm_at, bm_bytes - bm_at, keep=vb) if vb is not None else None
            o = Column(c.dtype, hi - lo, data, v, c.precision, c.scale, buffers=c.buffers, keep=(c, db, vb))
            o.n_buffers = c.n_buffers
            blk.append(o)
        blocks.append(blk)
    return blocks, starts


def concat_columns(cols):
    """DataBlock::concat for one column of several blocks (kernels/concat.rs:62-340, dbhip_concat_columns) -> Column. String blocks keep
    their data buffers (the views are rebased onto the concatenated buffer table)."""
    _ensure()
    t = cols[0].dtype
    total = sum(c.n for c in cols)
    nbuf = sum(c.n_buffers for c in cols) if t == L.T_STRING else 0
    out = DeviceBuffer(((total + 63) // 64) * 8 + 8) if t == L.T_BOOL else DeviceBuffer(max(total, 1) * ELEM_SIZE[t] + 64)
    vb = DeviceBuffer(((total + 63) // 64) * 8 + 8) if any(c.validity is not None for c in cols) else None
    bufs = DeviceBuffer(max(nbuf, 1) * 8) if nbuf else None
    rows = (C.c_int64 * len(cols))(*[c.n for c in cols])
    boffs = (C.c_int64 * len(cols))(*[c.boff for c in cols])
    got = C.c_int32(0)
    check(lib().dbhip_concat_columns(_cols(cols), rows, boffs, len(cols), C.c_void_p(out.ptr), C.c_void_p(vb.ptr if vb is not None else None),
                                     C.c_void_p(bufs.ptr if bufs is not None else None), C.byref(got), None))
    assert got.value == nbuf
    check(lib().dbhip_stream_sync(None))
    o = Column(t, total, out, vb, cols[0].precision, cols[0].scale, buffers=bufs, keep=tuple(cols))
    o.n_buffers = nbuf
    return o


def sort_perm(cols, desc=None, nulls_first=None, limit=0):
    """DataBlock::sort permutation (kernels/sort.rs:91-113) -> u32 row ids."""
    n = cols[0].n
    desc = desc or [0] * len(cols)
    nulls_first = nulls_first or [0] * len(cols)
    arr = _cols(cols)
    d = (C.c_uint8 * len(cols))(*[int(bool(x)) for x in desc])
    nf = (C.c_uint8 * len(cols))(*[int(bool(x)) for x in nulls_first])
    m = limit if 0 < limit < n else n
    out = DeviceBuffer(max(m, 1) * 4)
    check(lib().dbhip_sort_perm(arr, d, nf, len(cols), C.c_int64(n), C.c_int64(limit), C.c_void_p(out.ptr), None))
    return out.to_numpy(np.uint32, m)


def sort_perm_device(cols, desc=None, nulls_first=None, limit=0):
    """sort_perm with the permutation left in HBM -> (DeviceBuffer of u32 row ids, their number)."""
    n = cols[0].n
    d = (C.c_uint8 * len(cols))(*[int(bool(x)) for x in (desc or [0] * len(cols))])
    nf = (C.c_uint8 * len(cols))(*[int(bool(x)) for x in (nulls_first or [0] * len(cols))])
    m = limit if 0 < limit < n else n
    out = DeviceBuffer(max(m, 1) * 4)
    check(lib().dbhip_sort_perm(_cols(cols), d, nf, len(cols), C.c_int64(n), C.c_int64(limit), C.c_void_p(out.ptr), None))
    return out, m


def sort_bound_partition(cols, bounds, desc=None, nulls_first=None):
    """The distributed sort's range partition (sort_spill.rs:1008-1040 partition_point over Bounds; sort_exchange_injector.rs
    SortBoundScatter): `bounds` = the same key columns holding the ordered bounds. Returns (DeviceBuffer of the u32 partition
    of every row = number of bounds sorting strictly before it, numpy u64 rows per partition [nbounds + 1])."""
    n = cols[0].n
    nb = bounds[0].n if bounds else 0
    d = (C.c_uint8 * len(cols))(*[int(bool(x)) for x in (desc or [0] * len(cols))])
    nf = (C.c_uint8 * len(cols))(*[int(bool(x)) for x in (nulls_first or [0] * len(cols))])
    part = DeviceBuffer(max(n, 1) * 4)
    counts = DeviceBuffer((nb + 1) * 8)
    check(lib().dbhip_sort_bound_partition(_cols(cols), _cols(bounds) if nb else None, d, nf, len(cols), C.c_int64(n), C.c_int64(nb),
                                           C.c_void_p(part.ptr), C.c_void_p(counts.ptr), None))
    return part, counts.to_numpy(np.uint64, nb + 1)


def merge_sorted_perm(cols, run_offsets, desc=None, nulls_first=None, limit=0):
    """k-way merge of sorted runs laid back to back (Merger / loser tree, sorts/core/merger.rs) -> u32 row ids."""
    n = int(run_offsets[-1]) if len(run_offsets) else 0
    desc = desc or [0] * len(cols)
    nulls_first = nulls_first or [0] * len(cols)
    d = (C.c_uint8 * len(cols))(*[int(bool(x)) for x in desc])
    nf = (C.c_uint8 * len(cols))(*[int(bool(x)) for x in nulls_first])
    ro = (C.c_int64 * len(run_offsets))(*[int(x) for x in run_offsets])
    m = limit if 0 < limit < n else n
    out = DeviceBuffer(max(m, 1) * 4)
    check(lib().dbhip_merge_sorted_perm(_cols(cols), d, nf, len(cols), ro, len(run_offsets) - 1, C.c_int64(limit), C.c_void_p(out.ptr), None))
    return out.to_numpy(np.uint32, m)


class VectorColumn:
    """Flat row-major f32 vectors in HBM (VectorColumn::Float32, types/vector.rs:377-380)."""

    def __init__(self, arr):
        arr = np.ascontiguousarray(arr, dtype=np.float32)
        self.n, self.dim = arr.shape
        self.data = DeviceBuffer.from_numpy(arr.reshape(-1))


def vec_distance(metric, base, queries):
    out = DeviceBuffer(max(base.n * queries.n, 1) * 4)
    check(lib().dbhip_vec_distance(metric, C.c_void_p(base.data.ptr), C.c_int64(base.n), base.dim, C.c_void_p(queries.data.ptr),
                                   queries.n, C.c_void_p(out.ptr), None))
    return out.to_numpy(np.float32, base.n * queries.n).reshape(queries.n, base.n)


def vec_topk(metric, base, queries, k):
    oi, od = DeviceBuffer(max(queries.n * k, 1) * 4), DeviceBuffer(max(queries.n * k, 1) * 4)
    check(lib().dbhip_vec_topk(metric, C.c_void_p(base.data.ptr), C.c_int64(base.n), base.dim, C.c_void_p(queries.data.ptr),
                               queries.n, k, C.c_void_p(oi.ptr), C.c_void_p(od.ptr), None))
    return oi.to_numpy(np.uint32, queries.n * k).reshape(queries.n, k), od.to_numpy(np.float32, queries.n * k).reshape(queries.n, k)


def vec_topk_merge(dists, ids, k):
    """k best of every row of candidate lists [nq, m] (dbhip_vec_topk_merge) -> (idx [nq, k], dist [nq, k])."""
    dists = np.ascontiguousarray(dists, dtype=np.float32)
    ids = np.ascontiguousarray(ids, dtype=np.uint32)
    nq, m = dists.shape
    d, i = DeviceBuffer.from_numpy(dists.reshape(-1)), DeviceBuffer.from_numpy(ids.reshape(-1))
    oi, od = DeviceBuffer(max(nq * k, 1) * 4), DeviceBuffer(max(nq * k, 1) * 4)
    check(lib().dbhip_vec_topk_merge(C.c_void_p(d.ptr), C.c_void_p(i.ptr), C.c_int64(m), nq, k, C.c_void_p(oi.ptr), C.c_void_p(od.ptr), None))
    return oi.to_numpy(np.uint32, nq * k).reshape(nq, k), od.to_numpy(np.float32, nq * k).reshape(nq, k)


class Comm:
    """dbhip_comm: the RCCL communicator behind the C-ABI (one rank per GPU). Comm.local() = a world of one without RCCL;
    Comm(rank, world, id) = ncclCommInitRank with the 128-byte id of Comm.unique_id() (drawn by rank 0, shipped by the host)."""

    def __init__(self, rank=0, world=1, unique_id=None):
        _ensure()
        self.h = C.c_void_p()
        self.rank, self.world = rank, world
        idbuf = (C.c_uint8 * 128)(*unique_id) if unique_id is not None else None
        check(lib().dbhip_comm_create(rank, world, idbuf, C.byref(self.h)))

    @classmethod
    def local(cls):
        return cls(0, 1, None)

    @classmethod
    def loopback(cls, group_id, rank, world):
        """one rank of an IN-PROCESS world (dbhip_comm_create_loopback): `world` host threads of this process, one Comm each"""
        _ensure()
        self = cls.__new__(cls)
        self.h = C.c_void_p()
        self.rank, self.world = rank, world
        check(lib().dbhip_comm_create_loopback(C.c_uint64(group_id), rank, world, C.byref(self.h)))
        return self

    @staticmethod
    def unique_id():
        _ensure()
        buf = (C.c_uint8 * 128)()
        check(lib().dbhip_comm_unique_id(buf))
        return bytes(buf)

    def abort(self):
        """dbhip_comm_abort: this rank gives up; the other ranks' waiting (and later) collectives return an error instead of hanging"""
        check(lib().dbhip_comm_abort(self.h))

    def exchange_allgather(self, table, max_rows=256, stream=None):
        check(lib().dbhip_groupby_exchange_allgather(table.h, self.h, C.c_int64(max_rows), stream))

    def exchange_alltoall(self, table, max_rows=256, stream=None):
        check(lib().dbhip_groupby_exchange_alltoall(table.h, self.h, C.c_int64(max_rows), stream))

    def exchange_block(self, cols, dest_index, stream=None):
        """The ABI-owned exchange of a block (dbhip_exchange_begin / _finish: DataBlock::scatter by destination, ONE grouped all-to-all
        of every column) -> (received Columns, source starts [world + 1]). `dest_index`: DeviceBuffer of u32 destinations < world."""
        n = cols[0].n
        x = C.c_void_p()
        rows = C.c_int64()
        check(lib().dbhip_exchange_begin(self.h, _cols(cols), len(cols), C.c_void_p(dest_index.ptr if dest_index is not None else None), C.c_int64(n),
                                         C.byref(rows), C.byref(x), stream))
        return self._finish_exchange(x, rows, cols, stream)

    def shuffle_exchange_block(self, keys, cols, stream=None):
        """dbhip_shuffle_exchange_begin + _finish: the hash shuffle as ONE plan call — destination = siphash64(keys) % world
        (flight_scatter_hash.rs), then the exchange of `cols`. -> (received Columns, source starts)"""
        x, rows = C.c_void_p(), C.c_int64()
        check(lib().dbhip_shuffle_exchange_begin(self.h, _cols(keys), len(keys), _cols(cols), len(cols), C.c_int64(cols[0].n), C.byref(rows),
                                                 C.byref(x), stream))
        return self._finish_exchange(x, rows, cols, stream)

    def sort_exchange_block(self, keys, bounds, cols, desc=None, nulls_first=None, stream=None):
        """dbhip_sort_exchange_begin + _finish: the range partition of the distributed sort as ONE plan call — partition = bounds that sort
        strictly before the row (sort_spill.rs), sent to rank partition % world (sort_exchange_injector.rs)."""
        nk = len(keys)
        d = (C.c_uint8 * nk)(*(desc or [0] * nk))
        nf = (C.c_uint8 * nk)(*(nulls_first or [0] * nk))
        nb = bounds[0].n if bounds else 0
        x, rows = C.c_void_p(), C.c_int64()
        check(lib().dbhip_sort_exchange_begin(self.h, _cols(keys), _cols(bounds) if bounds else None, d, nf, nk, C.c_int64(nb), _cols(cols), len(cols),
                                              C.c_int64(cols[0].n), C.byref(rows), C.byref(x), stream))
        return self._finish_exchange(x, rows, cols, stream)

    def _finish_exchange(self, x, rows, cols, stream):
        try:
            m = rows.value
            outs = [DeviceBuffer(((m + 63) // 64) * 8 + 8) if c.dtype == L.T_BOOL else DeviceBuffer(max(m, 1) * ELEM_SIZE[c.dtype] + 64) for c in cols]
            vouts = [DeviceBuffer(((m + 63) // 64) * 8 + 8) if c.validity is not None else None for c in cols]
            dp = (C.c_void_p * len(cols))(*[b.ptr for b in outs])
            vp = (C.c_void_p * len(cols))(*[b.ptr if b is not None else None for b in vouts])
            starts = (C.c_int64 * (self.world + 1))()
            # String columns with data buffers: the long values arrive packed in one buffer per column (buffer 0 of the received column)
            sbytes = (C.c_int64 * max(len(cols), 1))()
            check(lib().dbhip_exchange_string_bytes(x, sbytes))
            sbufs = [DeviceBuffer(sbytes[k] + 64) if sbytes[k] else None for k in range(len(cols))]
            sp = (C.c_void_p * max(len(cols), 1))(*[b.ptr if b is not None else None for b in sbufs])
            check(lib().dbhip_exchange_finish_strings(x, dp, vp, sp, starts, stream))
        finally:
            lib().dbhip_exchange_destroy(x)
        out = []
        for k, (c, o, v) in enumerate(zip(cols, outs, vouts)):
            if sbufs[k] is not None:
                ptrs = DeviceBuffer.from_numpy(np.array([sbufs[k].ptr], dtype=np.uint64))
                out.append(Column(c.dtype, m, o, v, c.precision, c.scale, buffers=ptrs, keep=(sbufs[k],)))
            else:
                out.append(Column(c.dtype, m, o, v, c.precision, c.scale))
        return out, list(starts)

    def topk_allgather(self, idx, dist, nq, k, row_offset, stream=None):
        """dbhip_vec_topk_allgather: per-shard top-k (DeviceBuffers: u32 local ids, f32 distances, [nq][k]) -> global top-k on every rank
        -> (numpy u32 [nq, k], numpy f32 [nq, k])"""
        oi, od = DeviceBuffer(max(nq * k, 1) * 4), DeviceBuffer(max(nq * k, 1) * 4)
        check(lib().dbhip_vec_topk_allgather(self.h, C.c_void_p(idx.ptr), C.c_void_p(dist.ptr), nq, k, C.c_uint64(row_offset), C.c_void_p(oi.ptr),
                                             C.c_void_p(od.ptr), stream))
        return oi.to_numpy(np.uint32, nq * k).reshape(nq, k), od.to_numpy(np.float32, nq * k).reshape(nq, k)

    def allgather(self, send_ptr, recv_ptr, bytes_per_rank, stream=None):
        check(lib().dbhip_comm_allgather(self.h, C.c_void_p(send_ptr), C.c_void_p(recv_ptr), C.c_int64(bytes_per_rank), stream))

    def alltoall(self, send_ptr, recv_ptr, bytes_per_peer, stream=None):
        check(lib().dbhip_comm_alltoall(self.h, C.c_void_p(send_ptr), C.c_void_p(recv_ptr), C.c_int64(bytes_per_peer), stream))

    def allreduce_sum_u64(self, send_ptr, recv_ptr, count, stream=None):
        check(lib().dbhip_comm_allreduce_sum_u64(self.h, C.c_void_p(send_ptr), C.c_void_p(recv_ptr), C.c_int64(count), stream))

    def destroy(self):
        if self.h:
            lib().dbhip_comm_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.destroy()
        except Exception:  # noqa: BLE001
            pass


def kmeans(distance_type, data, rows_per_cluster, normalize_input=False):
    """KMeans::compute on the device (dbhip_kmeans). data: float32 [rows, dim] numpy -> (assignments u32[rows], distances f32[rows], k, iterations)"""
    data = np.ascontiguousarray(data, dtype=np.float32)
    rows, dim = data.shape
    buf = DeviceBuffer.from_numpy(data)
    a, d = DeviceBuffer(rows * 4 + 64), DeviceBuffer(rows * 4 + 64)
    k, it = C.c_int64(), C.c_int32()
    check(lib().dbhip_kmeans(distance_type, C.c_void_p(buf.ptr), C.c_int64(rows), dim, C.c_int64(rows_per_cluster), int(bool(normalize_input)),
                             C.c_void_p(a.ptr), C.c_void_p(d.ptr), C.byref(k), C.byref(it), None))
    return a.to_numpy(np.uint32, rows), d.to_numpy(np.float32, rows), int(k.value), int(it.value)


def vec_kernel_f32(which, a, b):
    """VectorDistanceKernel::{dot (0), l2_squared (1), l1 (2)} over the rows of two float32 [n, dim] arrays"""
    a, b = np.ascontiguousarray(a, dtype=np.float32), np.ascontiguousarray(b, dtype=np.float32)
    n, dim = a.shape
    da, db, out = DeviceBuffer.from_numpy(a), DeviceBuffer.from_numpy(b), DeviceBuffer(max(n, 1) * 4 + 64)
    check(lib().dbhip_vec_kernel_f32(which, C.c_void_p(da.ptr), C.c_void_p(db.ptr), C.c_int64(n), dim, C.c_void_p(out.ptr), None))
    return out.to_numpy(np.float32, n)


class VectorIndex:
    """Exact device vector index (dbhip_vec_index_*; stands where HNSWIndex::{build, search} stands in the reference,
    hnsw_index/hnsw.rs:62-315): bf16 MFMA pre-filter with an error bound + exact f32 re-scoring."""

    def __init__(self, metric, base):
        _ensure()
        self.base = base  # keeps the borrowed f32 column alive
        self.h = C.c_void_p()
        check(lib().dbhip_vec_index_build(metric, C.c_void_p(base.data.ptr), C.c_int64(base.n), base.dim, C.byref(self.h), None))

    def search(self, queries, k):
        oi, od = DeviceBuffer(max(queries.n * k, 1) * 4), DeviceBuffer(max(queries.n * k, 1) * 4)
        check(lib().dbhip_vec_index_search(self.h, C.c_void_p(queries.data.ptr), queries.n, k, C.c_void_p(oi.ptr), C.c_void_p(od.ptr), None))
        return oi.to_numpy(np.uint32, queries.n * k).reshape(queries.n, k), od.to_numpy(np.float32, queries.n * k).reshape(queries.n, k)

    def destroy(self):
        if self.h:
            lib().dbhip_vec_index_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


class HnswIndex:
    """dbhip_hnsw_*: the reference's HNSW + u8-quantised index (hnsw_index/hnsw.rs:62-315). `base`: a DeviceVectors-like
    object with .data.ptr / .n / .dim (f32 rows on the device); it is only read while the index is being built."""

    def __init__(self, handle, n, dim):
        self.h, self.n, self.dim = handle, n, dim

    @classmethod
    def build(cls, metric, base, m=10, ef_construct=40, seed=1):
        _ensure()
        h = C.c_void_p()
        check(lib().dbhip_hnsw_build(C.c_void_p(base.data.ptr), C.c_int64(base.n), base.dim, metric, m, ef_construct, C.c_uint64(seed),
                                     C.byref(h), None))
        return cls(h, base.n, base.dim)

    @classmethod
    def build_sequential(cls, metric, base, levels, m=10, ef_construct=40):
        """the deterministic build: given levels, points linked one after the other, the reference's summation order"""
        _ensure()
        levels = np.ascontiguousarray(levels, dtype=np.int32)
        assert len(levels) == base.n
        h = C.c_void_p()
        check(lib().dbhip_hnsw_build_sequential(C.c_void_p(base.data.ptr), C.c_int64(base.n), base.dim, metric, m, ef_construct,
                                                levels.ctypes.data_as(C.c_void_p), C.byref(h), None))
        return cls(h, base.n, base.dim)

    @classmethod
    def from_graph(cls, metric, base, m, levels, lists, entry_point, entry_level):
        """lists: the link lists in point-major, level-minor order (HNSWIndex::open over a given graph)"""
        _ensure()
        levels = np.ascontiguousarray(levels, dtype=np.int32)
        nl = np.array([len(x) for x in lists], dtype=np.int32)
        flat = np.ascontiguousarray(np.concatenate([np.asarray(x, dtype=np.uint32) for x in lists]) if len(lists) and nl.sum() else np.zeros(1, np.uint32),
                                    dtype=np.uint32)
        h = C.c_void_p()
        check(lib().dbhip_hnsw_from_graph(C.c_void_p(base.data.ptr), C.c_int64(base.n), base.dim, metric, m, levels.ctypes.data_as(C.c_void_p),
                                          flat.ctypes.data_as(C.c_void_p), nl.ctypes.data_as(C.c_void_p), C.c_uint32(entry_point),
                                          C.c_int32(entry_level), C.byref(h), None))
        return cls(h, base.n, base.dim)

    @classmethod
    def open(cls, metric, encoded, alpha, offset, multiplier, n, dim, m, levels, lists, entry_point, entry_level):
        """HNSWIndex::open over the stored form (databend_amd.hnsw_format.open_index): `encoded` = the encoded_u8_data bytes"""
        _ensure()
        levels = np.ascontiguousarray(levels, dtype=np.int32)
        nl = np.array([len(x) for x in lists], dtype=np.int32)
        flat = np.ascontiguousarray(np.concatenate([np.asarray(x, dtype=np.uint32) for x in lists]) if len(lists) and nl.sum() else np.zeros(1, np.uint32),
                                    dtype=np.uint32)
        enc = DeviceBuffer.from_numpy(np.ascontiguousarray(encoded, dtype=np.uint8)) if n else DeviceBuffer(16)
        h = C.c_void_p()
        check(lib().dbhip_hnsw_open(C.c_void_p(enc.ptr), C.c_float(float(alpha)), C.c_float(float(offset)), C.c_float(float(multiplier)), C.c_int64(n), dim,
                                    metric, m, levels.ctypes.data_as(C.c_void_p), flat.ctypes.data_as(C.c_void_p), nl.ctypes.data_as(C.c_void_p),
                                    C.c_uint32(entry_point), C.c_int32(entry_level), C.byref(h), None))
        check(lib().dbhip_stream_sync(None))   # `enc` is only read while the index is being made
        return cls(h, n, dim)

    def export_graph(self):
        """-> (levels, lists, entry_point, entry_level)"""
        levels = np.zeros(max(self.n, 1), dtype=np.int32)
        nlists, ep, el = C.c_int64(), C.c_uint32(), C.c_int32()
        check(lib().dbhip_hnsw_export_graph(self.h, levels.ctypes.data_as(C.c_void_p), None, None, C.byref(nlists), C.byref(ep), C.byref(el), None))
        nl = np.zeros(max(nlists.value, 1), dtype=np.int32)
        check(lib().dbhip_hnsw_export_graph(self.h, None, None, nl.ctypes.data_as(C.c_void_p), None, None, None, None))
        flat = np.zeros(max(int(nl[:nlists.value].sum()), 1), dtype=np.uint32)
        check(lib().dbhip_hnsw_export_graph(self.h, None, flat.ctypes.data_as(C.c_void_p), None, None, None, None, None))
        lists, off = [], 0
        for c in nl[:nlists.value]:
            lists.append(flat[off:off + c].copy())
            off += int(c)
        return levels[:self.n], lists, ep.value, el.value

    def search(self, queries, limit):
        oi, od = DeviceBuffer(max(queries.n * limit, 1) * 4), DeviceBuffer(max(queries.n * limit, 1) * 4)
        check(lib().dbhip_hnsw_search(self.h, C.c_void_p(queries.data.ptr), queries.n, limit, C.c_void_p(oi.ptr), C.c_void_p(od.ptr), None))
        return (oi.to_numpy(np.uint32, queries.n * limit).reshape(queries.n, limit),
                od.to_numpy(np.float32, queries.n * limit).reshape(queries.n, limit))

    def scores(self, queries):
        out = DeviceBuffer(max(queries.n * self.n, 1) * 4)
        check(lib().dbhip_hnsw_scores(self.h, C.c_void_p(queries.data.ptr), queries.n, C.c_void_p(out.ptr), None))
        return out.to_numpy(np.float32, queries.n * self.n).reshape(queries.n, self.n)

    def meta(self):
        a, o, m, ad = C.c_float(), C.c_float(), C.c_float(), C.c_int32()
        check(lib().dbhip_hnsw_meta(self.h, C.byref(a), C.byref(o), C.byref(m), C.byref(ad)))
        return np.float32(a.value), np.float32(o.value), np.float32(m.value), ad.value

    def encoded(self):
        _, _, _, ad = self.meta()
        out = DeviceBuffer(max(self.n, 1) * (ad + 4))
        check(lib().dbhip_hnsw_encoded(self.h, C.c_void_p(out.ptr), None))
        return out.to_numpy(np.uint8, self.n * (ad + 4)).reshape(self.n, ad + 4)

    def destroy(self):
        if self.h:
            lib().dbhip_hnsw_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


def score_u8(is_l1, query, base):
    query = np.ascontiguousarray(query, dtype=np.uint8)
    base = np.ascontiguousarray(base, dtype=np.uint8)
    n, dim = base.shape
    q, b, out = DeviceBuffer.from_numpy(query), DeviceBuffer.from_numpy(base.reshape(-1)), DeviceBuffer(max(n, 1) * 4)
    check(lib().dbhip_score_u8(int(is_l1), C.c_void_p(q.ptr), C.c_void_p(b.ptr), C.c_int64(n), dim, C.c_void_p(out.ptr), None))
    return out.to_numpy(np.float32, n)


class ParquetChunk:
    """dbhip_pq_chunk_*: one Parquet column chunk -> one HBM-resident Column (fuse .../parquet/deserialize.rs:33-81).
    `chunk` = the raw bytes of the column chunk (dictionary page first), as the block reader fetched them."""

    def __init__(self, chunk, physical_type, out_type, type_length=0, max_def_level=0, max_rep_level=0, codec=0,
                 precision=0, scale=0, device=False, list_of=None):
        """device=True: dbhip_pq_chunk_open_device / _decode_device — the host reads the page headers only; decompression (ZSTD / SNAPPY /
        LZ4_RAW), run headers, length prefixes and DELTA blocks are walked on the GPU from the chunk as stored.
        list_of=(list_nullable, element_nullable): a List<primitive> leaf (dbhip_pq_chunk_open_device_list; decode with decode_list())."""
        _ensure()
        self.host = np.frombuffer(chunk, dtype=np.uint8)   # zero-copy view; the bytes object stays referenced by the array
        self.out_type, self.precision, self.scale = out_type, precision, scale
        self.device = bool(device) or list_of is not None
        self.list_of = list_of
        self.h = C.c_void_p()
        self.info = L.PqInfo()
        hp = self.host.ctypes.data_as(C.c_void_p) if len(self.host) else C.c_void_p(0)
        if list_of is not None:
            check(lib().dbhip_pq_chunk_open_device_list(hp, C.c_int64(len(self.host)), C.c_int32(codec), C.c_int32(physical_type), C.c_int32(type_length),
                                                        C.c_int32(1 if list_of[0] else 0), C.c_int32(1 if list_of[1] else 0), C.c_int32(out_type),
                                                        C.byref(self.h), C.byref(self.info)))
        else:
            fn = lib().dbhip_pq_chunk_open_device if self.device else lib().dbhip_pq_chunk_open
            check(fn(hp if len(self.host) else self.host.ctypes.data_as(C.c_void_p), C.c_int64(len(self.host)),
                     C.c_int32(codec), C.c_int32(physical_type), C.c_int32(type_length), C.c_int32(max_def_level),
                     C.c_int32(max_rep_level), C.c_int32(out_type), C.byref(self.h), C.byref(self.info)))
        self.chunk_dev = None
        self.image_dev = None
        self.nulls = self.info.num_nulls

    def image(self):
        """what decode() reads: the chunk itself, or (compressed chunks) the decompressed page stream open() produced"""
        if self.info.image_bytes == 0:
            return self.host
        p, n = C.POINTER(C.c_uint8)(), C.c_int64()
        check(lib().dbhip_pq_chunk_image(self.h, C.byref(p), C.byref(n)))
        return np.ctypeslib.as_array(p, shape=(n.value,))

    def upload(self):
        """the image's bytes into HBM (+ 16 bytes of slack; they become buffer 0 of a string column). Device mode: the chunk AS STORED."""
        if self.chunk_dev is None:
            src = self.host if self.device else self.image()
            self.chunk_dev = DeviceBuffer.from_numpy(np.concatenate([src, np.zeros(16, np.uint8)]))
        return self.chunk_dev

    def device_image(self):
        """device mode, after decode(): the decompressed pages as the GPU wrote them (host copy; UNCOMPRESSED chunks: the chunk)"""
        if self.image_dev is None:
            return self.host
        return self.image_dev.to_numpy(np.uint8, self.info.image_bytes)

    def decode(self, stream=None):
        i = self.info
        chunk_dev = self.upload()
        out = DeviceBuffer(i.out_bytes + 16)
        val = DeviceBuffer(i.validity_bytes + 8) if i.has_validity else None
        buf0 = chunk_dev
        if self.device:
            if i.image_bytes and self.image_dev is None:
                self.image_dev = DeviceBuffer(i.image_bytes)
            nulls = C.c_int64(0)
            check(lib().dbhip_pq_chunk_decode_device(self.h, C.c_void_p(chunk_dev.ptr), C.c_void_p(self.image_dev.ptr) if self.image_dev else None,
                                                     C.c_void_p(out.ptr), C.c_void_p(val.ptr) if val is not None else None,
                                                     C.byref(nulls), stream))
            self.nulls = nulls.value
            if self.image_dev is not None:
                buf0 = self.image_dev
        else:
            check(lib().dbhip_pq_chunk_decode(self.h, C.c_void_p(chunk_dev.ptr), C.c_void_p(out.ptr),
                                              C.c_void_p(val.ptr) if val is not None else None, stream))
        bufs = None
        if self.out_type == L.T_STRING:
            bufs = DeviceBuffer.from_numpy(np.array([buf0.ptr], dtype=np.uint64))
        return Column(self.out_type, i.num_values, out, val, self.precision, self.scale, buffers=bufs, keep=(buf0,))

    def decode_list(self, stream=None):
        """dbhip_pq_chunk_decode_device_list -> (offsets numpy u64 [rows + 1], list validity numpy bool [rows] or None, element Column)"""
        i = self.info
        chunk_dev = self.upload()
        if i.image_bytes and self.image_dev is None:
            self.image_dev = DeviceBuffer(i.image_bytes)
        offs = DeviceBuffer((i.num_values + 1) * 8 + 16)
        lval = DeviceBuffer(i.validity_bytes + 8) if self.list_of[0] else None
        out = DeviceBuffer(i.out_bytes + 16)
        eval_ = DeviceBuffer(i.validity_bytes + 8) if self.list_of[1] else None
        rows, elems, nl = C.c_int64(), C.c_int64(), C.c_int64()
        check(lib().dbhip_pq_chunk_decode_device_list(self.h, C.c_void_p(chunk_dev.ptr), C.c_void_p(self.image_dev.ptr) if self.image_dev else None,
                                                      C.c_void_p(offs.ptr), C.c_void_p(lval.ptr) if lval is not None else None, C.c_void_p(out.ptr),
                                                      C.c_void_p(eval_.ptr) if eval_ is not None else None, C.byref(rows), C.byref(elems), C.byref(nl), stream))
        self.rows, self.elems, self.null_lists = rows.value, elems.value, nl.value
        buf0 = self.image_dev if self.image_dev is not None else chunk_dev
        bufs = DeviceBuffer.from_numpy(np.array([buf0.ptr], dtype=np.uint64)) if self.out_type == L.T_STRING else None
        col = Column(self.out_type, elems.value, out, eval_, self.precision, self.scale, buffers=bufs, keep=(buf0,))
        lv = unpack_bits(lval.to_numpy(np.uint8, (rows.value + 7) // 8), rows.value) if lval is not None else None
        return offs.to_numpy(np.uint64, rows.value + 1), lv, col

    @staticmethod
    def decode_many(chunks, stream=None, statuses=None):
        """dbhip_pq_chunks_decode_device: the column chunks of a block (or of several blocks) decoded by ONE launch set. `chunks` were
        opened with device=True. -> [Column]; `statuses` (a list) receives the per-chunk status codes instead of an exception for a
        chunk that fails its device checks (its Column is None then)."""
        n = len(chunks)
        if n == 0:
            return []
        outs, vals = [], []
        for pc in chunks:
            assert pc.device, "decode_many takes device-mode chunks"
            i = pc.info
            pc.upload()
            if i.image_bytes and pc.image_dev is None:
                pc.image_dev = DeviceBuffer(i.image_bytes)
            outs.append(DeviceBuffer(i.out_bytes + 16))
            vals.append(DeviceBuffer(i.validity_bytes + 8) if i.has_validity else None)
        P = C.c_void_p * n
        hs = P(*[pc.h for pc in chunks])
        cd = P(*[C.c_void_p(pc.chunk_dev.ptr) for pc in chunks])
        im = P(*[C.c_void_p(pc.image_dev.ptr) if pc.image_dev is not None else C.c_void_p(0) for pc in chunks])
        ov = P(*[C.c_void_p(o.ptr) for o in outs])
        vv = P(*[C.c_void_p(v.ptr) if v is not None else C.c_void_p(0) for v in vals])
        nulls = (C.c_int64 * n)()
        st = (C.c_int32 * n)()
        rc = lib().dbhip_pq_chunks_decode_device(hs, C.c_int32(n), cd, im, ov, vv, nulls, st, stream)
        if statuses is None:
            check(rc)
        else:
            statuses[:] = list(st)
            if rc and all(x == 0 for x in st):
                check(rc)
        cols = []
        for k, pc in enumerate(chunks):
            if st[k]:
                cols.append(None)
                continue
            pc.nulls = nulls[k]
            buf0 = pc.image_dev if pc.image_dev is not None else pc.chunk_dev
            bufs = DeviceBuffer.from_numpy(np.array([buf0.ptr], dtype=np.uint64)) if pc.out_type == L.T_STRING else None
            cols.append(Column(pc.out_type, pc.info.num_values, outs[k], vals[k], pc.precision, pc.scale, buffers=bufs, keep=(buf0,)))
        return cols

    def close(self):
        if self.h:
            lib().dbhip_pq_chunk_close(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
