"""Python (arbitrary-precision int) statement of the reference's decimal semantics for ALL storage classes, incl. Decimal256.
It is the third, independent statement next to oracle/decimal256.c (C, fixed 640-bit integers) and the device code
(databend_amd/csrc/dev_i256.h, 32-bit-limb long division): the CPU tests pin the oracle on it, the GPU tests pin the device
on the oracle.

Reference (src/query): functions/src/scalars/decimal/src/arithmetic.rs:80-316 (result_size, binary_decimal, unary minus
:514-590), cast.rs:701-753 (integer_to_decimal), :790-1035 (shrink / scale_reduction / expand / decimal_to_decimal),
comparison.rs:326-441; expression/src/types/decimal.rs:759-797 (i64), :1024-1060 (i128), :1343-1404 (i256 do_round_mul /
do_round_div), :1460-1487 (from_bigint).  Integer overflow wraps (Cargo.toml:577 overflow-checks = false).
"""

OP_PLUS, OP_MINUS, OP_MULTIPLY, OP_DIVIDE = 0, 1, 2, 3
MAXP = {64: 18, 128: 38, 256: 76}
INT_PROPS = {"i8": (3, 8, True), "u8": (3, 8, False), "i16": (5, 16, True), "u16": (5, 16, False), "i32": (10, 32, True),
             "u32": (10, 32, False), "i64": (19, 64, True), "u64": (20, 64, False)}


class RowError(Exception):
    pass


def wrap(v, bits):
    v &= (1 << bits) - 1
    return v - (1 << bits) if v >> (bits - 1) else v


def fits(v, bits):
    return -(1 << (bits - 1)) <= v < (1 << (bits - 1))


def tdiv(a, b):
    """Rust integer division: truncates toward zero"""
    q = abs(a) // abs(b)
    return -q if (a < 0) != (b < 0) else q


def trem(a, b):
    return a - tdiv(a, b) * b


def storage_bits(p):
    return 64 if p <= 18 else (128 if p <= 38 else 256)


def result_size(op, a, b):
    """a, b = (precision, scale) -> (left, right, ret) or None (arithmetic.rs:80-139)"""
    (ap, as_), (bp, bs) = a, b
    la, lb = ap - as_, bp - bs
    if op == OP_MULTIPLY:
        scale = min(as_ + bs, max(as_, bs, 12))
        precision = la + lb + scale
    elif op == OP_DIVIDE:
        scale = max(as_, min(as_ + 6, 12))
        precision = la + bs + scale
    else:
        scale = max(as_, bs)
        precision = max(la, lb) + scale + 1
    precision = min(precision, 38 if ap <= 38 and bp <= 38 else 76)
    if precision < 1 or scale > precision:
        return None
    if op == OP_MULTIPLY:
        left, right = (precision, as_), (precision, bs)
    elif op == OP_DIVIDE:
        pp = max(precision, ap, bp)
        left, right = (pp, as_), (pp, bs)
    else:
        left = right = (precision, scale)
    if left[1] > left[0] or right[1] > right[0]:
        return None
    return left, right, (precision, scale)


def from_bigint(v):
    """i256::from_bigint (decimal.rs:1460-1487) incl. its quirk: -2^255 comes back as DECIMAL_MIN"""
    mag = abs(v)
    if mag >> 256:
        return None
    if v > 0:
        return v if mag < (1 << 255) else None
    if v == 0:
        return 0
    if mag < (1 << 255):
        return v
    if mag == (1 << 255):
        return -(10 ** 76 - 1)
    return None


def round_mul(a, b, shift, overflow, bits):
    div, same = 10 ** shift, (a < 0) == (b < 0)
    half = div // 2
    if bits == 64:   # decimal.rs:759-786
        if not overflow:
            p = wrap(a * b, 64)
            return tdiv(wrap(p + half if same else p - half, 64), div)
        q = tdiv(a * b + half if same else a * b - half, div)   # computed in i128
        if q < -(10 ** 18 - 1) or q > 10 ** 18 - 1:
            raise RowError("Decimal multiply overflow")
        return q
    if bits == 128:  # decimal.rs:1024-1054
        if not overflow:
            p = wrap(a * b, 128)
            return tdiv(wrap(p + half if same else p - half, 128), div)
        q = tdiv(a * b + half if same else a * b - half, div)   # in i256
        if not fits(q, 128):
            raise RowError("Decimal multiply overflow")
        return q
    # i256, decimal.rs:1343-1376
    if not overflow:
        p = wrap(a * b, 256)
        return tdiv(wrap(p + half if same else p - half, 256), div)
    exact = a * b
    if fits(exact, 256):
        return tdiv(wrap(exact + half if same else exact - half, 256), div)
    r = from_bigint(tdiv(exact + half if same else exact - half, div))
    if r is None:
        raise RowError("Decimal multiply overflow")
    return r


def round_div(a, b, mul_scale, bits):
    same = (a < 0) == (b < 0)
    if bits == 64:   # decimal.rs:788-797: in i128, then `as i64`
        am = wrap(a * 10 ** mul_scale, 128)
        num = wrap(am + tdiv(b, 2) if same else am - tdiv(b, 2), 128)
        return wrap(tdiv(num, b), 64)
    if bits == 128:  # decimal.rs:1056-1064: low 128 bits of the i256 quotient
        am = a * 10 ** mul_scale
        num = am + tdiv(b, 2) if same else am - tdiv(b, 2)
        return wrap(tdiv(num, b), 128)

    def fallback():
        r = from_bigint(tdiv(a * 10 ** mul_scale + tdiv(b, 2) if same else a * 10 ** mul_scale - tdiv(b, 2), b))
        if r is None:
            raise RowError("Decimal div overflow")
        return r
    if mul_scale >= 76:
        return fallback()
    x = a * 10 ** mul_scale
    if not fits(x, 256):
        return fallback()
    s = wrap(x + tdiv(b, 2) if same else x - tdiv(b, 2), 256)
    return wrap(tdiv(s, b), 256)


def convert_operand(x, kind, from_scale, to, bits):
    """kind: "dec" or an integer type name. to = bound DecimalSize. bits = width of T."""
    tp, ts = to
    mx = 10 ** tp - 1
    if kind != "dec":
        if ts == 0:
            return wrap(x, bits)
        if not fits(x, bits):
            raise RowError("Decimal overflow")
        r = x * 10 ** ts
        if not fits(r, bits) or r > mx or r < -mx:
            raise RowError("Decimal overflow")
        return r
    if from_scale == ts:
        return wrap(x, bits)
    r = wrap(x, bits) * 10 ** (ts - from_scale)
    if not fits(r, bits) or r > mx or r < -mx:
        raise RowError("Decimal overflow")
    return r


def binary(op, x, xkind, xsize, y, ykind, ysize):
    """one row. xkind / ykind: "dec" or an integer type name; xsize = (p, s) of the operand (integers: their decimal
    properties). -> (value, (p, s)) ; raises RowError"""
    rs = result_size(op, xsize, ysize)
    assert rs is not None
    left, right, ret = rs
    bits = storage_bits(ret[0])
    overflow = ret[0] == MAXP[bits]
    a = convert_operand(x, xkind, xsize[1], left, bits)
    b = convert_operand(y, ykind, ysize[1], right, bits)
    if op in (OP_PLUS, OP_MINUS):
        t = wrap(a + b if op == OP_PLUS else a - b, bits)
        if overflow and (t < -(10 ** ret[0] - 1) or t > 10 ** ret[0] - 1):
            raise RowError("Decimal overflow")
        return t, ret
    if op == OP_MULTIPLY:
        sm = xsize[1] + ysize[1] - ret[1]
        if sm == 0:
            return wrap(a * b, bits), ret
        return round_mul(a, b, sm, overflow, bits), ret
    if b == 0:
        raise RowError("divided by zero")
    return round_div(a, b, ysize[1] + ret[1] - xsize[1], bits), ret


def negate(x, bits):
    return wrap(-x, bits)


def cmp3(a, asize, b, bsize):
    """DecimalCmp (comparison.rs:326-441) -> -1 / 0 / 1"""
    scale = max(asize[1], bsize[1])
    precision = max(asize[0] - asize[1], bsize[0] - bsize[1]) + scale
    precision = min(precision, 38 if asize[0] <= 38 and bsize[0] <= 38 else 76)
    bits = storage_bits(precision)
    a, b = wrap(a, bits), wrap(b, bits)
    fa, fb = 10 ** (scale - asize[1]), 10 ** (scale - bsize[1])
    c3 = lambda p, q: (p > q) - (p < q)  # noqa: E731
    if fa == fb:
        return c3(a, b)
    sa, sb = c3(a, 0), c3(b, 0)
    if sa != sb:
        return c3(a, b)
    if fa != 1:
        if not fits(a * fa, bits):
            return 1 if sa > 0 else -1
        a = a * fa
    if fb != 1:
        if not fits(b * fb, bits):
            return -1 if sb > 0 else 1
        b = b * fb
    return c3(a, b)


def scale_reduction(x, mx, factor, scale, scale_diff, rounding_mode, bits):
    q = tdiv(x, factor)
    rv = 0
    if rounding_mode and scale_diff != 0:
        m = trem(tdiv(x, 10 ** (scale_diff - 1)), 10)
        rv = 1 if m >= 5 else (-1 if m <= -5 else 0)
    y = q + rv
    if not fits(y, bits):
        raise RowError("Decimal overflow")
    int_part_zero = (x <= 10 ** scale - 1) if x >= 0 else (x >= -(10 ** scale - 1))
    if y > mx or y < -mx or (y == 0 and not int_part_zero):
        raise RowError("Decimal overflow")
    return y


def cast_decimal(x, src_bits, src, dst, rounding_mode=True):
    """decimal_to_decimal (cast.rs:981-1035): storage src_bits, DecimalSize src -> DecimalSize dst (storage by precision)"""
    (fp, fs), (dp, ds) = src, dst
    dbits = storage_bits(dp)
    mx = 10 ** dp - 1
    expand = src_bits == 64 or (src_bits == 128 and dbits >= 128) or (src_bits == 256 and dbits == 256)
    cbits = dbits if expand else src_bits   # the type the arithmetic runs in
    if expand and fs == ds and fp <= dp:
        return wrap(x, dbits)
    if ds == fs:
        if x > mx or x < -mx:
            raise RowError("Decimal overflow")
        return wrap(x, dbits)
    if ds > fs:
        r = x * 10 ** (ds - fs)
        if not fits(r, cbits) or r > mx or r < -mx:
            raise RowError("Decimal overflow")
        return wrap(r, dbits)
    return wrap(scale_reduction(x, mx, 10 ** (fs - ds), fs, fs - ds, rounding_mode, cbits), dbits)


def cast_integer(x, int_bits, dst):
    """integer_to_decimal (cast.rs:701-753)"""
    dp, ds = dst
    dbits = storage_bits(dp)
    mx = 10 ** dp - 1
    if ds == 0:
        return wrap(x, dbits)
    if not fits(x, dbits):
        raise RowError("Decimal overflow")
    r = x * 10 ** ds
    if not fits(r, dbits) or r > mx or r < -mx:
        raise RowError("Decimal overflow")
    return r
