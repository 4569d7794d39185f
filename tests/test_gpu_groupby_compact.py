"""GPU: the compact-row kernels of the hash aggregation (csrc/gb_compact.h, round 4) against the oracle's AggregateHashTable
(aggregate_hashtable.rs:168-333) through the C-ABI — every aggregate kind and argument type they accept, 1 and 2 keys, the no-partition
LDS path, the partitioned path (forced partition counts and the adaptive choice), tables that run full (spill), and the same cases
with the compact kernels switched off (the generic kernels must agree: both are checked against the oracle, not against each other)."""
import ctypes as C

import numpy as np
import pytest

from databend_amd import _lib as T
from tests import oracle_lib as O
from tests.test_gpu_parity import norm, oracle_groupby, oracle_rows

pytestmark = pytest.mark.gpu

NP = {T.T_I8: np.int8, T.T_I16: np.int16, T.T_I32: np.int32, T.T_I64: np.int64, T.T_U8: np.uint8, T.T_U16: np.uint16, T.T_U32: np.uint32,
      T.T_U64: np.uint64, T.T_F32: np.float32, T.T_F64: np.float64, T.T_DATE: np.int32, T.T_TIMESTAMP: np.int64, T.T_DEC64: np.int64}


def values(rng, t, n):
    if t in (T.T_F32, T.T_F64):
        return rng.integers(-1000, 1000, n).astype(NP[t])          # exactly representable: sums do not depend on the order
    info = np.iinfo(NP[t])
    lo, hi = max(info.min, -10**6), min(info.max, 10**6)
    return rng.integers(lo, hi, n, dtype=np.int64).astype(NP[t])


def set_compact(g, on):
    T.check(T.lib().dbhip_groupby_debug_set_compact(g.h, C.c_int32(1 if on else 0)))


def run_case(gpu, oracle, key_types, keys, aggs, args, n, compact=True, pbits=None, blocks=1):
    g = gpu.GroupBy(key_types, aggs, [0] * len(key_types))
    set_compact(g, compact)
    if pbits is not None:
        g.debug_set_partition_bits(pbits)
    cuts = [n * b // blocks for b in range(blocks + 1)]
    for lo, hi in zip(cuts[:-1], cuts[1:]):
        kc = [gpu.Column.from_numpy(k[lo:hi], t) for k, t in zip(keys, key_types)]
        ac = [gpu.Column.from_numpy(a[lo:hi], t) if a is not None else None for a, (_, t, *_r) in zip(args, aggs)]
        g.add_block(kc, ac, hi - lo)
    got = g.result()
    g.destroy()
    h = oracle_groupby(oracle, key_types, [0] * len(key_types), aggs, [O.HostCol(t, k) for k, t in zip(keys, key_types)],
                       [O.HostCol(t, a) if a is not None else None for a, (_, t, *_r) in zip(args, aggs)], n)
    exp = oracle_rows(oracle, h, key_types, aggs)
    oracle.orc_hashagg_destroy(h)
    assert norm(got) == norm(exp)
    return got


AGG_SETS = [
    ("sum+count", [(T.AGG_SUM, T.T_I64, 0, 0, 0), (T.AGG_COUNT, 0, 0, 0, 0)]),
    ("count only", [(T.AGG_COUNT, 0, 0, 0, 0)]),
    ("min max sum f64", [(T.AGG_MIN, T.T_I32, 0, 0, 0), (T.AGG_MAX, T.T_F64, 0, 0, 0), (T.AGG_SUM, T.T_F64, 0, 0, 0)]),
    ("four words", [(T.AGG_SUM, T.T_F32, 0, 0, 0), (T.AGG_MAX, T.T_U16, 0, 0, 0), (T.AGG_MIN, T.T_DATE, 0, 0, 0), (T.AGG_SUM, T.T_DEC64, 15, 2, 0)]),
    ("small ints", [(T.AGG_SUM, T.T_I8, 0, 0, 0), (T.AGG_SUM, T.T_U32, 0, 0, 0), (T.AGG_MIN, T.T_TIMESTAMP, 0, 0, 0)]),
    ("min f32 max i64", [(T.AGG_MIN, T.T_F32, 0, 0, 0), (T.AGG_MAX, T.T_I64, 0, 0, 0), (T.AGG_COUNT, T.T_I64, 0, 0, 0)]),
]


@pytest.mark.parametrize("name,aggs", AGG_SETS, ids=[a[0] for a in AGG_SETS])
@pytest.mark.parametrize("n,card,key_types", [(1, 1, [T.T_I64]), (5000, 37, [T.T_I32]), (300_000, 300, [T.T_I64]), (300_000, 2500, [T.T_U16, T.T_I64]),
                                              (400_000, 40_000, [T.T_I64]), (300_000, 250_000, [T.T_DATE, T.T_U8])])
def test_compact_kernels_match_the_oracle(gpu, oracle, name, aggs, n, card, key_types):
    rng = np.random.default_rng(n + card + len(aggs))
    base = rng.integers(0, card, n)
    if len(key_types) == 1:
        t = key_types[0]
        keys = [(base % 60000).astype(NP[t]) if t == T.T_U16 else (base * 7919 - 3).astype(NP[t])]
    else:       # the group is the pair (base // 200, base % 200)
        keys = [(base // 200).astype(NP[key_types[0]]), (base % 200).astype(NP[key_types[1]])]
    args = [values(rng, t, n) if (kind != T.AGG_COUNT or t) else None for kind, t, *_r in aggs]
    run_case(gpu, oracle, key_types, keys, aggs, args, n)


@pytest.mark.parametrize("compact", [True, False], ids=["compact", "generic"])
@pytest.mark.parametrize("pbits", [4, 7, 11, 14])
def test_forced_partition_counts_through_both_kernel_families(gpu, oracle, pbits, compact):
    """the partitioned path with 16 ... 16384 partitions (dbhip_groupby_debug_set_partition_bits), compact and generic kernels; three
    blocks, so that partial states of earlier blocks are merged into; key + sum + count and a two-key min / max layout"""
    n, card = 600_000, 30_000
    rng = np.random.default_rng(pbits)
    k = (rng.integers(0, card, n) * 104729).astype(np.int64)
    a = rng.integers(-10**9, 10**9, n).astype(np.int64)
    run_case(gpu, oracle, [T.T_I64], [k], [(T.AGG_SUM, T.T_I64, 0, 0, 0), (T.AGG_COUNT, 0, 0, 0, 0)], [a, None], n, compact=compact, pbits=pbits, blocks=3)
    k2 = rng.integers(0, 40, n).astype(np.int32)
    f = rng.integers(-500, 500, n).astype(np.float64)
    run_case(gpu, oracle, [T.T_I64, T.T_I32], [k % 1000, k2], [(T.AGG_MIN, T.T_F64, 0, 0, 0), (T.AGG_MAX, T.T_I64, 0, 0, 0), (T.AGG_SUM, T.T_I64, 0, 0, 0)],
             [f, a, a], n, compact=compact, pbits=pbits)


def test_skewed_keys_fill_the_tables_and_spill(gpu, oracle):
    """a partitioning chosen for few groups meets many: with 16 partitions forced, 200 K distinct keys overflow every workgroup's table;
    the rows that do not fit leave in table layout and go through the row path (or, past the spill buffer, the chunk is redone by the
    generic kernels) — the result is the oracle's either way. One heavy key takes half of the rows."""
    n = 1_500_000
    rng = np.random.default_rng(9)
    k = rng.integers(0, 200_000, n).astype(np.int64)
    k[rng.random(n) < 0.5] = 77
    a = rng.integers(-10**6, 10**6, n).astype(np.int64)
    aggs = [(T.AGG_SUM, T.T_I64, 0, 0, 0), (T.AGG_COUNT, 0, 0, 0, 0), (T.AGG_MAX, T.T_I64, 0, 0, 0)]
    got = run_case(gpu, oracle, [T.T_I64], [k], aggs, [a, None, a], n, pbits=4)
    heavy = [r for r in got if r[0] == 77]
    assert len(heavy) == 1 and heavy[0][2] == int((k == 77).sum())


@pytest.mark.parametrize("card", [9, 64, 200, 1000, 2400, 5000, 100_000, 1_500_000])
def test_adaptive_path_choice_over_the_cardinality_range(gpu, oracle, card):
    """one 6 M-row block per cardinality through plain add_block: few-groups kernel / compact LDS tables of growing size / partitioned /
    adaptive chunks are chosen by the library; sum + count against the oracle"""
    n = 6_000_000
    rng = np.random.default_rng(card)
    k = rng.integers(0, card, n).astype(np.int64)
    a = rng.integers(0, 1000, n).astype(np.int64)
    got = run_case(gpu, oracle, [T.T_I64], [k], [(T.AGG_SUM, T.T_I64, 0, 0, 0), (T.AGG_COUNT, 0, 0, 0, 0)], [a, None], n)
    assert sum(r[2] for r in got) == n


# ---- round 5: the layouts plans actually produce --------------------------------------------------------------------------------------
def _both(gpu, kind, data, validity=None):
    """one column for the device and the oracle: kind = a dbhip type, or ("dec128", p, s) / "str" """
    from databend_amd.device import make_views_general
    if kind == "str":
        v, buf = make_views_general(data)
        return T.T_STRING, gpu.Column.strings(data, validity=validity), O.HostCol(T.T_STRING, v, validity, buffers=[buf])
    if isinstance(kind, tuple):
        _, p, s = kind
        return T.T_DEC128, gpu.Column.decimal128(data, p, s, validity=validity), O.HostCol(T.T_DEC128, O.i128_array(data), validity, p, s)
    return kind, gpu.Column.from_numpy(data, kind, validity=validity), O.HostCol(kind, data, validity)


def run_layout(gpu, oracle, keys, key_nullable, aggs, args, n, compact=True, pbits=None, blocks=1):
    """keys / args: [(kind, data, validity or None)] (None = count(*)); slices the columns into `blocks` add_block calls"""
    key_types = [_both(gpu, k, d[:1], None)[0] for k, d, _v in keys]
    g = gpu.GroupBy(key_types, aggs, key_nullable)
    set_compact(g, compact)
    if pbits is not None:
        g.debug_set_partition_bits(pbits)
    cuts = [n * b // blocks for b in range(blocks + 1)]
    sl = lambda x, lo, hi: None if x is None else x[lo:hi]
    for lo, hi in zip(cuts[:-1], cuts[1:]):
        kc = [_both(gpu, k, d[lo:hi], sl(v, lo, hi))[1] for k, d, v in keys]
        ac = [None if a is None else _both(gpu, a[0], a[1][lo:hi], sl(a[2], lo, hi))[1] for a in args]
        g.add_block(kc, ac, hi - lo)
    got = g.result()
    assert g.num_groups() == len(got)
    g.destroy()
    hk = [_both(gpu, k, d, v)[2] for k, d, v in keys]
    ha = [None if a is None else _both(gpu, a[0], a[1], a[2])[2] for a in args]
    h = oracle_groupby(oracle, key_types, key_nullable, aggs, hk, ha, n)
    exp = oracle_rows(oracle, h, key_types, aggs)
    oracle.orc_hashagg_destroy(h)
    assert norm(got) == norm(exp)
    return got


def layout_cases(rng, n, card):
    """name -> (keys, key_nullable, aggs, args)"""
    base = rng.integers(0, card, n)
    kv = rng.random(n) > 0.06            # 6 % NULL keys
    av = rng.random(n) > 0.25            # 25 % NULL arguments
    av2 = rng.random(n) > 0.5
    i64 = (base * 7919 - 3).astype(np.int64)
    a1 = rng.integers(-10**6, 10**6, n).astype(np.int64)
    a2 = rng.integers(-1000, 1000, n).astype(np.int32)
    f = rng.integers(-500, 500, n).astype(np.float64)
    f32 = rng.integers(-500, 500, n).astype(np.float32)
    d128 = [int(x) * 10**11 + 7 for x in rng.integers(-10**17, 10**17, n)]        # |v| up to 1e28: two words, both signs
    c3 = max(2, round(card ** (1 / 3)))
    k3 = [(base % c3).astype(np.int64), ((base // c3) % c3).astype(np.int32), (base // (c3 * c3)).astype(np.int32)]
    strs = [b"k%d" % x for x in base]                                            # inline views (<= 12 bytes)
    strs2 = [b"F" if x & 1 else b"O" for x in base]
    SUM, CNT, MIN, MAX = T.AGG_SUM, T.AGG_COUNT, T.AGG_MIN, T.AGG_MAX
    return {
        "nullable key, nullable sum + count(col) + count(*)": (
            [(T.T_I64, i64, kv)], [1], [(SUM, T.T_I64, 0, 0, 1), (CNT, T.T_I64, 0, 0, 1), (CNT, 0, 0, 0, 0)], [(T.T_I64, a1, av), (T.T_I64, a1, av), None]),
        "nullable min max over two nullable columns": (
            [(T.T_I32, i64.astype(np.int32), None)], [0], [(MIN, T.T_I64, 0, 0, 1), (MAX, T.T_F64, 0, 0, 1), (SUM, T.T_F64, 0, 0, 1), (MAX, T.T_I64, 0, 0, 1)],
            [(T.T_I64, a1, av), (T.T_F64, f, av2), (T.T_F64, f, av2), (T.T_I64, a1, av)]),
        "three keys (Q3's group-by), sum(Decimal128)": (
            [(T.T_I64, k3[0], None), (T.T_DATE, k3[1], None), (T.T_I32, k3[2], None)], [0, 0, 0], [(SUM, T.T_DEC128, 31, 4, 0)], [(("dec128", 31, 4), d128, None)]),
        "three keys, one nullable; nullable Decimal128 sum + min": (
            [(T.T_I64, k3[0], kv), (T.T_DATE, k3[1], None), (T.T_I32, k3[2], None)], [1, 0, 0], [(SUM, T.T_DEC128, 31, 4, 1), (MIN, T.T_I32, 0, 0, 0), (CNT, 0, 0, 0, 0)],
            [(("dec128", 31, 4), d128, av), (T.T_I32, a2, None), None]),
        "Decimal128 key; sum, count": (
            [(("dec128", 38, 0), [int(x) * 10**20 - 5 for x in base], None)], [0], [(SUM, T.T_I64, 0, 0, 0), (CNT, 0, 0, 0, 0)], [(T.T_I64, a1, None), None]),
        "nullable Decimal128 key + i16 key; max(u16)": (
            [(("dec128", 38, 0), [int(x) * 10**20 - 5 for x in base // 7], kv), (T.T_I16, (base % 7).astype(np.int16), None)], [1, 0],
            [(MAX, T.T_U16, 0, 0, 0), (SUM, T.T_F32, 0, 0, 0)], [(T.T_U16, (a2 & 0xffff).astype(np.uint16), None), (T.T_F32, f32, None)]),
        "inline String key; sum, count": (
            [("str", strs, None)], [0], [(SUM, T.T_I64, 0, 0, 0), (CNT, 0, 0, 0, 0)], [(T.T_I64, a1, None), None]),
        "two String keys, Q1's six aggregates": (
            [("str", strs, None), ("str", strs2, None)], [0, 0],
            [(SUM, T.T_DEC64, 15, 2, 0), (SUM, T.T_DEC64, 15, 2, 0), (SUM, T.T_DEC128, 31, 4, 0), (SUM, T.T_DEC128, 38, 6, 0), (SUM, T.T_DEC64, 15, 2, 0), (CNT, 0, 0, 0, 0)],
            [(T.T_DEC64, a1, None), (T.T_DEC64, a1 * 3, None), (("dec128", 31, 4), d128, None), (("dec128", 38, 6), [x * 3 for x in d128], None), (T.T_DEC64, a2.astype(np.int64), None), None]),
        "eight aggregates over two columns": (
            [(T.T_I64, i64, None)], [0],
            [(SUM, T.T_I64, 0, 0, 0), (SUM, T.T_I32, 0, 0, 0), (MIN, T.T_I64, 0, 0, 0), (MAX, T.T_I64, 0, 0, 0), (SUM, T.T_I64, 0, 0, 0), (MIN, T.T_I32, 0, 0, 0), (MAX, T.T_I32, 0, 0, 0),
             (CNT, 0, 0, 0, 0)],
            [(T.T_I64, a1, None), (T.T_I32, a2, None), (T.T_I64, a1, None), (T.T_I64, a1, None), (T.T_I64, a1, None), (T.T_I32, a2, None), (T.T_I32, a2, None), None]),
        "eight aggregates over six columns, some nullable": (
            [(T.T_U16, (base % 60000).astype(np.uint16), None), (T.T_I64, (base // 60000).astype(np.int64), kv)], [0, 1],
            [(SUM, T.T_I64, 0, 0, 1), (SUM, T.T_F64, 0, 0, 0), (MIN, T.T_F32, 0, 0, 0), (MAX, T.T_I32, 0, 0, 1), (SUM, T.T_DEC128, 31, 4, 0), (CNT, T.T_I32, 0, 0, 1), (MIN, T.T_I64, 0, 0, 1),
             (SUM, T.T_I32, 0, 0, 0)],
            [(T.T_I64, a1, av), (T.T_F64, f, None), (T.T_F32, f32, None), (T.T_I32, a2, av2), (("dec128", 31, 4), d128, None), (T.T_I32, a2, av2), (T.T_I64, a1, av), (T.T_I32, a2, None)]),
    }


LAYOUT_NAMES = list(layout_cases(np.random.default_rng(0), 8, 3).keys())


@pytest.mark.parametrize("name", LAYOUT_NAMES)
@pytest.mark.parametrize("n,card", [(1, 1), (4000, 30), (200_000, 900), (250_000, 30_000)])
def test_compact_kernels_on_the_layouts_plans_produce(gpu, oracle, name, n, card):
    """nullable keys and arguments, 16-byte keys, three keys, eight aggregates, Decimal128 sums: the LDS path (and, at 30 K groups, whatever
    the library chooses) against the oracle's AggregateHashTable"""
    rng = np.random.default_rng(n + card)
    keys, kn, aggs, args = layout_cases(rng, n, card)[name]
    run_layout(gpu, oracle, keys, kn, aggs, args, n, blocks=2 if n > 1000 else 1)


@pytest.mark.parametrize("name", LAYOUT_NAMES)
@pytest.mark.parametrize("pbits", [4, 11])
def test_compact_layouts_through_forced_partitionings(gpu, oracle, name, pbits):
    """the same layouts through scatter (histogram-less at 16 partitions, histogram + scans at 2048) -> per-partition tables -> merge, three
    blocks so that earlier partial states are merged into; and once more with the compact kernels off (the generic kernels agree with the oracle too)"""
    n, card = 300_000, 20_000
    rng = np.random.default_rng(pbits)
    keys, kn, aggs, args = layout_cases(rng, n, card)[name]
    run_layout(gpu, oracle, keys, kn, aggs, args, n, pbits=pbits, blocks=3)
    if pbits == 4:
        run_layout(gpu, oracle, keys, kn, aggs, args, n, pbits=pbits, compact=False)


def test_compact_layouts_spill_and_long_strings_leave_the_fast_path(gpu, oracle):
    """(a) a wide layout whose groups overflow the tables of a forced 16-way partitioning: rows leave in table layout (positions queue,
    Decimal128 states, not-NULL bits) and the result is the oracle's; (b) a String key that turns out longer than an inline view: the compact
    kernels raise the long-string bit, nothing of the chunk is merged, the row path takes the block"""
    n = 600_000
    rng = np.random.default_rng(3)
    keys, kn, aggs, args = layout_cases(rng, n, 150_000)["eight aggregates over six columns, some nullable"]
    run_layout(gpu, oracle, keys, kn, aggs, args, n, pbits=4)
    ids = rng.integers(0, 500, n)
    strs = [b"group-%d" % x for x in ids]
    long_key = b"a key that does not fit an inline view"
    strs[n // 2] = long_key
    a = rng.integers(0, 100, n).astype(np.int64)
    g = gpu.GroupBy([T.T_STRING], [(T.AGG_SUM, T.T_I64, 0, 0, 0), (T.AGG_COUNT, 0, 0, 0, 0)], [0])
    g.add_block([gpu.Column.strings(strs)], [gpu.Column.from_numpy(a), None], n)
    got = {r[0]: (r[1], r[2]) for r in g.result()}
    g.destroy()
    exp = {}
    for s_, v in zip(strs, a.tolist()):          # (the oracle's result decoder reads inline views only: a plain dictionary is the reference here)
        t = exp.get(s_, (0, 0))
        exp[s_] = (t[0] + v, t[1] + 1)
    assert got == exp and got[long_key][1] == 1


@pytest.mark.parametrize("compact", [True, False], ids=["compact", "generic"])
@pytest.mark.parametrize("card,null", [(30_000, True), (30_000, False), (1_500_000, True), (400_000, False)])
def test_one_heavy_group_among_many(gpu, oracle, card, null, compact):
    """a quarter of the rows carry ONE key (NULL, or a value): its partition is worked on in sub-ranges by extra workgroups
    (gbc_split_map_kernel) whose partial rows travel in the packed list — next to per-partition lists (10^5+ groups: one workgroup per
    partition) and inside the one packed list (few partitions, several workgroups each); the compact kernels and the generic ones"""
    n = 6_000_000
    rng = np.random.default_rng(card)
    k = rng.integers(0, card, n).astype(np.int64)
    heavy = rng.random(n) < 0.25
    a = rng.integers(-1000, 1000, n).astype(np.int64)
    if null:
        keys, kn = [(T.T_I64, k, ~heavy)], [1]
    else:
        k[heavy] = card + 11
        keys, kn = [(T.T_I64, k, None)], [0]
    got = run_layout(gpu, oracle, keys, kn, [(T.AGG_SUM, T.T_I64, 0, 0, 0), (T.AGG_COUNT, 0, 0, 0, 0), (T.AGG_MIN, T.T_I64, 0, 0, 0)],
                     [(T.T_I64, a, None), None, (T.T_I64, a, None)], n, compact=compact)
    hk = None if null else card + 11
    row = [r for r in got if r[0] == hk]
    assert len(row) == 1 and row[0][2] == int(heavy.sum())


@pytest.mark.parametrize("card", [50, 40_000])
def test_float_sums_agree_with_the_oracle_within_the_stated_tolerance(gpu, oracle, card):
    """DESIGN §3's only carve-out: sum() over Float64 / Float32 adds in an order neither the reference nor the device fixes — per group
    |device - oracle| <= 1e-9 x sum(|x|); counts and the integer sum beside them are exact."""
    n = 300_000
    rng = np.random.default_rng(card)
    k = rng.integers(0, card, n).astype(np.int64)
    x = rng.standard_normal(n) * 10.0 ** rng.integers(-3, 6, n)
    x32 = rng.standard_normal(n).astype(np.float32)
    i = rng.integers(-10**6, 10**6, n).astype(np.int64)
    aggs = [(T.AGG_SUM, T.T_F64, 0, 0, 0), (T.AGG_SUM, T.T_F32, 0, 0, 0), (T.AGG_SUM, T.T_I64, 0, 0, 0), (T.AGG_COUNT, 0, 0, 0, 0)]
    g = gpu.GroupBy([T.T_I64], aggs, [0])
    g.add_block([gpu.Column.from_numpy(k)], [gpu.Column.from_numpy(x), gpu.Column.from_numpy(x32), gpu.Column.from_numpy(i), None], n)
    got = {r[0]: r[1:] for r in g.result()}
    g.destroy()
    h = oracle_groupby(oracle, [T.T_I64], [0], aggs, [O.HostCol(T.T_I64, k)], [O.HostCol(T.T_F64, x), O.HostCol(T.T_F32, x32), O.HostCol(T.T_I64, i), None], n)
    exp = {r[0]: r[1:] for r in oracle_rows(oracle, h, [T.T_I64], aggs)}
    oracle.orc_hashagg_destroy(h)
    assert got.keys() == exp.keys()
    mag = np.zeros(card); np.add.at(mag, k, np.abs(x))
    mag32 = np.zeros(card); np.add.at(mag32, k, np.abs(x32.astype(np.float64)))
    for key, (s64, s32, si, c) in got.items():
        e64, e32, ei, ec = exp[key]
        assert (si, c) == (ei, ec)
        assert abs(s64 - e64) <= 1e-9 * mag[key] and abs(s32 - e32) <= 1e-9 * mag32[key]


@pytest.mark.parametrize("name", ["nullable key, nullable sum + count(col) + count(*)", "three keys (Q3's group-by), sum(Decimal128)",
                                  "eight aggregates over six columns, some nullable", "inline String key; sum, count"])
@pytest.mark.parametrize("n,card,pbits", [(4000, 30, None), (250_000, 900, None), (300_000, 20_000, 4), (300_000, 20_000, 11), (2_000_000, 150_000, None)])
def test_compact_kernels_with_a_pushed_down_filter(gpu, oracle, name, n, card, pbits):
    """dbhip_groupby_add_block_filtered (the TransformFilter in front of the aggregate pushed down as a Bitmap, filter_executor.rs:81-118)
    through the compact kernels: LDS tables, both scatters, the aggregation of partitions — equal to the oracle over the rows that pass."""
    rng = np.random.default_rng(n + card)
    keys, kn, aggs, args = layout_cases(rng, n, card)[name]
    keep = rng.random(n) < 0.6
    keep[: n // 50] = False                      # a run of rejected rows (whole tiles fail)
    key_types = [_both(gpu, k, d[:1], None)[0] for k, d, _v in keys]
    g = gpu.GroupBy(key_types, aggs, kn)
    if pbits is not None:
        g.debug_set_partition_bits(pbits)
    half = n // 2
    for lo, hi in ((0, half), (half, n)):
        sl = lambda x: None if x is None else x[lo:hi]
        kc = [_both(gpu, k, d[lo:hi], sl(v))[1] for k, d, v in keys]
        ac = [None if a is None else _both(gpu, a[0], a[1][lo:hi], sl(a[2]))[1] for a in args]
        if hi > lo:
            g.add_block(kc, ac, hi - lo, filter=gpu.Column.boolean(keep[lo:hi]))
    got = g.result()
    g.destroy()
    idx = np.flatnonzero(keep)
    pick = lambda d: [d[i] for i in idx] if isinstance(d, list) else d[idx]
    hk = [_both(gpu, k, pick(d), None if v is None else v[idx])[2] for k, d, v in keys]
    ha = [None if a is None else _both(gpu, a[0], pick(a[1]), None if a[2] is None else a[2][idx])[2] for a in args]
    h = oracle_groupby(oracle, key_types, kn, aggs, hk, ha, len(idx))
    exp = oracle_rows(oracle, h, key_types, aggs)
    oracle.orc_hashagg_destroy(h)
    assert norm(got) == norm(exp)
