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
