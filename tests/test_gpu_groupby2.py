"""GPU: round-2 additions to the hash-aggregation path, against the oracle through the C-ABI —
pushed-down filter (add_block_filtered), the "seen a non-NULL row" flag of nullable sum/min/max, the serialized-state
block in BOTH directions (device flush -> oracle merge, oracle flush -> device merge), and the device hash partitioning
for the exchange (a12)."""
import ctypes as C

import numpy as np
import pytest

from databend_amd import _lib as T
from databend_amd.device import ELEM_SIZE, NP_OF, make_views_general, pack_bits, unpack_bits
from tests import oracle_lib as O
from tests.test_gpu_parity import norm, oracle_groupby, oracle_rows

pytestmark = pytest.mark.gpu


def mix(x):
    M = (1 << 64) - 1
    x &= M
    x ^= x >> 32
    x = (x * 0xD6E8FEB86659FD93) & M
    x ^= x >> 32
    x = (x * 0xD6E8FEB86659FD93) & M
    x ^= x >> 32
    return x


# ---------------------------------------------------------------------------------------------------------------
# pushed-down filter
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n,card,pbits,wide", [(1, 1, 0, False), (1000, 4, 0, False), (300_000, 4, 0, False), (300_000, 700, 0, False),
                                               (300_000, 5000, 0, False), (400_000, 390_000, 0, False), (300_000, 5000, 6, False),
                                               (300_000, 40, 4, True), (200_000, 150_000, 0, True), (100_000, 1000, 10, True)])
@pytest.mark.parametrize("keep", [0.0, 0.03, 0.986, 1.0])
def test_groupby_filtered_equals_groupby_of_the_taken_rows(gpu, oracle, n, card, pbits, wide, keep):
    """dbhip_groupby_add_block_filtered(block, Bitmap) == add_block(take(block, select(Bitmap))): on the LDS path, the
    radix-partitioned path and the row path, short and wide layouts, with selectivities from 'nothing' to Q1's 98.6 %."""
    rng = np.random.default_rng(n * 7 + card + int(keep * 1000))
    k = (rng.integers(0, card, n).astype(np.int64) * 2654435761) % (1 << 40)
    a = rng.integers(-10**9, 10**9, n).astype(np.int64)
    av = rng.integers(0, 5, n) > 0
    passes = rng.random(n) < keep
    key_types, key_nullable = [T.T_I64], [0]
    aggs = [(T.AGG_SUM, T.T_I64, 0, 0, 1), (T.AGG_COUNT, 0, 0, 0, 0)]
    gkeys, hk = [gpu.Column.from_numpy(k)], [k]
    gargs, ha = [gpu.Column.from_numpy(a, validity=av), None], [(T.T_I64, a, av), None]
    if wide:
        k2 = rng.integers(0, 3, n).astype(np.int32)
        d = [int(x) * 10**9 for x in rng.integers(-10**17, 10**17, n)]
        key_types, key_nullable = [T.T_I64, T.T_DATE, T.T_I64, T.T_I64, T.T_I64], [0, 0, 0, 0, 0]
        aggs += [(T.AGG_SUM, T.T_DEC128, 31, 4, 1), (T.AGG_MIN, T.T_I64, 0, 0, 1), (T.AGG_MAX, T.T_I64, 0, 0, 0)]
        gkeys += [gpu.Column.from_numpy(k2, T.T_DATE), gpu.Column.from_numpy(k), gpu.Column.from_numpy(k), gpu.Column.from_numpy(k)]
        gargs += [gpu.Column.decimal128(d, 31, 4, validity=av), gpu.Column.from_numpy(a, validity=av), gpu.Column.from_numpy(a)]
    g = gpu.GroupBy(key_types, aggs, key_nullable)
    if pbits:
        g.debug_set_partition_bits(pbits)
    g.add_block(gkeys, gargs, n, filter=gpu.Column.boolean(passes))
    got = g.result()
    # oracle over the taken rows
    sel = np.nonzero(passes)[0]
    m = len(sel)
    hkeys = [O.HostCol(T.T_I64, k[sel])]
    hargs = [O.HostCol(T.T_I64, a[sel], av[sel]), None]
    if wide:
        hkeys += [O.HostCol(T.T_DATE, k2[sel]), O.HostCol(T.T_I64, k[sel]), O.HostCol(T.T_I64, k[sel]), O.HostCol(T.T_I64, k[sel])]
        hargs += [O.HostCol(T.T_DEC128, O.i128_array([d[i] for i in sel]), av[sel], 31, 4), O.HostCol(T.T_I64, a[sel], av[sel]), O.HostCol(T.T_I64, a[sel])]
    if m == 0:
        assert got == [] and g.num_groups() == 0
        return
    h = oracle_groupby(oracle, key_types, key_nullable, aggs, hkeys, hargs, m)
    exp = oracle_rows(oracle, h, key_types, aggs)
    oracle.orc_hashagg_destroy(h)
    assert g.num_groups() == len(exp)
    assert norm(got) == norm(exp)


def test_q1_pushdown_plan_equals_fused_and_oracle(gpu, oracle):
    from databend_amd import tpch
    for n in (1, 129, 300_007):
        host = tpch.gen_lineitem(n, seed=n)
        li = tpch.LineitemDevice(host)
        exp = O.q1_run(host, tpch.Q1_CUTOFF, threads=1)
        assert tpch.q1_rows(tpch.q1_operator_pushdown(li)) == exp
        assert tpch.q1_rows(tpch.q1_fused(li)) == exp


# ---------------------------------------------------------------------------------------------------------------
# nullable sum / min / max: NULL for groups that never saw a value (AggregateNullUnaryAdaptor<true>)
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n,card", [(10, 3), (5000, 40), (300_000, 2000), (200_000, 150_000)])
def test_nullable_sum_min_max_yield_null_for_all_null_groups(gpu, oracle, n, card):
    rng = np.random.default_rng(n + card)
    k = rng.integers(0, card, n).astype(np.int32)
    a = rng.integers(-10**6, 10**6, n).astype(np.int64)
    f = rng.integers(-100, 100, n).astype(np.float64)
    d = [int(x) * 10**9 for x in rng.integers(-10**17, 10**17, n)]
    av = (rng.integers(0, 3, n) > 0) & (k % 3 != 0)      # every group with key % 3 == 0 sees only NULLs
    key_types, key_nullable = [T.T_I32], [0]
    aggs = [(T.AGG_SUM, T.T_I64, 0, 0, 1), (T.AGG_SUM, T.T_F64, 0, 0, 1), (T.AGG_SUM, T.T_DEC128, 31, 4, 1), (T.AGG_MIN, T.T_I64, 0, 0, 1),
            (T.AGG_MAX, T.T_F64, 0, 0, 1), (T.AGG_COUNT, T.T_I64, 0, 0, 1), (T.AGG_SUM, T.T_I64, 0, 0, 0)]
    g = gpu.GroupBy(key_types, aggs, key_nullable)
    g.add_block([gpu.Column.from_numpy(k)], [gpu.Column.from_numpy(a, validity=av), gpu.Column.from_numpy(f, validity=av),
                                             gpu.Column.decimal128(d, 31, 4, validity=av), gpu.Column.from_numpy(a, validity=av),
                                             gpu.Column.from_numpy(f, validity=av), gpu.Column.from_numpy(a, validity=av), gpu.Column.from_numpy(a)], n)
    got = g.result()
    h = oracle_groupby(oracle, key_types, key_nullable, aggs, [O.HostCol(T.T_I32, k)],
                       [O.HostCol(T.T_I64, a, av), O.HostCol(T.T_F64, f, av), O.HostCol(T.T_DEC128, O.i128_array(d), av, 31, 4),
                        O.HostCol(T.T_I64, a, av), O.HostCol(T.T_F64, f, av), O.HostCol(T.T_I64, a, av), O.HostCol(T.T_I64, a)], n)
    exp = oracle_rows(oracle, h, key_types, aggs)
    oracle.orc_hashagg_destroy(h)
    assert norm(got) == norm(exp)
    nulls = [r for r in got if r[0] % 3 == 0]
    assert nulls and all(r[1] is None and r[2] is None and r[3] is None and r[4] is None and r[5] is None and r[6] == 0 for r in nulls)
    assert all(r[1] is not None for r in got if r[0] % 3 != 0 and r[6] > 0)


# ---------------------------------------------------------------------------------------------------------------
# serialized-state block (payload_flush.rs:151-181), both directions against the oracle
# ---------------------------------------------------------------------------------------------------------------
SB_KEYS = ([T.T_I64, T.T_STRING], [1, 0])
SB_AGGS = [(T.AGG_SUM, T.T_I64, 0, 0, 0), (T.AGG_COUNT, 0, 0, 0, 0), (T.AGG_SUM, T.T_DEC128, 31, 4, 1), (T.AGG_SUM, T.T_F64, 0, 0, 1),
           (T.AGG_MIN, T.T_I32, 0, 0, 0), (T.AGG_MAX, T.T_F64, 0, 0, 1), (T.AGG_COUNT, T.T_I64, 0, 0, 1), (T.AGG_SUM, T.T_DEC64, 15, 2, 1),
           (T.AGG_MAX, T.T_DATE, 0, 0, 0), (T.AGG_MIN, T.T_DEC128, 31, 4, 1), (T.AGG_MAX, T.T_DEC128, 31, 4, 0)]   # r03: min / max over Decimal128


def sb_data(n, card, seed):
    rng = np.random.default_rng(seed)
    d = dict(k=rng.integers(0, card, n).astype(np.int64), kv=rng.integers(0, 9, n) > 0, s=[b"g%d" % (x % 7) for x in rng.integers(0, card, n)],
             a=rng.integers(-10**9, 10**9, n).astype(np.int64), dd=[int(x) * 10**9 for x in rng.integers(-10**17, 10**17, n)],
             f=rng.integers(-1000, 1000, n).astype(np.float64), i=rng.integers(-2**31, 2**31 - 1, n).astype(np.int32),
             dt=rng.integers(8000, 12000, n).astype(np.int32), dec=rng.integers(-10**14, 10**14, n).astype(np.int64))
    d["av"] = (rng.integers(0, 3, n) > 0) & (d["k"] % 4 != 0)
    return d


def sb_gpu_cols(gpu, d, lo, hi):
    sl = slice(lo, hi)
    av = d["av"][sl]
    keys = [gpu.Column.from_numpy(d["k"][sl], validity=d["kv"][sl]), gpu.Column.strings(d["s"][sl])]
    args = [gpu.Column.from_numpy(d["a"][sl]), None, gpu.Column.decimal128(d["dd"][sl], 31, 4, validity=av), gpu.Column.from_numpy(d["f"][sl], validity=av),
            gpu.Column.from_numpy(d["i"][sl]), gpu.Column.from_numpy(d["f"][sl], validity=av), gpu.Column.from_numpy(d["a"][sl], validity=av),
            gpu.Column.from_numpy(d["dec"][sl], T.T_DEC64, validity=av, precision=15, scale=2), gpu.Column.from_numpy(d["dt"][sl], T.T_DATE),
            gpu.Column.decimal128(d["dd"][sl], 31, 4, validity=av), gpu.Column.decimal128(d["dd"][sl], 31, 4)]
    return keys, args


def sb_host_cols(d, lo, hi):
    sl = slice(lo, hi)
    av = d["av"][sl]
    v, buf = make_views_general(d["s"][sl])
    keys = [O.HostCol(T.T_I64, d["k"][sl], d["kv"][sl]), O.HostCol(T.T_STRING, v, buffers=[buf])]
    args = [O.HostCol(T.T_I64, d["a"][sl]), None, O.HostCol(T.T_DEC128, O.i128_array(d["dd"][sl]), av, 31, 4), O.HostCol(T.T_F64, d["f"][sl], av),
            O.HostCol(T.T_I32, d["i"][sl]), O.HostCol(T.T_F64, d["f"][sl], av), O.HostCol(T.T_I64, d["a"][sl], av),
            O.HostCol(T.T_DEC64, d["dec"][sl], av, 15, 2), O.HostCol(T.T_DATE, d["dt"][sl]),
            O.HostCol(T.T_DEC128, O.i128_array(d["dd"][sl]), av, 31, 4), O.HostCol(T.T_DEC128, O.i128_array(d["dd"][sl]), None, 31, 4)]
    return keys, args


def oracle_table(oracle, aggs=SB_AGGS, keys=SB_KEYS):
    kt = (C.c_int32 * len(keys[0]))(*keys[0])
    kn = (C.c_uint8 * len(keys[0]))(*keys[1])
    ad = (O.OAgg * len(aggs))()
    for i, a in enumerate(aggs):
        ad[i].kind, ad[i].arg_type, ad[i].arg_precision, ad[i].arg_scale, ad[i].arg_nullable = a
    oracle.orc_hashagg_create.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int]
    return C.c_void_p(oracle.orc_hashagg_create(kt, kn, len(keys[0]), ad, len(aggs)))


def oracle_add(oracle, h, keys, args, n):
    aa = (O.OCol * len(args))()
    for i, a in enumerate(args):
        if a is not None:
            aa[i] = a.c()
    assert oracle.orc_hashagg_add_block(h, O.cols(keys), aa, C.c_int64(n)) == 0


def oracle_fields(oracle, h):
    t, a = (C.c_int32 * 96)(), (C.c_int32 * 96)()
    nf = oracle.orc_hashagg_state_fields(h, t, a)
    return [(t[i], a[i]) for i in range(nf)]


def test_state_fields_match_the_oracle_statement_of_serialize_type(gpu, oracle):
    g = gpu.GroupBy(SB_KEYS[0], SB_AGGS, SB_KEYS[1])
    h = oracle_table(oracle)
    assert g.state_fields() == oracle_fields(oracle, h)
    # count [u64]; sum [value] (+flag); min/max [has, value] (+flag)
    assert [t for t, a in g.state_fields() if a == 2] == [T.T_DEC128, T.T_BOOL]
    assert [t for t, a in g.state_fields() if a == 4] == [T.T_BOOL, T.T_I32]
    assert [t for t, a in g.state_fields() if a == 5] == [T.T_BOOL, T.T_F64, T.T_BOOL]
    oracle.orc_hashagg_destroy(h)


@pytest.mark.parametrize("n,card", [(50, 5), (20_000, 300), (150_000, 40_000)])
def test_device_state_block_feeds_the_cpu_final_stage(gpu, oracle, n, card):
    """device partial aggregates -> dbhip_groupby_flush_state_block -> the ORACLE's TransformDeserializer/batch_merge
    (orc_hashagg_merge_state_block) == the oracle over all rows."""
    d = sb_data(n, card, 7 + n)
    final = oracle_table(oracle)
    cuts = [0, n // 3, n]
    for lo, hi in zip(cuts[:-1], cuts[1:]):
        g = gpu.GroupBy(SB_KEYS[0], SB_AGGS, SB_KEYS[1])
        keys, args = sb_gpu_cols(gpu, d, lo, hi)
        g.add_block(keys, args, hi - lo)
        kcols, fcols = g.flush_state_block()
        m = kcols[0].n
        hk = [O.HostCol(T.T_I64, kcols[0].to_numpy(), kcols[0].validity_numpy()), O.HostCol(T.T_STRING, kcols[1].to_numpy())]
        hf = []
        for (t, _a), c in zip(g.state_fields(), fcols):
            if t == T.T_BOOL:
                hf.append(O.HostCol(T.T_BOOL, pack_bits(c.to_numpy())))
            elif t == T.T_DEC128:
                hf.append(O.HostCol(T.T_DEC128, O.i128_array(c.to_numpy()), None, c.precision, c.scale))
            else:
                hf.append(O.HostCol(t, c.to_numpy(), None, c.precision, c.scale))
        assert oracle.orc_hashagg_merge_state_block(final, O.cols(hk), O.cols(hf), C.c_int64(m)) == 0
    whole = oracle_table(oracle)
    keys, args = sb_host_cols(d, 0, n)
    oracle_add(oracle, whole, keys, args, n)
    got, exp = oracle_rows(oracle, final, SB_KEYS[0], SB_AGGS), oracle_rows(oracle, whole, SB_KEYS[0], SB_AGGS)
    oracle.orc_hashagg_destroy(final)
    oracle.orc_hashagg_destroy(whole)
    assert norm(got) == norm(exp)


@pytest.mark.parametrize("n,card", [(50, 5), (20_000, 300), (150_000, 40_000)])
def test_cpu_state_block_feeds_the_device_final_stage(gpu, oracle, n, card):
    """the ORACLE's partial aggregates -> Payload::aggregate_flush restated (orc_hashagg_flush_state_block) ->
    dbhip_groupby_merge_state_block == the device over all rows == the oracle over all rows."""
    d = sb_data(n, card, 11 + n)
    final = gpu.GroupBy(SB_KEYS[0], SB_AGGS, SB_KEYS[1])
    cuts = [0, n // 4, n // 2, n]
    for lo, hi in zip(cuts[:-1], cuts[1:]):
        h = oracle_table(oracle)
        keys, args = sb_host_cols(d, lo, hi)
        oracle_add(oracle, h, keys, args, hi - lo)
        m = oracle.orc_hashagg_num_groups(h)
        fields = oracle_fields(oracle, h)
        kb = [np.zeros(m * 8 + 16, np.uint8), np.zeros(m * 16 + 16, np.uint8)]
        kv = [np.zeros(m + 8, np.uint8), np.zeros(m + 8, np.uint8)]
        fb = [np.zeros(m * 16 + 16, np.uint8) for _ in fields]
        kp = (C.c_void_p * 2)(*[b.ctypes.data for b in kb])
        kvp = (C.c_void_p * 2)(*[b.ctypes.data for b in kv])
        fp = (C.c_void_p * len(fb))(*[b.ctypes.data for b in fb])
        assert oracle.orc_hashagg_flush_state_block(h, kp, kvp, fp, None) == 0
        oracle.orc_hashagg_destroy(h)
        gk = [gpu.Column.from_numpy(kb[0][:8 * m].view(np.int64), validity=kv[0][:m].astype(bool)), gpu.Column.from_views(kb[1][:16 * m].reshape(-1, 16))]
        gf = []
        for (t, a), b in zip(fields, fb):
            if t == T.T_BOOL:
                gf.append(gpu.Column.boolean(b[:m].astype(bool)))
            elif t == T.T_DEC128:
                gf.append(gpu.Column(T.T_DEC128, m, gpu.DeviceBuffer.from_numpy(b[:16 * m]), precision=38, scale=SB_AGGS[a][3]))
            else:
                gf.append(gpu.Column.from_numpy(b[:m * ELEM_SIZE[t]].view(NP_OF[t]), t))
        final.merge_state_block(gk, gf, m)
    whole = gpu.GroupBy(SB_KEYS[0], SB_AGGS, SB_KEYS[1])
    keys, args = sb_gpu_cols(gpu, d, 0, n)
    whole.add_block(keys, args, n)
    ow = oracle_table(oracle)
    hk, ha = sb_host_cols(d, 0, n)
    oracle_add(oracle, ow, hk, ha, n)
    exp = oracle_rows(oracle, ow, SB_KEYS[0], SB_AGGS)
    oracle.orc_hashagg_destroy(ow)
    assert norm(final.result()) == norm(whole.result()) == norm(exp)


def test_merge_state_block_rejects_bad_blocks_without_touching_the_table(gpu):
    """a rejected block must leave the table exactly as it was (layout and states)."""
    aggs = [(T.AGG_COUNT, 0, 0, 0, 0), (T.AGG_MIN, T.T_I64, 0, 0, 0)]
    g = gpu.GroupBy([T.T_I64], aggs)
    k = np.arange(100, dtype=np.int64) % 7
    a = np.arange(100, dtype=np.int64)
    g.add_block([gpu.Column.from_numpy(k)], [None, gpu.Column.from_numpy(a)], 100)
    before = sorted(g.result())
    cnt = gpu.Column.from_numpy(np.ones(7, np.uint64))
    has = gpu.Column.boolean(np.ones(7, bool))
    bad_val = gpu.Column.from_numpy(np.zeros(7, np.int32))      # wrong type for the min(i64) value field
    with pytest.raises(T.DbhipError) as e:
        g.merge_state_block([gpu.Column.from_numpy(np.arange(7, dtype=np.int64))], [cnt, has, bad_val], 7)
    assert e.value.code == T.ERR_INVALID
    assert sorted(g.result()) == before
    g.add_block([gpu.Column.from_numpy(k)], [None, gpu.Column.from_numpy(a)], 100)   # COUNT still counts rows
    assert sorted(r[1] for r in g.result()) == sorted(2 * int((k == key).sum()) for key in range(7))
    # and a well-formed min/max block merges (round 1 returned UNSUPPORTED here)
    good_val = gpu.Column.from_numpy(np.full(7, -5, np.int64))
    g.merge_state_block([gpu.Column.from_numpy(np.arange(7, dtype=np.int64))], [cnt, has, good_val], 7)
    assert all(r[2] == -5 for r in g.result())


# ---------------------------------------------------------------------------------------------------------------
# a12: device hash partitioning for the exchange
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("card,nb,max_rows", [(4, 8, 16), (1000, 8, 256), (1000, 3, 512), (100_000, 8, 16384), (5000, 8, 64)])
def test_partition_blocks_route_rows_by_hash_mod_buckets(gpu, card, nb, max_rows):
    """payload.rs:548-589: bucket = group hash % bucket count. Blocks mode: headers, the any-overflow flag, rows; a
    table rebuilt from its own blocks equals the original."""
    n = max(card * 4, 1000)
    rng = np.random.default_rng(card + nb)
    k = rng.integers(0, card, n).astype(np.int64)
    a = rng.integers(-10**6, 10**6, n).astype(np.int64)
    g = gpu.GroupBy([T.T_I64], [(T.AGG_SUM, T.T_I64, 0, 0, 0), (T.AGG_COUNT, 0, 0, 0, 0)])
    g.add_block([gpu.Column.from_numpy(k)], [gpu.Column.from_numpy(a), None], n)
    W = g.row_bytes() // 8
    allrows = g.flush_serialized()
    exp_parts = [allrows[allrows[:, 1] % np.uint64(nb) == b] for b in range(nb)]
    assert all(int(r[1]) == mix(int(r[0])) for r in allrows[:50])          # word 1 is the group hash of the i64 key
    blocks = gpu.DeviceBuffer(nb * (max_rows + 1) * W * 8)
    g.partition_blocks(blocks.ptr, nb, max_rows)
    got = blocks.to_numpy(np.uint64).reshape(nb, max_rows + 1, W)
    over = any(len(p) > max_rows for p in exp_parts)
    for b in range(nb):
        assert int(got[b, 0, 1]) == (1 if over else 0)
        if len(exp_parts[b]) > max_rows:
            assert int(got[b, 0, 0]) == (1 << 64) - 1
        else:
            assert int(got[b, 0, 0]) == len(exp_parts[b])
            rows = got[b, 1:1 + len(exp_parts[b])]
            assert sorted(map(tuple, rows.tolist())) == sorted(map(tuple, exp_parts[b].tolist()))
    before = sorted(g.result())
    if over:
        with pytest.raises(T.DbhipError) as e:
            g.replace_with_blocks(blocks.ptr, nb, max_rows)
        assert e.value.code == T.ERR_CAPACITY
        assert sorted(g.result()) == before            # untouched
    else:
        g.replace_with_blocks(blocks.ptr, nb, max_rows)
        assert sorted(g.result()) == before
    # variable-length form
    out = gpu.DeviceBuffer(max(len(allrows), 1) * W * 8)
    counts = g.flush_partitioned(nb, out.ptr, len(allrows))
    assert counts == [len(p) for p in exp_parts]
    rows = out.to_numpy(np.uint64, len(allrows) * W).reshape(-1, W)
    off = 0
    for b in range(nb):
        part = rows[off:off + counts[b]]
        assert np.all(part[:, 1] % np.uint64(nb) == b)
        assert sorted(map(tuple, part.tolist())) == sorted(map(tuple, exp_parts[b].tolist()))
        off += counts[b]


def test_merge_blocks_reports_the_callers_own_overflowed_block(gpu):
    """the owner of an overflowed block must get DBHIP_ERR_CAPACITY too (it skips its own block's ROWS, not its
    header), otherwise the other ranks enter the fallback collective alone."""
    g = gpu.GroupBy([T.T_I64], [(T.AGG_COUNT, 0, 0, 0, 0)])
    k = np.arange(500, dtype=np.int64)
    g.add_block([gpu.Column.from_numpy(k)], [None], 500)
    W = g.row_bytes() // 8
    max_rows = 64
    blocks = gpu.DeviceBuffer(2 * (max_rows + 1) * W * 8).zero()
    g.flush_block(blocks.ptr, max_rows)                         # own block (rank 0): overflowed
    hdr = blocks.to_numpy(np.uint64, W)
    assert int(hdr[0]) == (1 << 64) - 1
    with pytest.raises(T.DbhipError) as e:
        g.merge_blocks(blocks.ptr, 2, max_rows, skip_block=0)   # block 1 is empty (count 0)
    assert e.value.code == T.ERR_CAPACITY
    assert g.num_groups() == 500


# ---------------------------------------------------------------------------------------------------------------
# a9: string keys of any length (arena-backed rows)
# ---------------------------------------------------------------------------------------------------------------
def oracle_string_groups(oracle, h, nagg_cols):
    """[(key string bytes | None, second key, agg values...)] of an oracle table whose key 0 is a String of any length"""
    g = oracle.orc_hashagg_num_groups(h)
    oracle.orc_hashagg_key_strings.restype = C.c_int64
    total = oracle.orc_hashagg_key_strings(h, 0, None, None)
    offs = np.zeros(g + 1, np.int64)
    data = np.zeros(max(total, 1), np.uint8)
    oracle.orc_hashagg_key_strings(h, 0, offs.ctypes.data_as(C.c_void_p), data.ctypes.data_as(C.c_void_p))
    return [bytes(data[offs[i]:offs[i + 1]]) for i in range(g)]


@pytest.mark.parametrize("n,card,maxlen", [(1, 1, 40), (1000, 50, 13), (200_000, 3000, 64), (300_000, 250_000, 30), (100_000, 7, 200)])
def test_groupby_string_keys_of_any_length(gpu, oracle, n, card, maxlen):
    """GROUP BY c_name-style keys: strings of 0..maxlen bytes (inline and long, sharing prefixes and lengths), a nullable
    string key plus an Int32 key, sum / count / min — the device table (arena-backed rows, payload.rs:361-486) against the
    oracle (which keeps `(len, ptr into its arena)` like the reference), compared as sorted sets; then a second block, a
    serialized round trip with the arena, and the result columns' long views."""
    rng = np.random.default_rng(n + card + maxlen)
    # pool of distinct strings: a common 12-byte prefix + varying tails, so prefix + length alone never decide equality
    pool = []
    for i in range(card):
        ln = int(rng.integers(0, maxlen + 1))
        body = (b"Customer#000" + (b"%09d" % i) + b"x" * maxlen)[:ln] if ln > 4 else (b"%d" % i)[:ln]
        pool.append(body)
    pool = sorted(set(pool))
    idx = rng.integers(0, len(pool), n)
    strs = [pool[i] for i in idx]
    sv = rng.integers(0, 12, n) > 0
    k2 = rng.integers(0, 3, n).astype(np.int32)
    a = rng.integers(-10**9, 10**9, n).astype(np.int64)
    key_types, key_nullable = [T.T_STRING, T.T_I32], [1, 0]
    aggs = [(T.AGG_SUM, T.T_I64, 0, 0, 0), (T.AGG_COUNT, 0, 0, 0, 0), (T.AGG_MIN, T.T_I64, 0, 0, 0)]

    def gcols(lo, hi):
        return [gpu.Column.strings(strs[lo:hi], validity=sv[lo:hi]), gpu.Column.from_numpy(k2[lo:hi])], \
               [gpu.Column.from_numpy(a[lo:hi]), None, gpu.Column.from_numpy(a[lo:hi])]

    def expect(lo, hi):
        out = {}
        for i in range(lo, hi):
            key = (strs[i] if sv[i] else None, int(k2[i]))
            s = out.setdefault(key, [0, 0, None])
            s[0] += int(a[i]); s[1] += 1; s[2] = int(a[i]) if s[2] is None else min(s[2], int(a[i]))
        return sorted(((k[0] is None, k[0] or b"", k[1]) + tuple(v) for k, v in out.items()))

    def rows_of(g):
        return sorted(((r[0] is None, r[0] or b"", r[1]) + tuple(r[2:]) for r in g.result()))

    cut = n // 2
    g = gpu.GroupBy(key_types, aggs, key_nullable)
    keys, args = gcols(0, cut) if cut else gcols(0, n)
    g.add_block(keys, args, cut if cut else n)
    if cut:
        keys, args = gcols(cut, n)
        g.add_block(keys, args, n - cut)
    assert rows_of(g) == expect(0, n)
    # the oracle agrees (its own arena-backed payload rows) — group count and the multiset of key strings
    v, buf = make_views_general(strs)
    h = oracle_groupby(oracle, key_types, key_nullable, aggs, [O.HostCol(T.T_STRING, v, sv, buffers=[buf]), O.HostCol(T.T_I32, k2)],
                       [O.HostCol(T.T_I64, a), None, O.HostCol(T.T_I64, a)], n)
    okeys = oracle_string_groups(oracle, h, 3)
    oracle.orc_hashagg_destroy(h)
    assert g.num_groups() == len(okeys)
    assert sorted(okeys) == sorted((r[1] for r in rows_of(g)))
    # serialized round trip: rows + arena -> another table (combine_payload with re-based strings)
    rows = g.flush_serialized()
    g2 = gpu.GroupBy(key_types, aggs, key_nullable)
    keys, args = gcols(0, min(n, 100))
    g2.add_block(keys, args, min(n, 100))
    g2.merge_serialized_arena(rows, g.arena_numpy())
    exp2 = {}
    for src in (expect(0, n), expect(0, min(n, 100))):
        for r in src:
            s = exp2.setdefault(r[:3], [0, 0, None])
            s[0] += r[3]; s[1] += r[4]; s[2] = r[5] if s[2] is None else min(s[2], r[5])
    assert rows_of(g2) == sorted(k + tuple(v) for k, v in exp2.items())


def test_fixed_block_exchange_refuses_tables_with_long_string_keys(gpu):
    """flush_block / partition_blocks / flush_partitioned / merge_blocks / replace_with_blocks move rows without
    the arena; a key longer than 12 bytes in such a row is an OFFSET that the receiver would read as an address. They must
    return DBHIP_ERR_UNSUPPORTED (the table exchanges through flush_serialized + arena -> merge_serialized_arena instead);
    a table whose strings all fit inline still goes through."""
    from databend_amd._lib import DbhipError, ERR_UNSUPPORTED
    D = gpu
    aggs = [(T.AGG_SUM, T.T_I64, 0, 0, 0)]
    n = 1000
    a = np.arange(n, dtype=np.int64)

    def table(strs):
        g = D.GroupBy([T.T_STRING], aggs, [0])
        g.add_block([D.Column.strings(strs)], [D.Column.from_numpy(a)], n)
        return g

    long_t = table([b"Customer#%09d-long" % (i % 37) for i in range(n)])
    W = long_t.row_bytes() // 8
    buf = D.DeviceBuffer(4 * 257 * W * 8)
    for call in (lambda: long_t.flush_block(buf.ptr, 256), lambda: long_t.partition_blocks(buf.ptr, 4, 256),
                 lambda: long_t.flush_partitioned(4, buf.ptr, 1024), lambda: long_t.merge_blocks(buf.ptr, 1, 256),
                 lambda: long_t.replace_with_blocks(buf.ptr, 1, 256)):
        with pytest.raises(DbhipError) as e:
            call()
        assert e.value.code == ERR_UNSUPPORTED and "arena" in str(e.value)
    assert long_t.num_groups() == 37   # nothing was touched
    # the supported route for such a table: rows + arena
    rows, arena = long_t.flush_serialized(), long_t.arena_numpy()
    other = D.GroupBy([T.T_STRING], aggs, [0])
    other.merge_serialized_arena(rows, arena)
    assert sorted(other.result()) == sorted(long_t.result())
    short_t = table([b"k%d" % (i % 37) for i in range(n)])
    short_t.partition_blocks(buf.ptr, 4, 256)
    recv = D.GroupBy([T.T_STRING], aggs, [0])
    recv.replace_with_blocks(buf.ptr, 4, 256)
    assert sorted(recv.result()) == sorted(short_t.result())


@pytest.mark.parametrize("card,nb,max_rows", [(6, 8, 256), (1500, 8, 256), (3000, 8, 256), (40, 3, 64)])
def test_block_exchange_queued_and_read_back_forms_agree(gpu, card, nb, max_rows):
    """a table with room for every row the blocks COULD hold (nb x max_rows) merges / is rebuilt with the headers judged
    on the device and ONE read-back at the end (gbk_api.h blocks_queued); a table without that room reads the headers first (rounds 2-5).
    Both forms: the same groups, the same refusal of an overflowed sender with the table left as it was, the same count on the host."""
    rng = np.random.default_rng(card * nb)
    aggs = [(T.AGG_SUM, T.T_I64, 0, 0, 0), (T.AGG_COUNT, 0, 0, 0, 0), (T.AGG_MIN, T.T_I64, 0, 0, 0)]
    n = 20_000
    k = rng.integers(0, card, n).astype(np.int64)
    a = rng.integers(-10**9, 10**9, n).astype(np.int64)
    src = gpu.GroupBy([T.T_I64], aggs)
    src.add_block([gpu.Column.from_numpy(k)], [gpu.Column.from_numpy(a), None, gpu.Column.from_numpy(a)], n)
    W = src.row_bytes() // 8
    blocks = gpu.DeviceBuffer(nb * (max_rows + 1) * W * 8)
    src.partition_blocks(blocks.ptr, nb, max_rows)
    src.partition_blocks(blocks.ptr, nb, max_rows)     # twice: the cursors are left zeroed by the first call's header kernel
    heads = blocks.to_numpy(np.uint64).reshape(nb, max_rows + 1, W)[:, 0, 0]
    over = bool((heads == np.uint64((1 << 64) - 1)).any())
    assert over == (card == 3000)
    expect = sorted(src.result())
    # some other content in the receiving tables: replace must drop it, merge must keep it
    k2 = rng.integers(card, card + 17, 500).astype(np.int64)
    a2 = rng.integers(-100, 100, 500).astype(np.int64)

    def table(capacity):
        g = gpu.GroupBy([T.T_I64], aggs, capacity=capacity)
        g.add_block([gpu.Column.from_numpy(k2)], [gpu.Column.from_numpy(a2), None, gpu.Column.from_numpy(a2)], 500)
        return g
    own = sorted(table(1024).result())
    for capacity in (1024, 1 << 16):                   # 1024 slots < nb x max_rows x 1.35: headers read first; 65,536: queued
        for replace in (True, False):
            g = table(capacity)
            if over:
                with pytest.raises(T.DbhipError) as e:
                    g.replace_with_blocks(blocks.ptr, nb, max_rows) if replace else g.merge_blocks(blocks.ptr, nb, max_rows)
                assert e.value.code == T.ERR_CAPACITY and "overflowed" in str(e.value)
                assert sorted(g.result()) == own and g.num_groups() == len(own)
            elif replace:
                g.replace_with_blocks(blocks.ptr, nb, max_rows)
                assert sorted(g.result()) == expect and g.num_groups() == len(expect)
            else:
                g.merge_blocks(blocks.ptr, nb, max_rows, skip_block=1)
                got = blocks.to_numpy(np.uint64).reshape(nb, max_rows + 1, W)
                skipped = {int(r[0]) for r in got[1, 1:1 + int(got[1, 0, 0])].view(np.int64)}
                assert sorted(g.result()) == sorted(own + [r for r in expect if r[0] not in skipped])
            # the table keeps working after either form
            g.add_block([gpu.Column.from_numpy(k2)], [gpu.Column.from_numpy(a2), None, gpu.Column.from_numpy(a2)], 500)
            g.destroy()
    src.destroy()


# ---------------------------------------------------------------------------------------------------------------
# min / max over Decimal128 (r03): a three-word state merged under a per-state lock, row path only
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n,card", [(7, 2), (5000, 3), (400_000, 4), (300_000, 2500), (200_000, 150_000)])
def test_min_max_over_decimal128(gpu, oracle, n, card):
    """values spread over the whole i128 range of a Decimal(38, s) — equal high words with different low words, negative values, exact
    duplicates — in two blocks, few groups (every wave fights for the same states) and many; nullable and not; against the oracle,
    and merged across two tables through serialized rows (what the exchange does)."""
    rng = np.random.default_rng(n * 31 + card)
    k = rng.integers(0, card, n).astype(np.int64)
    hi = rng.integers(-3, 4, n)                               # few distinct high parts: the low word decides often
    lo = rng.integers(0, 2**63, n, dtype=np.int64)
    vals = [int(h) * 2**64 + int(l) * (1 if i % 3 else 2) % 2**64 for i, (h, l) in enumerate(zip(hi.tolist(), lo.tolist()))]
    vals = [v if abs(v) < 10**38 else v % 10**37 for v in vals]
    if n > 10:
        vals[5] = vals[3]
        vals[7] = -(10**38 - 1)
        vals[9] = 10**38 - 1
    av = (rng.integers(0, 4, n) > 0) & (k % 5 != 0)
    key_types, key_nullable = [T.T_I64], [0]
    aggs = [(T.AGG_MIN, T.T_DEC128, 38, 2, 1), (T.AGG_MAX, T.T_DEC128, 38, 2, 1), (T.AGG_MIN, T.T_DEC128, 38, 2, 0), (T.AGG_MAX, T.T_DEC128, 38, 2, 0),
            (T.AGG_COUNT, 0, 0, 0, 0), (T.AGG_MAX, T.T_I64, 0, 0, 0)]

    def add(g, lo_, hi_):
        sl = slice(lo_, hi_)
        dn = gpu.Column.decimal128(vals[sl], 38, 2, validity=av[sl])
        dd = gpu.Column.decimal128(vals[sl], 38, 2)
        g.add_block([gpu.Column.from_numpy(k[sl])], [dn, dn, dd, dd, None, gpu.Column.from_numpy(k[sl])], hi_ - lo_)

    g = gpu.GroupBy(key_types, aggs, key_nullable)
    half = n // 2
    add(g, 0, half)
    add(g, half, n)
    got = g.result()
    hv = O.i128_array(vals)
    h = oracle_groupby(oracle, key_types, key_nullable, aggs, [O.HostCol(T.T_I64, k)],
                       [O.HostCol(T.T_DEC128, hv, av, 38, 2), O.HostCol(T.T_DEC128, hv, av, 38, 2), O.HostCol(T.T_DEC128, hv, None, 38, 2),
                        O.HostCol(T.T_DEC128, hv, None, 38, 2), None, O.HostCol(T.T_I64, k)], n)
    exp = oracle_rows(oracle, h, key_types, aggs)
    oracle.orc_hashagg_destroy(h)
    assert norm(got) == norm(exp)
    # python statement, independent of both
    by = {}
    for key, v, ok in zip(k.tolist(), vals, av.tolist()):
        e = by.setdefault(key, [None, None, None, None])
        if ok:
            e[0] = v if e[0] is None else min(e[0], v)
            e[1] = v if e[1] is None else max(e[1], v)
        e[2] = v if e[2] is None else min(e[2], v)
        e[3] = v if e[3] is None else max(e[3], v)
    assert {r[0]: list(r[1:5]) for r in got} == by
    # two tables merged through serialized rows (the exchange): same result
    g1, g2 = gpu.GroupBy(key_types, aggs, key_nullable), gpu.GroupBy(key_types, aggs, key_nullable)
    add(g1, 0, half)
    add(g2, half, n)
    g1.merge_serialized(g2.flush_serialized())
    assert norm(g1.result()) == norm(exp)
