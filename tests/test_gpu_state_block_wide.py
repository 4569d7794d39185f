"""GPU: the round-4 / round-5 aggregate states at the boundary — min / max over String, sum / min / max over Decimal256 and
Decimal256 group keys in the serialized-state block, both directions against the oracle:
  * min / max over String serialize as ONE Nullable(String) column (aggregate_min_max_any.rs:163-205: serialize_type / batch_serialize /
    batch_merge) — in the flattened ABI its two buffers, [Boolean validity][String values], long values by offset into the table's arena;
  * min / max over Decimal as ONE Nullable(Decimal) column (aggregate_min_max_any_decimal.rs:140-190) — [Boolean validity][values];
  * sum over Decimal256 as the Decimal256 total (aggregate_sum.rs:281-298 with T = i256).
and min / max over Decimal256 itself (aggregate_min_max_any_decimal.rs:45-138), new this round, against the oracle's restatement."""
import ctypes as C

import numpy as np
import pytest

from databend_amd import _lib as T
from databend_amd.device import ints_to_limbs, make_views_general, pack_bits
from tests import oracle_lib as O
from tests.test_gpu_parity import norm, oracle_groupby, oracle_rows

pytestmark = pytest.mark.gpu

KEY_TYPES, KEY_NULLABLE = [T.T_I64, T.T_DEC256], [0, 1]
AGGS = [(T.AGG_MIN, T.T_STRING, 0, 0, 1), (T.AGG_MAX, T.T_STRING, 0, 0, 0), (T.AGG_SUM, T.T_DEC256, 76, 4, 0), (T.AGG_SUM, T.T_DEC256, 60, 2, 1),
        (T.AGG_MIN, T.T_DEC256, 76, 4, 0), (T.AGG_MAX, T.T_DEC256, 76, 4, 1), (T.AGG_COUNT, 0, 0, 0, 0)]


def big(rng, n, digits):
    return [int(a) * 10**(digits - 18) + int(b) for a, b in zip(rng.integers(-10**17, 10**17, n), rng.integers(0, 10**9, n))]


def data(n, card, seed):
    rng = np.random.default_rng(seed)
    pool = big(rng, max(card // 3, 1), 60) + [0, -1, 10**75, -(10**75)]
    d = dict(k1=rng.integers(0, card, n).astype(np.int64), k2=[pool[int(i)] for i in rng.integers(0, len(pool), n)], k2v=rng.random(n) > 0.1)
    alphabet = [bytes([c]) for c in b"ab\x00\xffzQ"]
    d["s1"] = [b"".join(alphabet[int(x)] for x in rng.integers(0, len(alphabet), int(ln))) for ln in rng.integers(0, 31, n)]
    d["s1v"] = rng.random(n) > 0.3
    d["s1v"][d["k1"] == 2] = False                       # a group whose nullable String argument is NULL everywhere
    d["s2"] = [b"row-%07d-with-a-tail-longer-than-twelve" % int(i) for i in rng.integers(0, 10**7, n)]
    d["d1"] = big(rng, n, 70)
    d["d2"] = big(rng, n, 50)
    d["d2v"] = rng.random(n) > 0.3
    d["d2v"][d["k1"] == 1] = False
    d["d3"] = big(rng, n, 76)                            # min / max candidates near +-10^76
    d["d3"][::5] = [-(10**75) - int(i) for i in range(len(d["d3"][::5]))]
    d["d4v"] = rng.random(n) > 0.4
    return d


def gpu_cols(D, d, lo, hi):
    keys = [D.Column.from_numpy(d["k1"][lo:hi]), D.Column.decimal256(d["k2"][lo:hi], 76, 0, validity=d["k2v"][lo:hi])]
    args = [D.Column.strings(d["s1"][lo:hi], validity=d["s1v"][lo:hi]), D.Column.strings(d["s2"][lo:hi]), D.Column.decimal256(d["d1"][lo:hi], 76, 4),
            D.Column.decimal256(d["d2"][lo:hi], 60, 2, validity=d["d2v"][lo:hi]), D.Column.decimal256(d["d3"][lo:hi], 76, 4),
            D.Column.decimal256(d["d3"][lo:hi], 76, 4, validity=d["d4v"][lo:hi]), None]
    return keys, args


def host_cols(d, lo, hi):
    def h256(v, valid=None, p=76, s=4):
        return O.HostCol(T.T_DEC256, ints_to_limbs(v, 256), valid, p, s)
    v1, b1 = make_views_general(d["s1"][lo:hi])
    v2, b2 = make_views_general(d["s2"][lo:hi])
    keys = [O.HostCol(T.T_I64, d["k1"][lo:hi]), h256(d["k2"][lo:hi], d["k2v"][lo:hi], 76, 0)]
    args = [O.HostCol(T.T_STRING, v1, d["s1v"][lo:hi], buffers=[b1]), O.HostCol(T.T_STRING, v2, buffers=[b2]), h256(d["d1"][lo:hi]),
            h256(d["d2"][lo:hi], d["d2v"][lo:hi], 60, 2), h256(d["d3"][lo:hi]), h256(d["d3"][lo:hi], d["d4v"][lo:hi]), None]
    return keys, args


def oracle_add(oracle, h, keys, args, n):
    a = (O.OCol * len(args))()
    for i, c in enumerate(args):
        if c is not None:
            a[i] = c.c()
    assert oracle.orc_hashagg_add_block(h, O.cols(keys), a, C.c_int64(n)) == 0


def oracle_table(oracle):
    kt = (C.c_int32 * 2)(*KEY_TYPES)
    kn = (C.c_uint8 * 2)(*KEY_NULLABLE)
    ad = (O.OAgg * len(AGGS))()
    for i, (k, t, p, s, nu) in enumerate(AGGS):
        ad[i].kind, ad[i].arg_type, ad[i].arg_precision, ad[i].arg_scale, ad[i].arg_nullable = k, t, p, s, nu
    oracle.orc_hashagg_create.restype = C.c_void_p
    return C.c_void_p(oracle.orc_hashagg_create(kt, kn, 2, ad, len(AGGS)))


@pytest.mark.parametrize("n,card", [(40, 3), (30_000, 6), (90_000, 20_000)])
def test_min_max_over_decimal256_equal_the_oracle(gpu, oracle, n, card):
    d = data(n, card, 3 + n)
    g = gpu.GroupBy(KEY_TYPES, AGGS, KEY_NULLABLE)
    h = None
    for lo, hi in ((0, n // 2), (n // 2, n)):
        keys, args = gpu_cols(gpu, d, lo, hi)
        g.add_block(keys, args, hi - lo)
        hk, ha = host_cols(d, lo, hi)
        if h is None:
            h = oracle_groupby(oracle, KEY_TYPES, KEY_NULLABLE, AGGS, hk, ha, hi - lo)
        else:
            oracle_add(oracle, h, hk, ha, hi - lo)
    exp = oracle_rows(oracle, h, KEY_TYPES, AGGS)
    oracle.orc_hashagg_destroy(h)
    got = g.result()
    assert norm(got) == norm(exp)
    # what was asked for really happened: extreme values on both sides, a NULL maximum for a group without valid rows is possible
    assert any(r[6] is not None and r[6] < -(10**75) + 1 for r in got)


def test_state_fields_of_the_wide_states(gpu):
    g = gpu.GroupBy(KEY_TYPES, AGGS, KEY_NULLABLE)
    f = g.state_fields()
    assert [t for t, a in f if a == 0] == [T.T_BOOL, T.T_STRING, T.T_BOOL]      # Nullable(String) as (validity, values) + the adaptor's flag
    assert [t for t, a in f if a == 1] == [T.T_BOOL, T.T_STRING]
    assert [t for t, a in f if a == 2] == [T.T_DEC256]
    assert [t for t, a in f if a == 3] == [T.T_DEC256, T.T_BOOL]
    assert [t for t, a in f if a == 4] == [T.T_BOOL, T.T_DEC256]
    assert [t for t, a in f if a == 5] == [T.T_BOOL, T.T_DEC256, T.T_BOOL]
    assert [t for t, a in f if a == 6] == [T.T_U64]


@pytest.mark.parametrize("n,card", [(60, 4), (40_000, 300), (120_000, 30_000)])
def test_device_state_block_feeds_the_cpu_final_stage(gpu, oracle, n, card):
    """device partial aggregates -> dbhip_groupby_flush_state_block -> the ORACLE's batch_merge == the oracle over all rows"""
    d = data(n, card, 17 + n)
    final = oracle_table(oracle)
    cuts = [0, n // 3, n]
    for lo, hi in zip(cuts[:-1], cuts[1:]):
        g = gpu.GroupBy(KEY_TYPES, AGGS, KEY_NULLABLE)
        keys, args = gpu_cols(gpu, d, lo, hi)
        g.add_block(keys, args, hi - lo)
        kcols, fcols = g.flush_state_block()
        m = kcols[0].n
        arena = g.arena_numpy()
        arena = arena if len(arena) else np.zeros(16, np.uint8)
        hk = [O.HostCol(T.T_I64, kcols[0].to_numpy()),
              O.HostCol(T.T_DEC256, kcols[1].data.to_numpy(np.uint8, 32 * m), kcols[1].validity_numpy(), 76, 0)]
        hf = []
        for (t, _a), c in zip(g.state_fields(), fcols):
            if t == T.T_BOOL:
                hf.append(O.HostCol(T.T_BOOL, pack_bits(c.to_numpy())))
            elif t == T.T_STRING:
                hf.append(O.HostCol(T.T_STRING, c.data.to_numpy(np.uint8, 16 * m).reshape(-1, 16), buffers=[arena]))
            elif t == T.T_DEC256:
                hf.append(O.HostCol(T.T_DEC256, c.data.to_numpy(np.uint8, 32 * m), None, c.precision, c.scale))
            else:
                hf.append(O.HostCol(t, c.to_numpy(), None, c.precision, c.scale))
        assert oracle.orc_hashagg_merge_state_block(final, O.cols(hk), O.cols(hf), C.c_int64(m)) == 0
    whole = oracle_table(oracle)
    hk, ha = host_cols(d, 0, n)
    oracle_add(oracle, whole, hk, ha, n)
    got, exp = oracle_rows(oracle, final, KEY_TYPES, AGGS), oracle_rows(oracle, whole, KEY_TYPES, AGGS)
    oracle.orc_hashagg_destroy(final)
    oracle.orc_hashagg_destroy(whole)
    assert norm(got) == norm(exp)


@pytest.mark.parametrize("n,card", [(60, 4), (40_000, 300), (120_000, 30_000)])
def test_cpu_state_block_feeds_the_device_final_stage(gpu, oracle, n, card):
    """the ORACLE's partial aggregates -> its restatement of Payload::aggregate_flush -> dbhip_groupby_merge_state_block == the device over all
    rows == the oracle over all rows; then the device's own block back into a second device table (partial -> block -> final on the GPU)."""
    d = data(n, card, 29 + n)
    final = gpu.GroupBy(KEY_TYPES, AGGS, KEY_NULLABLE)
    relay = gpu.GroupBy(KEY_TYPES, AGGS, KEY_NULLABLE)
    cuts = [0, n // 4, n // 2, n]
    fields = gpu.GroupBy(KEY_TYPES, AGGS, KEY_NULLABLE).state_fields()
    oracle.orc_hashagg_bytes.restype = C.c_void_p
    for lo, hi in zip(cuts[:-1], cuts[1:]):
        h = oracle_table(oracle)
        hk, ha = host_cols(d, lo, hi)
        oracle_add(oracle, h, hk, ha, hi - lo)
        m = oracle.orc_hashagg_num_groups(h)
        kb = [np.zeros(m * 8 + 16, np.uint8), np.zeros(m * 32 + 32, np.uint8)]
        kv = [np.zeros(m + 8, np.uint8), np.zeros(m + 8, np.uint8)]
        fb = [np.zeros(m * 32 + 32, np.uint8) for _ in fields]
        kp = (C.c_void_p * 2)(*[b.ctypes.data for b in kb])
        kvp = (C.c_void_p * 2)(*[b.ctypes.data for b in kv])
        fp = (C.c_void_p * len(fb))(*[b.ctypes.data for b in fb])
        assert oracle.orc_hashagg_flush_state_block(h, kp, kvp, fp, None) == 0
        blen = C.c_int64()
        base = oracle.orc_hashagg_bytes(h, C.byref(blen))
        store = np.frombuffer(C.string_at(base, blen.value) if blen.value else b"\0" * 16, np.uint8).copy()
        oracle.orc_hashagg_destroy(h)
        store_dev = gpu.DeviceBuffer.from_numpy(np.concatenate([store, np.zeros(16, np.uint8)]))
        bufs = gpu.DeviceBuffer.from_numpy(np.array([store_dev.ptr], dtype=np.uint64))
        gk = [gpu.Column.from_numpy(kb[0][:8 * m].view(np.int64)),
              gpu.Column(T.T_DEC256, m, gpu.DeviceBuffer.from_numpy(kb[1][:32 * m]), gpu.DeviceBuffer.from_numpy(pack_bits(kv[1][:m].astype(bool))), 76, 0)]
        gf = []
        for (t, a), b in zip(fields, fb):
            if t == T.T_BOOL:
                gf.append(gpu.Column.boolean(b[:m].astype(bool)))
            elif t == T.T_STRING:
                # the oracle's 16-byte form (u32 length, inline bytes | u64 offset at +8) -> Arrow views {len, prefix, buffer 0, offset}
                v = b[:16 * m].reshape(-1, 16).copy()
                for i in range(m):
                    ln = int(v[i, :4].view(np.uint32)[0])
                    if ln > 12:
                        off = int(v[i, 8:16].view(np.uint64)[0])
                        v[i, 4:8] = store[off:off + 4]
                        v[i, 8:12] = 0
                        v[i, 12:16] = np.frombuffer(np.uint32(off).tobytes(), np.uint8)
                gf.append(gpu.Column(T.T_STRING, m, gpu.DeviceBuffer.from_numpy(v), None, buffers=bufs, keep=(store_dev,)))
            elif t == T.T_DEC256:
                gf.append(gpu.Column(T.T_DEC256, m, gpu.DeviceBuffer.from_numpy(b[:32 * m]), None, 76, AGGS[a][3]))
            else:
                gf.append(gpu.Column.from_numpy(b[:m * 8].view(np.uint64), t))
        final.merge_state_block(gk, gf, m)
        del gf, gk, store_dev, bufs          # (the table keeps its own copy of every winning string)
        # device partial -> its own state block -> a second device table
        part = gpu.GroupBy(KEY_TYPES, AGGS, KEY_NULLABLE)
        keys, args = gpu_cols(gpu, d, lo, hi)
        part.add_block(keys, args, hi - lo)
        pk, pf = part.flush_state_block()
        relay.merge_state_block(pk, pf, pk[0].n)
        del pk, pf
        part.destroy()
    whole = gpu.GroupBy(KEY_TYPES, AGGS, KEY_NULLABLE)
    keys, args = gpu_cols(gpu, d, 0, n)
    whole.add_block(keys, args, n)
    ow = oracle_table(oracle)
    hk, ha = host_cols(d, 0, n)
    oracle_add(oracle, ow, hk, ha, n)
    exp = oracle_rows(oracle, ow, KEY_TYPES, AGGS)
    oracle.orc_hashagg_destroy(ow)
    assert norm(final.result()) == norm(whole.result()) == norm(exp)
    assert norm(relay.result()) == norm(exp)
