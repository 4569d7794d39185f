"""GPU: min / max over String (aggregate_min_max_any.rs:62-110, StringState) in the hash aggregation — a three-word state merged under
the per-state lock, the winners' bytes kept in the table's arena — against the oracle's StringState and the reference's own goldens
(min(s) / max(s) of min.txt / max.txt / *_group_by.txt run in tests/test_gpu_golden.py)."""
import numpy as np
import pytest

from databend_amd import _lib as T
from tests import oracle_lib as O
from tests.test_gpu_parity import norm, oracle_groupby, oracle_rows

pytestmark = pytest.mark.gpu

AGGS = [(T.AGG_MIN, T.T_STRING, 0, 0, 1), (T.AGG_MAX, T.T_STRING, 0, 0, 1), (T.AGG_SUM, T.T_I64, 0, 0, 0), (T.AGG_COUNT, 0, 0, 0, 0),
        (T.AGG_MAX, T.T_STRING, 0, 0, 0)]


def make_block(rng, n, card, maxlen):
    keys = rng.integers(0, card, n).astype(np.int64)
    alphabet = [bytes([c]) for c in b"ab\x00\xffzQ"]
    vals = [b"".join(alphabet[int(x)] for x in rng.integers(0, len(alphabet), int(ln))) for ln in rng.integers(0, maxlen + 1, n)]
    valid = rng.random(n) > 0.25
    valid[keys == 3] = False                        # a group whose nullable argument is NULL everywhere
    other = [b"row-%07d-with-a-tail-longer-than-twelve" % int(i) for i in rng.integers(0, 10**7, n)]     # NOT NULL argument, all long
    w = rng.integers(-1000, 1000, n).astype(np.int64)
    return keys, vals, valid, other, w


@pytest.mark.parametrize("n,card,maxlen", [(1, 1, 5), (300, 4, 12), (20_000, 4, 40), (50_000, 3000, 30), (120_000, 60_000, 20)])
def test_min_max_over_strings_equal_the_oracle(gpu, oracle, n, card, maxlen):
    """Three blocks into one table — every block's columns are dropped before the next one arrives, so a state that still pointed into a
    block's buffers would read freed memory — against the oracle over the same three blocks: short (inline) and long values, embedded
    0x00 / 0xFF bytes, NULL rows, an all-NULL group, a handful of groups (every row of a wave fights for one state's lock) and tens of
    thousands."""
    from databend_amd.device import make_views_general
    D = gpu
    rng = np.random.default_rng(n + card)
    g = D.GroupBy([T.T_I64], AGGS)
    h = None
    hold = []
    for b in range(3):
        keys, vals, valid, other, w = make_block(rng, n, card, maxlen)
        cs, co = D.Column.strings(vals, validity=valid), D.Column.strings(other)
        g.add_block([D.Column.from_numpy(keys)], [cs, cs, D.Column.from_numpy(w), None, co], n)
        del cs, co
        v, buf = make_views_general(vals)
        v2, buf2 = make_views_general(other)
        hs, ho = O.HostCol(T.T_STRING, v, valid, buffers=[buf]), O.HostCol(T.T_STRING, v2, buffers=[buf2])
        hold.append((hs, ho))
        hk, hw = O.HostCol(T.T_I64, keys), O.HostCol(T.T_I64, w)
        if h is None:
            h = oracle_groupby(oracle, [T.T_I64], [0], AGGS, [hk], [hs, hs, hw, None, ho], n)
        else:
            import ctypes as C
            args = (O.OCol * len(AGGS))()
            for i, a in enumerate([hs, hs, hw, None, ho]):
                if a is not None:
                    args[i] = a.c()
            assert oracle.orc_hashagg_add_block(h, O.cols([hk]), args, C.c_int64(n)) == 0
    exp = oracle_rows(oracle, h, [T.T_I64], AGGS)
    oracle.orc_hashagg_destroy(h)
    got = g.result()
    assert norm(got) == norm(exp)
    if card > 3 and n > 1000:
        assert any(r[0] == 3 and r[1] is None and r[2] is None for r in got)


def test_string_states_merge_between_live_tables_and_do_not_travel(gpu, oracle):
    """TransformFinalAggregate's merge of partial tables (dbhip_groupby_flush_serialized -> dbhip_groupby_merge_serialized, both tables
    alive in this process): the receiving table copies the winners into ITS arena, so the result survives the source table. The exchange
    forms that would carry an address out of the process (fixed-size blocks, partitions) are refused; the serialized-state block is the
    form that travels (a Nullable(String) column with the table's arena as its data buffer)."""
    D = gpu
    rng = np.random.default_rng(5)
    n = 30_000
    aggs = [(T.AGG_MIN, T.T_STRING, 0, 0, 0), (T.AGG_MAX, T.T_STRING, 0, 0, 0), (T.AGG_COUNT, 0, 0, 0, 0)]
    parts, allk, allv = [], [], []
    for p in range(3):
        keys = rng.integers(0, 500, n).astype(np.int64)
        vals = [b"partial-%d-value-%06d" % (p, int(x)) for x in rng.integers(0, 10**6, n)]
        t = D.GroupBy([T.T_I64], aggs)
        cs = D.Column.strings(vals)
        t.add_block([D.Column.from_numpy(keys)], [cs, cs, None], n)
        parts.append(t)
        allk += keys.tolist()
        allv += vals
    final = D.GroupBy([T.T_I64], aggs)
    for t in parts:
        final.merge_serialized(t.flush_serialized())
        t.destroy()                                   # the source table (and its arena) is gone before the result is read
    exp = {}
    for k, s in zip(allk, allv):
        e = exp.setdefault(k, [s, s, 0])
        e[0], e[1], e[2] = min(e[0], s), max(e[1], s), e[2] + 1
    assert {r[0]: list(r[1:]) for r in final.result()} == exp
    import ctypes as C
    block = D.DeviceBuffer(8 * 64 * 1024)
    assert T.lib().dbhip_groupby_flush_block(final.h, C.c_void_p(block.ptr), C.c_int64(1000), None) == T.ERR_UNSUPPORTED
    # (round 5: the serialized-state block carries the strings as a Nullable(String) column — tests/test_gpu_state_block_wide.py)
    nf = C.c_int32()
    assert T.lib().dbhip_groupby_state_fields(final.h, None, None, 0, C.byref(nf)) == 0 and nf.value == 5


def test_the_arena_does_not_keep_displaced_winners_forever(gpu):
    """max() over a column whose strings grow block after block replaces every group's long winner per block: the bytes of the displaced
    winners are reclaimed once the arena holds more than twice its live bytes — 60 blocks pin ~5 MB, the arena stays near
    its live size, and the result is the last block's strings."""
    D = gpu
    groups, blocks = 2000, 60
    g = D.GroupBy([T.T_I64], [(T.AGG_MAX, T.T_STRING, 0, 0, 0)])
    keys = np.arange(groups, dtype=np.int64)
    last = None
    for b in range(blocks):
        strs = [b"block %04d group %06d padded to forty bytes." % (b, k) for k in keys]
        g.add_block([D.Column.from_numpy(keys)], [D.Column.strings(strs)], groups)
        last = strs
    _, used = g.arena()
    got = dict(g.result())
    g.destroy()
    assert got == {int(k): s for k, s in zip(keys, last)}
    live = groups * 48
    assert used < 2 * live + (3 << 20), (used, live)          # (without compaction: blocks x groups x 48 = 5.8 MB)
