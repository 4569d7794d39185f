"""GPU: the PIPELINED fused aggregation (dbhip_groupby_set_pipelined / dbhip_groupby_checkpoint, round 6) — the call shape for a
host that hands over the reference's own <= 65,536-row DataBlocks (settings_default.rs:142-148, one
TransformPartialAggregate::transform per block, transform_aggregate_partial.rs:262-270). Results against the CPU oracle; the
deferred errors against what the synchronous call returns for the same blocks."""
import ctypes as C
import threading

import numpy as np
import pytest

from databend_amd import _lib as T
from databend_amd import tpch
from databend_amd._lib import check, lib
from tests import oracle_lib as O

pytestmark = pytest.mark.gpu

BLOCK = 65536


def _slice(D, col, lo, hi):
    es = D.ELEM_SIZE[col.dtype]
    return D.Column(col.dtype, hi - lo, D.BorrowedBuffer(col.data.ptr + lo * es, (hi - lo) * es, keep=col), precision=col.precision, scale=col.scale)


def _li_slice(D, li, lo, hi):
    class _S:
        pass
    s = _S()
    s.n = hi - lo
    for name in ("qty", "price", "disc", "tax", "rf", "ls", "ship"):
        setattr(s, name, _slice(D, getattr(li, name), lo, hi))
    return s


def _merge_tables(D, tables):
    """TransformFinalAggregate: the threads' partial tables into the first one"""
    fin = tables[0]
    for g in tables[1:]:
        fin.merge_serialized(g.flush_serialized())
    return fin


def test_q1_as_65536_row_blocks_from_four_threads_equals_the_oracle(gpu):
    """Q1 fed as the reference's max_block_size blocks, four pipeline threads, each with its own stream and
    its own pipelined partial table; the merged result equals the CPU restatement of the reference's pipeline."""
    D = gpu
    n = 40 * BLOCK + 12345
    host = tpch.gen_lineitem(n, seed=9)
    li = tpch.LineitemDevice(host)
    exp = O.q1_run(host, tpch.Q1_CUTOFF, threads=4)
    nthreads = 4
    tables, errors = [D.GroupBy.q1() for _ in range(nthreads)], []
    blocks = [(lo, min(lo + BLOCK, n)) for lo in range(0, n, BLOCK)]

    def work(t):
        try:
            stream = C.c_void_p()
            check(lib().dbhip_stream_create(C.byref(stream)))
            g = tables[t]
            g.set_pipelined(True, stream=stream)
            tpch.q1_fused_program(li, g, prepare=True)    # PREPARE: the specialised kernels of the shape, incl. the multi-block one
            keep = []
            for lo, hi in blocks[t::nthreads]:
                s = _li_slice(D, li, lo, hi)
                plan = tpch.q1_program(s)
                keep.append((s, plan))      # the blocks' buffers live until the checkpoint
                tpch.q1_fused_program(s, g, plan=plan, stream=stream)
            assert g.checkpoint(stream=stream) == len(keep)
            check(lib().dbhip_stream_destroy(stream))
        except BaseException as e:  # noqa: BLE001
            errors.append((t, repr(e)))

    threads = [threading.Thread(target=work, args=(t,)) for t in range(nthreads)]
    for th in threads:
        th.start()
    for th in threads:
        th.join(300)
    assert not errors, errors
    got = tpch.q1_rows(_merge_tables(D, tables))
    assert got == exp


def _small_program(D, k, x, flt=None):
    """sum(x), count(*) grouped by k, filter x >= flt"""
    ck, cx = D.Column.from_numpy(k), D.Column.from_numpy(x)
    p = D.ExprProgram([cx])
    v = p.load(0)
    f = p.cmp(T.EX_GTE, v, p.const(flt, T.T_I64), keep=(v,)) if flt is not None else -1
    return ck, cx, p, [v, None], f


AGGS = [(T.AGG_SUM, T.T_I64, 0, 0, 0), (T.AGG_COUNT, 0, 0, 0, 0)]


def _expected(k, x, flt=None):
    keep = np.ones(len(k), bool) if flt is None else x >= flt
    return sorted((int(key), int(x[keep & (k == key)].sum()), int((keep & (k == key)).sum())) for key in np.unique(k[keep]))


@pytest.mark.parametrize("groups", [3, 6, 8])
def test_pipelined_blocks_equal_numpy_incl_the_eight_slot_replay(gpu, groups):
    """<= 4 groups stay on the 4-slot kernel; 6 and 8 groups make its workgroups give up — the checkpoint replays the blocks that
    did not commit through the 8-slot kernel (what the synchronous call's second pass does) and still reports every block merged."""
    D = gpu
    rng = np.random.default_rng(groups)
    n, nb = 30_000, 37
    g = D.GroupBy([T.T_I64], AGGS)
    g.set_pipelined(True)
    if groups != 6:   # PREPAREd: the blocks go 32 to a launch (FA_MULTI); groups == 6 leaves them to single launches + the background compile
        ck, cx, p, regs, f = _small_program(D, np.zeros(1, np.int64), np.zeros(1, np.int64), flt=-5 * 10**8)
        g.prepare_program([ck], p, regs, filter_reg=f)
    keep, ks, xs = [], [], []
    for b in range(nb):
        k = rng.integers(0, groups, n).astype(np.int64) * 1_000_003
        x = rng.integers(-10**9, 10**9, n).astype(np.int64)
        ck, cx, p, regs, f = _small_program(D, k, x, flt=-5 * 10**8)
        g.add_block_program([ck], p, regs, n, filter_reg=f)
        keep.append((ck, cx, p))
        ks.append(k), xs.append(x)
    assert g.checkpoint() == nb
    assert sorted(g.result()) == _expected(np.concatenate(ks), np.concatenate(xs), -5 * 10**8)
    # the table stays usable: more pipelined blocks, then a reader that checkpoints by itself
    k = rng.integers(0, groups, n).astype(np.int64) * 1_000_003
    x = rng.integers(-10**9, 10**9, n).astype(np.int64)
    ck, cx, p, regs, f = _small_program(D, k, x, flt=-5 * 10**8)
    g.add_block_program([ck], p, regs, n, filter_reg=f)
    ks.append(k), xs.append(x)
    assert sorted(g.result()) == _expected(np.concatenate(ks), np.concatenate(xs), -5 * 10**8)   # flush_result drains the pipeline first


def test_a_block_with_too_many_groups_is_given_back_with_the_blocks_behind_it(gpu):
    """More than 8 groups inside one workgroup: the synchronous call returns DBHIP_ERR_CAPACITY and merges nothing of the block.
    Pipelined, the checkpoint returns the same code and the number of blocks that WERE merged: windows commit in order, the
    window holding the offending block and everything behind it merge nothing — the table then holds exactly the committed
    blocks, and the caller hands blocks [committed, queued) to the operator-at-a-time path (here: plain add_block)."""
    D = gpu
    rng = np.random.default_rng(5)
    n, nb, bad = 3_000, 4500, 4200
    g = D.GroupBy([T.T_I64], AGGS)
    g.set_pipelined(True)
    ck, cx, p, regs, f = _small_program(D, np.zeros(1, np.int64), np.zeros(1, np.int64))
    g.prepare_program([ck], p, regs)
    keep, ks, xs = [], [], []
    for b in range(nb):
        k = rng.integers(0, 200 if b == bad else 4, n).astype(np.int64)
        x = rng.integers(-10**9, 10**9, n).astype(np.int64)
        ck, cx, p, regs, f = _small_program(D, k, x)
        g.add_block_program([ck], p, regs, n)
        keep.append((ck, cx, p))
        ks.append(k), xs.append(x)
    rc, committed = g.checkpoint(raise_on_error=False)
    assert rc == T.ERR_CAPACITY
    assert committed == 4096             # whole windows (2,048 blocks each) before the offending block's window
    assert b"were not merged" in lib().dbhip_last_error()
    assert sorted(g.result()) == _expected(np.concatenate(ks[:committed]), np.concatenate(xs[:committed]))
    # the caller's fallback for the rest: the operator-at-a-time path on the same (still pipelined) table
    for b in range(committed, nb):
        g.add_block([keep[b][0]], [keep[b][1], None], n)
    assert sorted(g.result()) == _expected(np.concatenate(ks), np.concatenate(xs))
    # synchronous reference: the same block alone returns the same code
    g2 = D.GroupBy([T.T_I64], AGGS)
    ck, cx, p, regs, f = _small_program(D, ks[bad], xs[bad])
    with pytest.raises(T.DbhipError) as e:
        g2.add_block_program([ck], p, regs, n)
    assert e.value.code == T.ERR_CAPACITY


def test_row_errors_of_the_fused_maps_surface_at_the_checkpoint(gpu):
    """A map that raises for a live row (i64 divide by zero) fails the block as a whole in the synchronous call
    (DBHIP_ERR_ROW_ERRORS, nothing merged); pipelined, the checkpoint reports it with the committed block count."""
    D = gpu
    rng = np.random.default_rng(11)
    n, nb, bad = 20_000, 30, 29
    aggs = [(T.AGG_SUM, T.T_F64, 0, 0, 0), (T.AGG_COUNT, 0, 0, 0, 0)]
    g = D.GroupBy([T.T_I64], aggs)
    g.set_pipelined(True)
    keep, data = [], []
    for b in range(nb):
        k = rng.integers(0, 3, n).astype(np.int64)
        a = rng.integers(1, 1000, n).astype(np.int64)
        d = rng.integers(1, 50, n).astype(np.int64)
        if b == bad:
            d[n // 2] = 0
        ck, ca, cd = D.Column.from_numpy(k), D.Column.from_numpy(a), D.Column.from_numpy(d)
        p = D.ExprProgram([ca, cd])
        q = p.arith(T.EX_DIVIDE, p.load(0), p.load(1))
        g.add_block_program([ck], p, [q, None], n)
        keep.append((ck, ca, cd, p))
        data.append((k, a, d))
    rc, committed = g.checkpoint(raise_on_error=False)
    assert rc == T.ERR_ROW_ERRORS and 0 <= committed <= bad
    got = sorted(g.result())
    kk = np.concatenate([data[b][0] for b in range(committed)]) if committed else np.zeros(0, np.int64)
    assert [(r[0], r[2]) for r in got] == [(int(key), int((kk == key).sum())) for key in np.unique(kk)]


def test_reset_drops_queued_blocks_and_a_second_stream_is_refused(gpu):
    D = gpu
    rng = np.random.default_rng(3)
    n = 50_000
    k = rng.integers(0, 4, n).astype(np.int64)
    x = rng.integers(-100, 100, n).astype(np.int64)
    g = D.GroupBy([T.T_I64], AGGS)
    g.set_pipelined(True)
    ck, cx, p, regs, f = _small_program(D, k, x)
    g.add_block_program([ck], p, regs, n)
    other = C.c_void_p()
    check(lib().dbhip_stream_create(C.byref(other)))
    with pytest.raises(T.DbhipError) as e:
        g.add_block_program([ck], p, regs, n, stream=other)
    assert e.value.code == T.ERR_INVALID
    g.reset()
    assert g.result() == []
    g.add_block_program([ck], p, regs, n, stream=other)      # after a reset (or a checkpoint) the table may move to another stream
    assert g.checkpoint(stream=other) == 1
    assert sorted(g.result()) == _expected(k, x)
    g.set_pipelined(False)
    g.add_block_program([ck], p, regs, n)                     # synchronous again
    assert sorted(g.result()) == _expected(np.concatenate([k, k]), np.concatenate([x, x]))
    check(lib().dbhip_stream_destroy(other))


def test_many_windows_and_a_table_that_has_to_grow(gpu):
    """Blocks whose workgroups each see <= 8 groups while the table as a whole collects thousands (a clustered key): the merges
    queued behind the windows must never push the table past its load factor — the host's bound makes it look (and grow) in time."""
    D = gpu
    n, nb = 16384, 300
    g = D.GroupBy([T.T_I64], AGGS, capacity=1024)
    g.set_pipelined(True)
    keep, ks, xs = [], [], []
    rng = np.random.default_rng(8)
    for b in range(nb):
        # rows in runs of 4096 equal keys: a workgroup's chunk range meets one or two of them
        k = (np.arange(n, dtype=np.int64) // 4096) + 4 * b
        x = rng.integers(-10**6, 10**6, n).astype(np.int64)
        ck, cx, p, regs, f = _small_program(D, k, x)
        g.add_block_program([ck], p, regs, n)
        keep.append((ck, cx, p))
        ks.append(k), xs.append(x)
    assert g.checkpoint() == nb
    assert sorted(g.result()) == _expected(np.concatenate(ks), np.concatenate(xs))


def test_plain_add_block_queues_on_a_pipelined_table(gpu):
    """dbhip_groupby_add_block (no fused program: what TransformPartialAggregate::transform calls per block) on a pipelined table goes
    through the same queue — the block's columns as a program without instructions — incl. a pushed-down filter Bitmap; a block whose
    keys are not for the few-groups kernel makes the checkpoint give the window back, after which the table takes blocks synchronously
    (and correctly) by itself."""
    D = gpu
    rng = np.random.default_rng(21)
    n, nb = 40_000, 50
    g = D.GroupBy([T.T_I64], AGGS)
    g.set_pipelined(True)
    keep, ks, xs, fs = [], [], [], []
    for b in range(nb):
        k = rng.integers(0, 5, n).astype(np.int64) - 2
        x = rng.integers(-10**9, 10**9, n).astype(np.int64)
        f = rng.random(n) < 0.7
        ck, cx = D.Column.from_numpy(k), D.Column.from_numpy(x)
        fb = D.Column.boolean(f)
        if b % 2:
            g.add_block([ck], [cx, None], n, filter=fb)
            fs.append(f)
        else:
            g.add_block([ck], [cx, None], n)
            fs.append(np.ones(n, bool))
        keep.append((ck, cx, fb))
        ks.append(k), xs.append(x)
    assert g.checkpoint() == nb
    K, X, F = np.concatenate(ks), np.concatenate(xs), np.concatenate(fs)
    exp = sorted((int(key), int(X[F & (K == key)].sum()), int((F & (K == key)).sum())) for key in np.unique(K[F]))
    assert sorted(g.result()) == exp
    # many groups: the window is given back once, then the table stops queueing plain blocks
    k = rng.integers(0, 5000, n).astype(np.int64)
    x = rng.integers(-100, 100, n).astype(np.int64)
    ck, cx = D.Column.from_numpy(k), D.Column.from_numpy(x)
    g2 = D.GroupBy([T.T_I64], AGGS)
    g2.set_pipelined(True)
    g2.add_block([ck], [cx, None], n)
    rc, committed = g2.checkpoint(raise_on_error=False)
    assert rc == T.ERR_CAPACITY and committed == 0
    g2.add_block([ck], [cx, None], n)            # synchronous now (LDS / partitioned paths)
    g2.add_block([ck], [cx, None], n)
    assert g2.checkpoint() == 0
    assert sorted(g2.result()) == _expected(np.concatenate([k, k]), np.concatenate([x, x]))


def test_multi_block_launches_carry_nullable_keys_arguments_and_filters(gpu):
    """The block table of a multi-block launch is packed by the query shape (fagg_device.h fa_blk_in_off / fa_blk_key_off /
    fa_blk_filter_off: validity pointers and offsets only where the shape has a Bitmap, the filter last). A shape that uses every part of
    the layout — nullable key, nullable arguments, a pushed-down filter on every block — queued as 40 blocks of one launch must equal
    the synchronous table fed the same blocks, and numpy."""
    import time
    D = gpu
    rng = np.random.default_rng(77)
    aggs = [(T.AGG_SUM, T.T_I64, 0, 0, 1), (T.AGG_COUNT, 0, 0, 0, 0), (T.AGG_MIN, T.T_I64, 0, 0, 1)]
    n, nb = 30_000, 40

    def table(pipelined):
        g = D.GroupBy([T.T_I64], aggs, [1])
        if pipelined:
            g.set_pipelined(True)
        return g

    def block():
        k = rng.integers(0, 3, n).astype(np.int64) * 7 - 7
        kv = rng.random(n) < 0.9
        x = rng.integers(-10**9, 10**9, n).astype(np.int64)
        xv = rng.random(n) < 0.8
        f = rng.random(n) < 0.6
        cols = (D.Column.from_numpy(k, validity=kv), D.Column.from_numpy(x, validity=xv), D.Column.boolean(f))
        return (k, kv, x, xv, f), cols

    def launches():
        out = (C.c_uint64 * 3)()
        check(lib().dbhip_fagg_stats(out))
        return out[0]

    # the multi-block kernel of this shape is compiled in the background on first sight: wait until two queued blocks take ONE launch
    warm = table(True)
    _, (ck, cx, fb) = block()
    for _ in range(240):
        before = launches()
        warm.add_block([ck], [cx, None, cx], n, filter=fb)
        warm.add_block([ck], [cx, None, cx], n, filter=fb)
        warm.checkpoint()
        if launches() - before == 1:
            break
        time.sleep(0.5)
    else:
        pytest.fail("the multi-block specialisation never arrived")
    pipe, sync = table(True), table(False)
    data, keep = [], []
    before = launches()
    for _ in range(nb):
        d, (ck, cx, fb) = block()
        pipe.add_block([ck], [cx, None, cx], n, filter=fb)
        data.append(d), keep.append((ck, cx, fb))
    assert pipe.checkpoint() == nb
    assert launches() - before == 1                      # 40 blocks, one launch
    for ck, cx, fb in keep:
        sync.add_block([ck], [cx, None, cx], n, filter=fb)
    got = sorted(pipe.result(), key=repr)
    assert got == sorted(sync.result(), key=repr) and len(got) == 4          # -7, 0, 7 and the NULL key
    K, KV, X, XV, F = (np.concatenate([d[i] for d in data]) for i in range(5))
    assert sum(r[2] for r in got) == int(F.sum())                             # count(*) over the rows the filter keeps
    assert sum(r[1] for r in got if r[1] is not None) == int(X[F & XV].sum())
    for r in got:
        sel = F & (~KV if r[0] is None else (KV & (K == r[0])))
        assert r[2] == int(sel.sum()) and r[1] == int(X[sel & XV].sum()) and r[3] == int(X[sel & XV].min())
