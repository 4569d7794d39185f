"""GPU: the C-ABI from several host threads at once (executor threads run many processor instances in parallel;
one instance is never run concurrently with itself). Every thread owns its stream, its columns and its handles; ctypes
releases the GIL for the duration of a call, so the calls really overlap."""
import ctypes as C
import threading

import numpy as np
import pytest

from databend_amd import _lib as T
from databend_amd._lib import check, lib

pytestmark = pytest.mark.gpu


def worker(gpu, tid, rounds, errors):
    try:
        D, L = gpu, lib()
        stream = C.c_void_p()
        check(L.dbhip_stream_create(C.byref(stream)))
        rng = np.random.default_rng(1000 + tid)
        n = 200_000 + 1000 * tid
        a = rng.integers(-10**6, 10**6, n).astype(np.int64)
        b = rng.integers(1, 1000, n).astype(np.int64)
        k = rng.integers(0, 50 + tid, n).astype(np.int64)
        ca, cb, ck = D.Column.from_numpy(a), D.Column.from_numpy(b), D.Column.from_numpy(k)
        exp_sum = a + b
        exp_groups = sorted((int(key), int(a[k == key].sum()), int((k == key).sum())) for key in np.unique(k))
        for r in range(rounds):
            # arithmetic + comparison + selection on this thread's stream
            out = D.DeviceBuffer(n * 8 + 64)
            xa, xb = ca.c(), cb.c()
            check(L.dbhip_arith(T.OP_PLUS, C.byref(xa), C.byref(xb), C.c_int64(n), T.T_I64, C.c_void_p(out.ptr), None, None, stream))
            bm = D.DeviceBuffer(((n + 63) // 64) * 8 + 8)
            check(L.dbhip_cmp(T.CMP_GT, C.byref(xa), C.byref(xb), C.c_int64(n), C.c_void_p(bm.ptr), stream))
            sel = D.DeviceBuffer(n * 4 + 64)
            cnt = D.DeviceBuffer(8)
            check(L.dbhip_filter_select(C.c_void_p(bm.ptr), C.c_int64(0), C.c_int64(n), C.c_void_p(sel.ptr), C.c_void_p(cnt.ptr), stream))
            check(L.dbhip_stream_sync(stream))
            assert np.array_equal(out.to_numpy(np.int64, n), exp_sum), f"thread {tid} round {r}: arith"
            assert int(cnt.to_numpy(np.uint64, 1)[0]) == int((a > b).sum()), f"thread {tid} round {r}: select"
            got_sel = sel.to_numpy(np.uint32, int((a > b).sum()))
            assert np.array_equal(got_sel, np.nonzero(a > b)[0].astype(np.uint32)), f"thread {tid} round {r}: selection order"
            # hash aggregation: own table, own stream
            g = D.GroupBy([T.T_I64], [(T.AGG_SUM, T.T_I64, 0, 0, 0), (T.AGG_COUNT, 0, 0, 0, 0)])
            g.add_block([ck], [ca, None], n, stream=stream)
            assert sorted(g.result()) == exp_groups, f"thread {tid} round {r}: group-by"
            # sort permutation (library stream: shared by all threads, calls are stream-ordered)
            perm = D.sort_perm([ck, ca], limit=100)
            order = np.lexsort((a, k))[:100]
            assert np.array_equal(k[perm], k[order]) and np.array_equal(a[perm], a[order]), f"thread {tid} round {r}: sort"
        check(L.dbhip_stream_destroy(stream))
    except BaseException as e:  # noqa: BLE001 — reported by the main thread
        errors.append((tid, repr(e)))


def test_c_abi_is_reentrant_across_host_threads(gpu):
    errors = []
    threads = [threading.Thread(target=worker, args=(gpu, t, 6, errors)) for t in range(6)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(300)
    assert not errors, errors
    assert all(not t.is_alive() for t in threads)


def test_one_thread_may_keep_several_streams_busy(gpu):
    """internal scratch is keyed by (thread, stream): asynchronous calls queued by ONE thread on TWO streams do not share a
    scratch buffer (r02's rule was one stream at a time per thread). Interleaved filter_select calls of different sizes — each
    call's block sums live in scratch until its kernels have run — and nothing is synchronised until everything is queued."""
    D, L = gpu, lib()
    s1, s2 = C.c_void_p(), C.c_void_p()
    check(L.dbhip_stream_create(C.byref(s1)))
    check(L.dbhip_stream_create(C.byref(s2)))
    rng = np.random.default_rng(77)
    jobs = []
    for r in range(12):
        n = int(rng.integers(3_000_000, 6_000_000)) if r % 2 == 0 else int(rng.integers(50_000, 90_000))
        bits = rng.random(n) < (0.3 if r % 3 else 0.9)
        packed = np.packbits(bits, bitorder="little")
        bm = D.DeviceBuffer.from_numpy(np.concatenate([packed, np.zeros(16, np.uint8)]))
        sel, cnt = D.DeviceBuffer(n * 4 + 64), D.DeviceBuffer(8)
        jobs.append((bits, bm, sel, cnt, n, s1 if r % 2 == 0 else s2))
    for bits, bm, sel, cnt, n, st in jobs:   # everything queued back to back, big and small alternating between the streams
        check(L.dbhip_filter_select(C.c_void_p(bm.ptr), C.c_int64(0), C.c_int64(n), C.c_void_p(sel.ptr), C.c_void_p(cnt.ptr), st))
    check(L.dbhip_stream_sync(s1))
    check(L.dbhip_stream_sync(s2))
    for bits, bm, sel, cnt, n, st in jobs:
        k = int(bits.sum())
        assert int(cnt.to_numpy(np.uint64, 1)[0]) == k
        assert np.array_equal(sel.to_numpy(np.uint32, k), np.nonzero(bits)[0].astype(np.uint32))
    check(L.dbhip_stream_destroy(s1))
    check(L.dbhip_stream_destroy(s2))
