"""GPU, BASELINE.json's FULL sizes: the oracle cannot finish these in seconds, so the HIP path is checked through
size-independent properties of the domain (partition invariance, closed-form / independent integer statements,
sortedness + permutation, self-retrieval), with inputs generated on the device (torch is plumbing here: random
columns and int64 reference sums; every result under test comes from libdbhip through the C-ABI)."""
import ctypes as C

import numpy as np
import pytest

from databend_amd import _lib as T

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


class Borrowed:
    """non-owning view of a torch tensor's storage with the DeviceBuffer surface"""

    def __init__(self, t):
        self.t = t
        self.ptr = t.data_ptr()
        self.nbytes = t.numel() * t.element_size()


def col(gpu, t, dtype, **kw):
    # the library runs on its own non-blocking stream: whatever torch queued to produce `t` must be complete
    torch.cuda.synchronize()
    return gpu.Column(dtype, t.shape[0], Borrowed(t), **kw)


def gen(seed):
    g = torch.Generator(device="cuda")
    g.manual_seed(seed)
    return g


def views_from_chars(chars):
    """[n] uint8 -> [n,16] uint8 inline views of 1-byte strings"""
    v = torch.zeros((chars.shape[0], 16), dtype=torch.uint8, device=chars.device)
    v[:, 0] = 1
    v[:, 4] = chars
    return v


def test_q1_sf10_partition_invariance_and_independent_integer_statement(gpu):
    """configs[1]: 59,986,052 lineitem rows. (1) the fused kernel over the whole column == the merge of the fused
    kernel over two row ranges (hash aggregation is a monoid homomorphism: exactly what the multi-GPU path relies
    on); (2) every state equals an independent torch int64 statement of Q1 (all SF10 totals fit i64)."""
    from databend_amd import tpch
    n = tpch.rows_for_sf(10)
    g = gen(2)
    dev = "cuda"
    qty = torch.randint(1, 51, (n,), device=dev, dtype=torch.int64, generator=g) * 100
    price = torch.randint(90000, 10494951, (n,), device=dev, dtype=torch.int64, generator=g)
    disc = torch.randint(0, 11, (n,), device=dev, dtype=torch.int64, generator=g)
    tax = torch.randint(0, 9, (n,), device=dev, dtype=torch.int64, generator=g)
    ship = torch.randint(tpch.SHIP_LO, tpch.SHIP_HI + 1, (n,), device=dev, dtype=torch.int32, generator=g)
    rf_c = torch.where(ship + 15 <= tpch.CURRENT, torch.where(torch.randint(0, 2, (n,), device=dev, generator=g) == 0, 65, 82), 78).to(torch.uint8)
    ls_c = torch.where(ship > tpch.CURRENT, 79, 70).to(torch.uint8)
    rf, ls = views_from_chars(rf_c), views_from_chars(ls_c)
    dec = dict(precision=15, scale=2)

    def run(lo, hi, table=None):
        t = table or gpu.GroupBy.q1()
        gpu.q1_fused(t, col(gpu, qty[lo:hi], T.T_DEC64, **dec), col(gpu, price[lo:hi], T.T_DEC64, **dec), col(gpu, disc[lo:hi], T.T_DEC64, **dec),
                     col(gpu, tax[lo:hi], T.T_DEC64, **dec), col(gpu, rf[lo:hi], T.T_STRING), col(gpu, ls[lo:hi], T.T_STRING),
                     col(gpu, ship[lo:hi], T.T_DATE), tpch.Q1_CUTOFF, hi - lo)
        return t

    whole = tpch.q1_rows(run(0, n))
    cut = 23_456_784  # a multiple of 16 keeps the second range's columns 16-byte aligned
    parts = run(0, cut)
    run(cut, n, parts)
    assert tpch.q1_rows(parts) == whole
    # independent statement
    keep = ship <= tpch.Q1_CUTOFF
    dp = price * (100 - disc)
    ch = dp * (100 + tax)
    seen = 0
    for (a, b), st in whole.items():
        m = keep & (rf_c == a[0]) & (ls_c == b[0])
        assert st["count"] == int(m.sum().item())
        assert st["sum_qty"] == int(qty[m].sum().item()) and st["sum_base_price"] == int(price[m].sum().item())
        assert st["sum_disc"] == int(disc[m].sum().item())
        assert st["sum_disc_price"] == int(dp[m].sum().item()) and st["sum_charge"] == int(ch[m].sum().item())
        seen += st["count"]
    assert seen == int(keep.sum().item()) and len(whole) == 4


@pytest.mark.parametrize("card", [4, 1000, 1300, 1600, 100_000, 1_000_000])
def test_groupby_60m_rows_equals_bincount(gpu, card):
    """AggregateHashTable at scale through the LDS / radix-partitioned / row paths: sum and count per group equal
    torch.bincount over the same device columns (exact int64), and adding the block twice doubles every state."""
    n = 60_000_000
    g = gen(7 + card)
    keys = torch.randint(0, card, (n,), device="cuda", dtype=torch.int64, generator=g)
    vals = torch.randint(0, 1000, (n,), device="cuda", dtype=torch.int64, generator=g)
    t = gpu.GroupBy([T.T_I64], [(T.AGG_SUM, T.T_I64, 0, 0, 0), (T.AGG_COUNT, 0, 0, 0, 0)], capacity=2 * card)
    t.add_block([col(gpu, keys, T.T_I64)], [col(gpu, vals, T.T_I64), None], n)
    exp_cnt = torch.bincount(keys, minlength=card).cpu().numpy()
    exp_sum = torch.bincount(keys, weights=vals.to(torch.float64), minlength=card).cpu().numpy()  # < 2^53: exact
    rows = sorted(t.result())
    assert [r[0] for r in rows] == [k for k in range(card) if exp_cnt[k]]
    assert [r[2] for r in rows] == [int(exp_cnt[r[0]]) for r in rows]
    assert [r[1] for r in rows] == [int(exp_sum[r[0]]) for r in rows]
    t.add_block([col(gpu, keys, T.T_I64)], [col(gpu, vals, T.T_I64), None], n)
    assert sorted(t.result()) == [(k, 2 * s, 2 * c) for k, s, c in rows]


def test_sort_64m_keys_is_a_sorted_permutation(gpu):
    """dbhip_sort_perm at 64 M rows: the output is a permutation (every row id once) and the keys read through it
    are non-decreasing with ties in ascending row id (stable); LIMIT 10 == its first 10 entries."""
    n = 64_000_000
    keys = torch.randint(-2**40, 2**40, (n,), device="cuda", dtype=torch.int64, generator=gen(3))
    keys[:1000] = 7  # a block of ties
    arr = (T.Col * 1)(col(gpu, keys, T.T_I64).c())
    z = (C.c_uint8 * 1)(0)
    perm = torch.empty(n, dtype=torch.int32, device="cuda")
    T.check(T.lib().dbhip_sort_perm(arr, z, z, 1, C.c_int64(n), C.c_int64(0), C.c_void_p(perm.data_ptr()), None))
    p = perm.to(torch.int64)
    assert int(torch.bincount(p, minlength=n).max().item()) == 1 and int(p.min().item()) == 0 and int(p.max().item()) == n - 1
    sk = keys[p]
    assert bool((sk[1:] >= sk[:-1]).all().item())
    tie = sk[1:] == sk[:-1]
    assert bool((p[1:][tie] > p[:-1][tie]).all().item())
    top = torch.empty(10, dtype=torch.int32, device="cuda")
    T.check(T.lib().dbhip_sort_perm(arr, z, z, 1, C.c_int64(n), C.c_int64(10), C.c_void_p(top.data_ptr()), None))
    assert torch.equal(top, perm[:10])


def test_join_150m_probe_rows_against_15m_build_rows(gpu):
    """Inner hash join at the Q3 build/probe ratio: unique build keys; every pair joins equal keys, the pair count
    equals the number of probe keys that occur in the build side (torch.isin), pairs are ordered by probe row."""
    nb, npr = 15_000_000, 150_000_000
    g = gen(11)
    bk = torch.randperm(4 * nb, device="cuda", generator=g)[:nb].to(torch.int64)
    pk = torch.randint(0, 4 * nb, (npr,), device="cuda", dtype=torch.int64, generator=g)
    j = gpu.HashJoin(nb)
    j.add_block(col(gpu, bk, T.T_U64))
    j.final_build()
    op, ob, m = j.probe_block_device(col(gpu, pk, T.T_U64))
    present = torch.zeros(4 * nb, dtype=torch.bool, device="cuda")
    present[bk] = True
    assert m == int(present[pk].sum().item())
    pi = torch.from_numpy(op.to_numpy(np.uint32, m).astype(np.int64)).cuda()
    bi = torch.from_numpy(ob.to_numpy(np.uint32, m).astype(np.int64)).cuda()
    assert bool((pk[pi] == bk[bi]).all().item())
    assert bool((pi[1:] > pi[:-1]).all().item())  # unique build keys: at most one pair per probe row, ascending
    marks = j.probe_mark(col(gpu, pk[:1_000_003], T.T_U64))
    assert np.array_equal(marks, present[pk[:1_000_003]].cpu().numpy())


@pytest.mark.parametrize("n_random", [56, 292])
def test_vector_index_10m_by_768_self_retrieval_and_exact_scan_agreement(gpu, n_random):
    """configs[4] on one GPU: 10,000,000 x 768 f32. Queries that ARE base rows must come back as their own nearest
    neighbour at cosine distance ~0 (self-retrieval), and the index must return the same ids as the exact f32 scan
    for random queries (recall@10 = 1.0). 64 queries go through the 128 x 128 filter kernel; 300 queries (a ragged second 256-query
    tile) through bf16_filter256_kernel, the kernel of the headline configuration, at its real base size."""
    n, dim, k = 10_000_000, 768, 10
    base = torch.randn((n, dim), device="cuda", dtype=torch.float32, generator=gen(5))
    ids = torch.tensor([0, 1, 8191, 8192, 156_249, 156_250, 4_999_999, n - 1], device="cuda")
    q = torch.cat([base[ids], torch.randn((n_random, dim), device="cuda", dtype=torch.float32, generator=gen(6))])
    nq = q.shape[0]

    class Vec:
        def __init__(self, t):
            self.n, self.dim, self.data = t.shape[0], t.shape[1], Borrowed(t)

    torch.cuda.synchronize()
    ix = gpu.VectorIndex(T.VEC_COSINE, Vec(base))
    idx, dist = ix.search(Vec(q), k)
    assert idx[: len(ids), 0].tolist() == ids.tolist()
    assert np.all(np.abs(dist[: len(ids), 0]) <= 2e-6)
    eidx, edist = gpu.vec_topk(T.VEC_COSINE, Vec(base), Vec(q), k)
    assert np.array_equal(idx, eidx)
    assert np.allclose(dist, edist, rtol=2e-5, atol=2e-6)
    assert np.all(np.diff(dist, axis=1) >= 0)
    ix.destroy()


def test_sharded_topk_device_merge_path(gpu):
    """The RCCL path's merge step (dist.merge_shard_topk_device) with a stand-in communicator of world size 3 that
    'gathers' three different shards' lists on one GPU: equals the numpy merge of the same candidates."""
    from databend_amd import dist as DX
    nq, k, world = 37, 10, 3
    g = gen(9)
    shards_i = [torch.randint(0, 1000, (nq, k), device="cuda", dtype=torch.int32, generator=g) for _ in range(world)]
    shards_d = [torch.rand((nq, k), device="cuda", dtype=torch.float32, generator=g).sort(dim=1)[0] for _ in range(world)]
    shards_i[1][:, 7:] = -1  # a short shard: empty slots
    shards_d[1][:, 7:] = float("inf")
    offs = [0, 1000, 2000]

    class FakeDist:
        def __init__(self):
            self.calls = 0

        def get_world_size(self):
            return world

        def all_gather_into_tensor(self, out, inp):
            # rank 0's view: its own contribution is `inp`; the other ranks' blocks are produced the way they would be
            src = shards_i if inp.dtype == torch.int32 else shards_d
            out = out.view(world, *inp.shape)   # rank-major concatenation, as the collective lays it out
            for r in range(world):
                if inp.dtype == torch.int32:
                    out[r] = torch.where(src[r] == -1, src[r], src[r] + offs[r])
                else:
                    out[r] = src[r]
            self.calls += 1

    def merge_dev(d_ptr, i_ptr, nq_, m, k_, oi_ptr, od_ptr):
        T.check(T.lib().dbhip_vec_topk_merge(C.c_void_p(d_ptr), C.c_void_p(i_ptr), C.c_int64(m), nq_, k_, C.c_void_p(oi_ptr), C.c_void_p(od_ptr), None))

    fd = FakeDist()
    oi, od = DX.merge_shard_topk_device(shards_i[0], shards_d[0], offs[0], k, fd, torch, lambda: T.check(T.lib().dbhip_stream_sync(None)), merge_dev)
    assert fd.calls == 2
    got_i, got_d = oi.cpu().numpy().view(np.uint32), od.cpu().numpy()
    for q in range(nq):
        cand = []
        for r in range(world):
            for j in range(k):
                i = int(shards_i[r][q, j].item())
                if i != -1:
                    cand.append((float(shards_d[r][q, j].item()), i + offs[r]))
        cand.sort()
        assert [(float(d), int(i)) for d, i in zip(got_d[q], got_i[q])] == cand[:k]
