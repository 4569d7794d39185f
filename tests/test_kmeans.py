"""Vector-cluster KMeans (src/query/storages/common/index/src/kmeans.rs) and the f32 VectorDistanceKernel (vector.rs): CPU —
the oracle's restatement against an independent numpy statement of the Avx summation order and of the LCG; GPU — dbhip_kmeans
/ dbhip_vec_kernel_f32 BIT-IDENTICAL to the oracle (assignments, distances, iteration count) for L1 / L2 / Dot."""
import ctypes as C

import numpy as np
import pytest

from tests import oracle_lib as O


def oracle_kmeans(dt, data, rpc):
    L = O.load()
    L.orc_kmeans.restype = C.c_int64
    data = np.ascontiguousarray(data, np.float32)
    rows, dim = data.shape
    a, d, it = np.zeros(rows, np.uint32), np.zeros(rows, np.float32), C.c_int()
    k = L.orc_kmeans(dt, data.ctypes.data_as(C.c_void_p), C.c_int64(rows), dim, C.c_int64(rpc), a.ctypes.data_as(C.c_void_p),
                     d.ctypes.data_as(C.c_void_p), C.byref(it))
    return a, d, int(k), it.value


def oracle_vdk(which, a, b):
    L = O.load()
    L.orc_vdk.restype = C.c_float
    return np.array([L.orc_vdk(which, x.ctypes.data_as(C.c_void_p), y.ctypes.data_as(C.c_void_p), len(x)) for x, y in zip(a, b)], np.float32)


def avx_numpy(which, x, y):
    """independent statement of impl_f32_{dot,l2_sqr,l1}_avx (vector.rs:190-260) with numpy float32 scalars; the fused
    multiply-add is emulated exactly in float64 (a 24 x 24-bit product is exact in 53 bits, one rounding at the end)"""
    f = np.float32
    n = len(x)
    m = n - n % 8
    v = [f(0)] * 8
    for i in range(0, m, 8):
        for j in range(8):
            a, b = x[i + j], y[i + j]
            if which == 0:
                v[j] = f(np.float64(a) * np.float64(b) + np.float64(v[j]))
            elif which == 1:
                d = f(a - b)
                v[j] = f(np.float64(d) * np.float64(d) + np.float64(v[j]))
            else:
                v[j] = f(v[j] + abs(f(a - b)))
    s = f(0)
    for j in range(8):
        s = f(s + v[j])
    t = f(0)
    for i in range(m, n):
        a, b = x[i], y[i]
        if which == 0:
            t = f(t + f(a * b))
        elif which == 1:
            d = f(a - b)
            t = f(t + f(d * d))
        else:
            t = f(t + abs(f(a - b)))
    return f(s + t)


def clustered(rng, rows, dim, k, spread=1.0):
    cent = (rng.normal(size=(k, dim)) * 4).astype(np.float32)
    return (cent[rng.integers(0, k, rows)] + rng.normal(size=(rows, dim)).astype(np.float32) * np.float32(spread)).astype(np.float32)


def test_distance_kernel_restatement_equals_the_avx_order_statement():
    rng = np.random.default_rng(1)
    for dim in (1, 7, 8, 9, 64, 77, 768):
        a = rng.normal(size=(6, dim)).astype(np.float32)
        b = rng.normal(size=(6, dim)).astype(np.float32)
        for which in (0, 1, 2):
            got = oracle_vdk(which, a, b)
            exp = np.array([avx_numpy(which, x, y) for x, y in zip(a, b)], np.float32)
            assert np.array_equal(got.view(np.uint32), exp.view(np.uint32)), (dim, which)


def test_kmeans_restatement_properties():
    """closed-form cases: k = ceil(rows / rows_per_cluster); one cluster -> all zeros; well separated blobs are recovered; the
    first centroid is row LCG(seed) % rows (kmeans.rs:252-255)"""
    rng = np.random.default_rng(2)
    data = clustered(rng, 3000, 16, 5, 0.05)
    a, d, k, it = oracle_kmeans(1, data, 3000)
    assert k == 1 and it == 0 and not a.any() and not d.any()
    a, d, k, it = oracle_kmeans(1, data, 600)
    assert k == 5 and 1 <= it <= 100
    # tight blobs: rows of one blob share one cluster id, and the distance is the distance to the blob's mean
    truth = {}
    for i in range(3000):
        truth.setdefault(a[i], []).append(i)
    assert len(truth) == 5
    for idx in truth.values():
        mean = data[idx].mean(axis=0)
        assert np.allclose(np.linalg.norm(data[idx] - mean, axis=1), d[idx], rtol=1e-3, atol=1e-4)
    state = (0xD47ABA5EC1A57E12 * 6364136223846793005 + 1442695040888963407) % (1 << 64)
    assert state % 3000 >= 0   # (the draw is checked end to end by the GPU test: host LCG vs the oracle's)


@pytest.mark.gpu
@pytest.mark.parametrize("dim", [1, 8, 13, 128, 768])
def test_device_distance_kernel_is_bit_identical(gpu, dim):
    rng = np.random.default_rng(dim)
    a = rng.normal(size=(1003, dim)).astype(np.float32)
    b = rng.normal(size=(1003, dim)).astype(np.float32)
    for which in (0, 1, 2):
        assert np.array_equal(gpu.vec_kernel_f32(which, a, b).view(np.uint32), oracle_vdk(which, a, b).view(np.uint32)), which


@pytest.mark.gpu
@pytest.mark.parametrize("dt", [0, 1, 2])
@pytest.mark.parametrize("rows,dim,rpc", [(2, 3, 1), (1000, 8, 100), (5000, 33, 500), (40_000, 128, 1000), (20_000, 768, 2000), (3000, 16, 47)])
def test_device_kmeans_is_bit_identical_to_the_restatement(gpu, dt, rows, dim, rpc):
    rng = np.random.default_rng(rows + dim + dt)
    data = clustered(rng, rows, dim, max(2, rows // rpc), 1.0)
    if dt == 2:   # Dot: vector_samples normalises the rows first
        L = O.load()
        for r in data:
            L.orc_normalize_vector(r.ctypes.data_as(C.c_void_p), dim)
    ea, ed, ek, eit = oracle_kmeans(dt, data, rpc)
    a, d, k, it = gpu.kmeans(dt, data, rpc)
    assert (k, it) == (ek, eit)
    assert np.array_equal(a, ea)
    assert np.array_equal(d.view(np.uint32), ed.view(np.uint32))


@pytest.mark.gpu
def test_device_kmeans_normalises_like_vector_samples_and_handles_degenerate_input(gpu):
    rng = np.random.default_rng(9)
    raw = clustered(rng, 6000, 24, 6, 0.5) * np.float32(3)
    raw[5] = 0   # a zero vector stays as it is (norm <= EPSILON)
    L = O.load()
    normed = raw.copy()
    for r in normed:
        L.orc_normalize_vector(r.ctypes.data_as(C.c_void_p), 24)
    ea, ed, ek, eit = oracle_kmeans(2, normed, 1000)
    a, d, k, it = gpu.kmeans(2, raw, 1000, normalize_input=True)
    assert (k, it) == (ek, eit) and np.array_equal(a, ea) and np.array_equal(d.view(np.uint32), ed.view(np.uint32))
    # all rows equal: every kmeans++ total is 0 -> the gen_range branch; clusters stay empty and take the "farthest" row
    same = np.tile(rng.normal(size=(1, 12)).astype(np.float32), (500, 1))
    ea, ed, ek, eit = oracle_kmeans(1, same, 100)
    a, d, k, it = gpu.kmeans(1, same, 100)
    assert (k, it) == (ek, eit) and np.array_equal(a, ea) and np.array_equal(d.view(np.uint32), ed.view(np.uint32))
