#!/usr/bin/env python3
"""HNSW reference-comparable mode: m=10, ef_construct=40, cosine, ef = 4k — build time, queries/s and
recall@10 against the exact top-10, beside the exact index of the same base."""
import argparse
import json
import sys
import time

import numpy as np
import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from databend_amd import device as D  # noqa: E402
from databend_amd import _lib as T    # noqa: E402
import ctypes as C                    # noqa: E402


def run(rows=1_000_000, dim=768, queries=10_000, k=10, reps=3, clusters=0, metric="cosine", normalize=False, cpu=0):
    a = argparse.Namespace(rows=rows, dim=dim, queries=queries, k=k, reps=reps, clusters=clusters, metric=metric, normalize=normalize, cpu=cpu)
    return _run(a)


def cpu_search(a, idx, base, queries, dev_ids):
    """cpu_baseline leg: the CPU restatement of GraphLayers::search + the u8 scorer (oracle/hnsw_oracle.c, one thread) over the graph
    the device built, on the first a.cpu queries; also checks that the device returned the same ids"""
    import ctypes as C
    from tests import hnsw_oracle as H
    L = H.lib()
    levels = np.zeros(a.rows, dtype=np.int32)
    nlists, ep, el = C.c_int64(), C.c_uint32(), C.c_int32()
    lib = D.lib()
    D.check(lib.dbhip_hnsw_export_graph(idx.h, levels.ctypes.data_as(C.c_void_p), None, None, C.byref(nlists), C.byref(ep), C.byref(el), None))
    nl = np.zeros(nlists.value, dtype=np.int32)
    D.check(lib.dbhip_hnsw_export_graph(idx.h, None, None, nl.ctypes.data_as(C.c_void_p), None, None, None, None))
    flat = np.zeros(max(int(nl.sum()), 1), dtype=np.uint32)
    D.check(lib.dbhip_hnsw_export_graph(idx.h, None, flat.ctypes.data_as(C.c_void_p), None, None, None, None, None))
    g = H.Graph(L, a.rows, 10, 40, levels)
    L.orc_hnsw_graph_import(g.h, H.ptr(flat), H.ptr(nl), C.c_uint32(ep.value), C.c_int(el.value))
    raw = base.cpu().numpy()
    data = H.preprocess(L, raw, a.metric)
    quant = H.Quantised(L, data, a.metric)
    pq = H.preprocess(L, queries[:a.cpu].cpu().numpy(), a.metric)
    t0 = time.perf_counter()
    out = [g.search(quant, pq[i], a.k)[0] for i in range(a.cpu)]
    dt = time.perf_counter() - t0
    same = all(np.array_equal(out[i], dev_ids[i][:len(out[i])]) for i in range(a.cpu))
    g.free()
    return {"value": a.cpu / dt, "unit": "queries/s", "cores": 1, "kind": "port", "sample": f"{a.cpu} queries, oracle/hnsw_oracle.c over the device-built graph",
            "device_ids_equal_cpu_ids": bool(same)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=1_000_000)
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--queries", type=int, default=10_000)
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--clusters", type=int, default=0, help="0: i.i.d. N(0,1) (SURVEY C5); > 0: that many Gaussian clusters (centres N(0,1), spread 0.3)")
    ap.add_argument("--metric", default="cosine", choices=["cosine", "l2", "l1"])
    ap.add_argument("--normalize", action="store_true", help="unit-length base vectors (what embedding models emit)")
    ap.add_argument("--cpu", type=int, default=0, help="time the CPU restatement of the reference's search (oracle/, one thread) on this many queries over the SAME graph")
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    res = _run(a)
    print(json.dumps(res))
    if a.out:
        json.dump(res, open(a.out, "w"), indent=1)


def _run(a):
    D.init(0)
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev)
    g.manual_seed(5)
    if a.clusters > 0:
        cent = torch.randn(a.clusters, a.dim, device=dev, generator=g, dtype=torch.float32)
        base = cent[torch.randint(0, a.clusters, (a.rows,), device=dev, generator=g)] + 0.3 * torch.randn(a.rows, a.dim, device=dev, generator=g, dtype=torch.float32)
        queries = cent[torch.randint(0, a.clusters, (a.queries,), device=dev, generator=g)] + 0.3 * torch.randn(a.queries, a.dim, device=dev, generator=g, dtype=torch.float32)
    else:
        base = torch.randn(a.rows, a.dim, device=dev, generator=g, dtype=torch.float32)
        queries = torch.randn(a.queries, a.dim, device=dev, generator=g, dtype=torch.float32)

    if a.normalize:
        base = base / base.norm(dim=1, keepdim=True)
    MET = {"cosine": T.VEC_COSINE, "l2": T.VEC_L2, "l1": T.VEC_L1}[a.metric]

    class V:   # a borrowed device tensor in the shape HnswIndex / VectorIndex expect
        def __init__(self, t):
            self.t = t
            self.n, self.dim = t.shape
            self.data = type("P", (), {"ptr": t.data_ptr()})()
    vb, vq = V(base), V(queries)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    idx = D.HnswIndex.build(MET, vb, m=10, ef_construct=40, seed=5)
    torch.cuda.synchronize()
    build_s = time.perf_counter() - t0
    best = 1e9
    for _ in range(a.reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ids, dist = idx.search(vq, a.k)
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    # ground truth: the exact index (recall 1.0 by construction)
    ex = D.VectorIndex(MET, vb)
    eids, _ = ex.search(vq, a.k)
    hits = sum(len(set(ids[i].tolist()) & set(eids[i].tolist())) for i in range(a.queries))
    # what the quantiser alone costs: top-k by quantised distance over ALL rows (generate_scores) for the first 64 queries
    nqx = min(64, a.queries)
    sc = idx.scores(V(queries[:nqx].contiguous()))
    qx = np.argsort(sc, axis=1, kind="stable")[:, :a.k]
    hits_q = sum(len(set(qx[i].tolist()) & set(eids[i].tolist())) for i in range(nqx))
    hits_g = sum(len(set(ids[i].tolist()) & set(qx[i].tolist())) for i in range(nqx))
    tb = 1e9
    for _ in range(a.reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ex.search(vq, a.k)
        torch.cuda.synchronize()
        tb = min(tb, time.perf_counter() - t0)
    dist_name = "N(0,1) i.i.d." if a.clusters == 0 else str(a.clusters) + " Gaussian clusters"
    res = {"workload": f"HNSW reference-comparable mode: {a.rows} x {a.dim} f32 {dist_name}{' unit length' if a.normalize else ''}, {a.metric}, m=10 ef_construct=40, {a.queries} queries, k={a.k}, ef={4 * a.k}, u8-quantised scoring",
           "build_seconds": build_s, "build_points_per_s": a.rows / build_s, "search_ms": best * 1e3, "queries_per_s": a.queries / best,
           "recall_at_10_vs_exact": hits / (a.queries * a.k),
           "quantised_exhaustive_recall_vs_exact": hits_q / (nqx * a.k), "graph_recall_vs_quantised_exhaustive": hits_g / (nqx * a.k),
           "exact_index_queries_per_s": a.queries / tb, "exact_index_recall": 1.0}
    if a.cpu > 0:
        res["cpu_baseline"] = cpu_search(a, idx, base, queries, ids)
    del idx, ex, base, queries
    torch.cuda.empty_cache()
    return res


if __name__ == "__main__":
    main()
