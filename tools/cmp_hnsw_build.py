#!/usr/bin/env python3
"""One-off: recall of the device-built HNSW graph against the sequential CPU restatement of the reference's builder
(oracle/hnsw_oracle.c) at a size where the concurrency of the device build could matter. Test infrastructure only."""
import json
import sys
import time

import numpy as np

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from databend_amd import device as D  # noqa: E402
from tests import hnsw_oracle as H    # noqa: E402

n, dim, dist = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
METRIC = {"cosine": 0, "l2": 1, "l1": 3}
D.init(0)
L = H.lib()
rng = np.random.default_rng(11)
cent = rng.standard_normal((200, dim)).astype(np.float32)
raw = (cent[rng.integers(0, 200, n)] + 0.5 * rng.standard_normal((n, dim))).astype(np.float32)
queries = (cent[rng.integers(0, 200, 500)] + 0.5 * rng.standard_normal((500, dim))).astype(np.float32)
t0 = time.perf_counter()
idx = D.HnswIndex.build(METRIC[dist], D.VectorColumn(raw), m=10, ef_construct=40, seed=3)
t_dev = time.perf_counter() - t0
levels, lists, ep, el = idx.export_graph()
ids, _ = idx.search(D.VectorColumn(queries), 10)
data = H.preprocess(L, raw, dist)
quant = H.Quantised(L, data, dist)
pq = H.preprocess(L, queries, dist)
sc = idx.scores(D.VectorColumn(queries))
truth = np.argsort(sc, axis=1, kind="stable")[:, :10]           # quantised exhaustive = the best any graph can return
r_dev = np.mean([len(set(ids[i].tolist()) & set(truth[i].tolist())) for i in range(500)]) / 10
t0 = time.perf_counter()
g = H.Graph(L, n, 10, 40, levels)
g.build(raw, dist)
t_cpu = time.perf_counter() - t0
r_cpu = np.mean([len(set(g.search(quant, pq[i], 10)[0].tolist()) & set(truth[i].tolist())) for i in range(500)]) / 10
deg_dev = np.mean([len(lists[i]) for i in np.cumsum(np.r_[0, levels[:-1] + 1])])
deg_cpu = np.mean([len(g.links(p, 0)) for p in range(0, n, 7)])
print(json.dumps({"n": n, "dim": dim, "distance": dist, "graph_recall_device_build": r_dev, "graph_recall_sequential_reference_builder": r_cpu,
                  "device_build_s": t_dev, "cpu_sequential_build_s": t_cpu, "mean_degree_level0_device": float(deg_dev), "mean_degree_level0_cpu": float(deg_cpu)}))
