#!/usr/bin/env python3
"""How much of the bf16 filter's time is per-output-tile overhead (pipeline fill + bound test)? Same flop count at dim 768 / 1536 / 3072
(12 / 24 / 48 K-tiles per 256 x 256 output tile): if the kernel's TF rises with dim, the fill / epilogue of a tile is what it loses at
dim 768. Prints one JSON line per dim; run under rocprofv3 --kernel-trace --stats for the kernel's own time."""
import ctypes as C
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from databend_amd import _lib as L  # noqa: E402
from databend_amd import device as D  # noqa: E402
from databend_amd._lib import check, lib  # noqa: E402

D.init(0)
Lb = lib()
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev)
g.manual_seed(7)
nq = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
for dim in [int(x) for x in os.environ.get("ANN_DIMS", "768,1536,3072").split(",")]:
    n = 1_250_000 * 768 // dim
    n -= n % 256
    base = torch.randn((n, dim), device=dev, dtype=torch.float32, generator=g)
    q = torch.randn((nq, dim), device=dev, dtype=torch.float32, generator=g)
    ix = C.c_void_p()
    check(Lb.dbhip_vec_index_build(L.VEC_COSINE, C.c_void_p(base.data_ptr()), C.c_int64(n), dim, C.byref(ix), None))
    oi = torch.empty(nq * 10, dtype=torch.int32, device=dev)
    od = torch.empty(nq * 10, dtype=torch.float32, device=dev)
    f = lambda: check(Lb.dbhip_vec_index_search(ix, C.c_void_p(q.data_ptr()), nq, 10, C.c_void_p(oi.data_ptr()), C.c_void_p(od.data_ptr()), None))  # noqa: E731
    for _ in range(2):
        f()
    check(Lb.dbhip_stream_sync(None))
    t0 = time.perf_counter()
    reps = 5
    for _ in range(reps):
        f()
    check(Lb.dbhip_stream_sync(None))
    ms = (time.perf_counter() - t0) / reps * 1e3
    flops = 2.0 * n * dim * nq
    print(json.dumps({"dim": dim, "rows": n, "queries": nq, "search_ms": ms, "tflops_whole_search": flops / ms / 1e9, "k_tiles_per_output_tile": dim // 64}))
    check(Lb.dbhip_vec_index_destroy(ix))
    del base, q
