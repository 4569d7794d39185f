import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from databend_amd import device as D, tpch
from databend_amd._lib import check, lib
mode = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 59_986_052
D.init(0)
li = tpch.LineitemTorch(n, seed=2, torch=torch)
ref = tpch.q1_rows(tpch.q1_fused(li))
keep = []
for i in range(8):
    if mode == "keep":
        keep.append(tpch.q1_fused_program(li))
    elif mode == "drop":
        tpch.q1_fused_program(li)
    elif mode == "same":
        if not keep:
            keep.append(D.GroupBy.q1())
        keep[0].reset()
        tpch.q1_fused_program(li, keep[0])
    elif mode == "check":
        assert tpch.q1_rows(tpch.q1_fused_program(li)) == ref
    check(lib().dbhip_stream_sync(None))
    print(mode, "iteration", i, "ok", file=sys.stderr, flush=True)
print(mode, "done")
