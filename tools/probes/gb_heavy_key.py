"""One heavy group among many: `--share` of the rows carry ONE key (NULL with --null, else a value), the rest are uniform over --card groups.
Times add_block (sum + count) and prints the library's path choices for one traced call (DBHIP_TRACE)."""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from databend_amd import _lib as L          # noqa: E402
from databend_amd import device as D        # noqa: E402
from tools.microbench import Borrowed, col, timed   # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=60_000_000)
    ap.add_argument("--card", type=int, default=20000)
    ap.add_argument("--share", type=float, default=0.03)
    ap.add_argument("--null", action="store_true")
    args = ap.parse_args()
    torch.cuda.set_device(0)
    D.init(0)
    g = torch.Generator(device="cuda")
    g.manual_seed(1)
    n = args.n
    keys = torch.randint(0, args.card, (n,), device="cuda", dtype=torch.int64, generator=g)
    heavy = torch.rand(n, device="cuda", generator=g) < args.share
    vals = torch.randint(0, 1000, (n,), device="cuda", dtype=torch.int64, generator=g)
    if args.null:
        bits = (~heavy).to(torch.uint8).reshape(-1, 8)
        w = torch.tensor([1, 2, 4, 8, 16, 32, 64, 128], device="cuda", dtype=torch.uint8)
        vb = torch.cat([(bits * w).sum(dim=1).to(torch.uint8), torch.zeros(64, dtype=torch.uint8, device="cuda")])
        kc = col(keys, L.T_I64, validity=Borrowed(vb))
        gb = D.GroupBy([L.T_I64], [(L.AGG_SUM, L.T_I64, 0, 0, 0), (L.AGG_COUNT, 0, 0, 0, 0)], key_nullable=[1], capacity=max(1024, args.card * 2))
    else:
        keys[heavy] = args.card + 5
        kc = col(keys, L.T_I64)
        gb = D.GroupBy([L.T_I64], [(L.AGG_SUM, L.T_I64, 0, 0, 0), (L.AGG_COUNT, 0, 0, 0, 0)], capacity=max(1024, args.card * 2))
    vc = col(vals, L.T_I64)

    def f():
        gb.reset()
        gb.add_block([kc], [vc, None], n)
    f()
    os.environ["DBHIP_TRACE"] = "1"
    sys.stderr.write("---- traced call ----\n")
    f()
    del os.environ["DBHIP_TRACE"]
    sys.stderr.write("---- end ----\n")
    ms = timed(f, reps=3, warm=1)
    print(f"card={args.card} share={args.share} null={args.null}: {ms[0]:.3f} ms avg, {ms[1]:.3f} best, groups={gb.num_groups()}")


if __name__ == "__main__":
    main()
