#!/usr/bin/env python3
"""Host -> device copy rate of this box (pinned and pageable host memory), for the PCIe-inclusive note of DESIGN.md §4: a cold-start Q1
moves 68 B per row over the link before the first kernel can run.

    python tools/probes/h2d_rate.py > gpurun_out/h2d_rate.json
"""
import json
import time

import torch


def rate(host, dev, reps=4):
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        dev.copy_(host, non_blocking=True)
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return host.numel() / best / 1e9


def main():
    nbytes = 2 << 30
    dev = torch.empty(nbytes, dtype=torch.uint8, device="cuda:0")
    pageable = torch.ones(nbytes, dtype=torch.uint8)
    pinned = torch.ones(nbytes, dtype=torch.uint8).pin_memory()
    out = {"bytes": nbytes, "pinned_GBps": rate(pinned, dev), "pageable_GBps": rate(pageable, dev)}
    out["q1_rows_per_s_cold_start_pinned"] = out["pinned_GBps"] * 1e9 / 68
    out["q1_rows_per_s_cold_start_pageable"] = out["pageable_GBps"] * 1e9 / 68
    print(json.dumps(out))


if __name__ == "__main__":
    main()
