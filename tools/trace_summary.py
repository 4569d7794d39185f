#!/usr/bin/env python3
"""Per (kernel, grid) duration summary of a rocprofv3 kernel trace CSV: count, median / min / max microseconds, and the gaps
between consecutive kernels of one stream. Usage: trace_summary.py <kernel_trace.csv> [top]"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
top = int(sys.argv[2]) if len(sys.argv) > 2 else 16
by = collections.defaultdict(list)
for r in rows:
    by[(r["Kernel_Name"].split("(")[0][-44:], int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1))].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
print(f"{len(rows)} dispatches")
for k, v in sorted(by.items(), key=lambda kv: -sum(kv[1]))[:top]:
    v.sort()
    print(f"{k[0]:46s} wgs {k[1]:6d} n {len(v):6d} med {v[len(v) // 2] / 1e3:8.1f} us min {v[0] / 1e3:8.1f} max {v[-1] / 1e3:8.1f} total {sum(v) / 1e6:8.2f} ms")
# gaps between back-to-back dispatches of the same queue
byq = collections.defaultdict(list)
for r in rows:
    byq[r["Queue_Id"]].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
gaps = []
for q, v in byq.items():
    v.sort()
    gaps += [b[0] - a[1] for a, b in zip(v, v[1:]) if 0 <= b[0] - a[1] < 200_000]
if gaps:
    gaps.sort()
    print(f"gaps between consecutive dispatches of a queue (< 200 us): n {len(gaps)} med {gaps[len(gaps) // 2] / 1e3:.1f} us p10 {gaps[len(gaps) // 10] / 1e3:.1f} p90 {gaps[len(gaps) * 9 // 10] / 1e3:.1f}")
