"""Sum a rocprofv3 counter_collection.csv per (kernel, counter): kernel,counter,dispatches,total,per_dispatch."""
import collections
import csv
import sys

agg = collections.defaultdict(float)
disp = collections.defaultdict(set)
per = collections.defaultdict(float)   # (kernel, counter, dispatch) -> value (a counter comes in one row per XCD / instance)
for r in csv.DictReader(open(sys.argv[1])):
    k = (r["Kernel_Name"][:70], r["Counter_Name"])
    agg[k] += float(r["Counter_Value"])
    disp[k].add(r["Dispatch_Id"])
    per[k + (r["Dispatch_Id"],)] += float(r["Counter_Value"])
mx = collections.defaultdict(float)
for (kn, c, _d), v in per.items():
    mx[(kn, c)] = max(mx[(kn, c)], v)
print("kernel,counter,dispatches,total,per_dispatch,max_dispatch")
for (k, c), v in sorted(agg.items(), key=lambda kv: -kv[1]):
    n = max(1, len(disp[(k, c)]))
    print(f'"{k}",{c},{n},{v:.0f},{v / n:.1f},{mx[(k, c)]:.0f}')
