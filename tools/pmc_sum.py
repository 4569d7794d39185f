"""Sum a rocprofv3 counter_collection.csv per (kernel, counter): kernel,counter,dispatches,total,per_dispatch."""
import collections
import csv
import sys

agg = collections.defaultdict(float)
disp = collections.defaultdict(set)
for r in csv.DictReader(open(sys.argv[1])):
    k = (r["Kernel_Name"][:70], r["Counter_Name"])
    agg[k] += float(r["Counter_Value"])
    disp[k].add(r["Dispatch_Id"])
print("kernel,counter,dispatches,total,per_dispatch")
for (k, c), v in sorted(agg.items(), key=lambda kv: -kv[1]):
    n = max(1, len(disp[(k, c)]))
    print(f'"{k}",{c},{n},{v:.0f},{v / n:.1f}')
