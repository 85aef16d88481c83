"""Print the top kernels of a rocprofv3 --kernel-trace --stats CSV (run on the GPU box after profiling)."""
import csv
import glob
import sys

d = sys.argv[1]
f = glob.glob(d + "/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[: int(sys.argv[2]) if len(sys.argv) > 2 else 14]:
    print(r["Name"][:70], r["Calls"], r["TotalDurationNs"], r["AverageNs"])
