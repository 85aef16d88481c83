#!/bin/bash
# rocprofv3 kernel statistics of a bench.py invocation: bash scripts/kstats.sh TAG [bench args...]; prints the top kernels
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$root/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_stats -- python3 $root/bench.py "$@" > $out/${tag}.json 2> $out/${tag}.err
python3 - <<PY
import csv, glob, json
try:
    d = json.load(open("$out/${tag}.json")); print(d["value"], d["unit"], d["ms_per_step"], "ms/step")
except Exception as e:
    print("no bench json:", e)
f = glob.glob("$out/${tag}_stats/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:16]:
    print("%-72s calls %6s avg %9.1f us  %6s%%" % (r["Name"][:72], r["Calls"], float(r["AverageNs"]) / 1000, r["Percentage"][:6]))
PY
