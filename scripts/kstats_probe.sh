#!/bin/bash
# rocprofv3 kernel statistics of a scripts/probe_pair.py invocation: bash scripts/kstats_probe.sh TAG [probe args...]
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$root/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_stats -- python3 $root/scripts/probe_pair.py "$@" > $out/${tag}.txt 2> $out/${tag}.err
grep -E "world|lib=" $out/${tag}.txt
python3 - <<PY
import csv, glob
f = glob.glob("$out/${tag}_stats/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:14]:
    print("%-72s calls %6s avg %9.1f us  %6s%%" % (r["Name"][:72], r["Calls"], float(r["AverageNs"]) / 1000, r["Percentage"][:6]))
PY
