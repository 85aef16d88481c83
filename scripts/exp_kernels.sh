#!/bin/bash
# kernel-time experiments: run bench.py under rocprofv3 --stats for a set of (label, env) pairs, print the pair-kernel averages
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$root/gpurun_out
cd /tmp && export TMPDIR=/tmp
run() {
    label=$1; shift
    ( export "$@"; timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/exp_$label -- python3 $root/bench.py --no-cpu-baseline --steps 100 --warmup 20 > $out/exp_$label.json 2>/dev/null )
    f=$(find $out/exp_$label -name "*kernel_stats.csv" | head -1)
    echo "== $label: $(python3 -c "import json;d=json.load(open('$out/exp_$label.json'));print(d['value'],'ns/day',d['ms_per_step'],'ms')" 2>/dev/null)"
    grep -E "k_pair_tab|k_pair_nlist|k_build_nlist<false" $f | awk -F'","' '{printf "   %-60s calls %s avg %.1f us\n", substr($1,2,60), $2, $4/1000}'
}
