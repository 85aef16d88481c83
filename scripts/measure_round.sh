#!/bin/bash
# Round measurement on the GPU box: smoke, rocprofv3 kernel statistics of bench.py, the two PMC passes (HBM traffic),
# the bench line with the CPU baseline.  Everything lands in gpurun_out/<tag>_*; copy what is to be judged to profiles/.
#   gpurun --timeout 1100 -- 'bash scripts/measure_round.sh r01'
set -e -o pipefail
tag=${1:-r01}
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$root/gpurun_out
mkdir -p "$out"
cd "$root"
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/${tag}_stats" -- python3 "$root/bench.py" --no-cpu-baseline --pme-steps 0 > "$out/${tag}_bench_under_rocprof.json" 2> "$out/${tag}_rocprof.err"
echo "stats pass done"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$out/${tag}_pmc_fetch" -- python3 "$root/bench.py" --no-cpu-baseline --pme-steps 0 --steps 150 --warmup 10 > /dev/null 2>&1
echo "fetch pass done"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$out/${tag}_pmc_write" -- python3 "$root/bench.py" --no-cpu-baseline --pme-steps 0 --steps 150 --warmup 10 > /dev/null 2>&1
echo "write pass done"
rocprofv3 --pmc SQ_INSTS_VALU --output-format csv -d "$out/${tag}_pmc_valu" -- python3 "$root/bench.py" --no-cpu-baseline --pme-steps 0 --steps 150 --warmup 10 > /dev/null 2>&1
echo "valu pass done"
cd "$root"
python3 scripts/pmc_summary.py "$out/${tag}_pmc_summary.txt" --traffic-json "$out/${tag}_traffic.json" "$out/${tag}_pmc_fetch" "$out/${tag}_pmc_write" "$out/${tag}_pmc_valu" > /dev/null
cp "$out/${tag}_traffic.json" profiles/${tag}_traffic.json      # bench.py quotes the traffic measured on this box
cp "$(find "$out/${tag}_stats" -name '*kernel_stats.csv' | head -1)" "$out/${tag}_bench_kernel_stats.csv"
python3 bench.py > "$out/${tag}_bench.json" 2> "$out/${tag}_bench.err"
cat "$out/${tag}_bench.json"
