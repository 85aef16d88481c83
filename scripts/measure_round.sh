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
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/${tag}_stats" -- python3 "$root/bench.py" --no-cpu-baseline --pme-steps 0 --no-legs > "$out/${tag}_bench_under_rocprof.json" 2> "$out/${tag}_rocprof.err"
echo "stats pass done"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$out/${tag}_pmc_fetch" -- python3 "$root/bench.py" --no-cpu-baseline --pme-steps 0 --no-legs --steps 150 --warmup 10 > /dev/null 2>&1
echo "fetch pass done"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$out/${tag}_pmc_write" -- python3 "$root/bench.py" --no-cpu-baseline --pme-steps 0 --no-legs --steps 150 --warmup 10 > /dev/null 2>&1
echo "write pass done"
rocprofv3 --pmc SQ_INSTS_VALU --output-format csv -d "$out/${tag}_pmc_valu" -- python3 "$root/bench.py" --no-cpu-baseline --pme-steps 0 --no-legs --steps 150 --warmup 10 > /dev/null 2>&1
echo "valu pass done"
cd "$root"
python3 scripts/pmc_summary.py "$out/${tag}_pmc_summary.txt" --traffic-json "$out/${tag}_traffic.json" "$out/${tag}_pmc_fetch" "$out/${tag}_pmc_write" "$out/${tag}_pmc_valu" > /dev/null
# the other BASELINE configs on the line (detail.c2 / detail.c5): their dominant kernels' HBM traffic, added to the same json
for cfg in c2 c5; do
    if [ $cfg = c2 ]; then st="--steps 300 --warmup 50"; else st="--steps 8 --warmup 4"; fi
    (cd /tmp && rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$out/${tag}_pmc_${cfg}_fetch" -- python3 "$root/bench.py" --config $cfg --no-cpu-baseline $st > /dev/null 2>&1)
    (cd /tmp && rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$out/${tag}_pmc_${cfg}_write" -- python3 "$root/bench.py" --config $cfg --no-cpu-baseline $st > /dev/null 2>&1)
    python3 scripts/pmc_summary.py "$out/${tag}_pmc_${cfg}_summary.txt" --traffic-json "$out/${tag}_traffic.json" --prefix ${cfg}_ "$out/${tag}_pmc_${cfg}_fetch" "$out/${tag}_pmc_${cfg}_write" > /dev/null
    echo "$cfg passes done"
done
cp "$out/${tag}_traffic.json" profiles/${tag}_traffic.json      # bench.py quotes the traffic measured on this box
cp "$(find "$out/${tag}_stats" -name '*kernel_stats.csv' | head -1)" "$out/${tag}_bench_kernel_stats.csv"
python3 bench.py > "$out/${tag}_bench.json" 2> "$out/${tag}_bench.err"
cat "$out/${tag}_bench.json"
