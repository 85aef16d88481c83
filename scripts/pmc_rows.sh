#!/bin/bash
# cache / issue counters of the molecule-row traversal at two box sizes (C3 and C5): does a row cost more in the bigger box
# because its partners miss the L2?   gpurun --timeout 900 -- 'bash scripts/pmc_rows.sh r04p'
set -o pipefail
tag=${1:-r04p}
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$root/gpurun_out
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
run_pass() {
    cfg=$1; name=$2; steps=$3
    shift 3
    timeout -k 10 200 rocprofv3 --pmc "$@" --output-format csv -d "$out/${tag}_${cfg}_$name" -- python3 "$root/bench.py" --config $cfg --no-cpu-baseline --pme-steps 0 --steps $steps --warmup 4 > "$out/${tag}_${cfg}_$name.json" 2> "$out/${tag}_${cfg}_$name.err" || echo "pass $cfg $name FAILED"
}
for cfg in c3 c5; do
    steps=40; [ $cfg = c5 ] && steps=10
    run_pass $cfg sq $steps SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM
    run_pass $cfg tcc $steps TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum
    run_pass $cfg tcp $steps TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum
    run_pass $cfg fetch $steps FETCH_SIZE
    cd "$root"
    python3 scripts/pmc_summary.py "$out/${tag}_${cfg}.txt" "$out/${tag}_${cfg}_sq" "$out/${tag}_${cfg}_tcc" "$out/${tag}_${cfg}_tcp" "$out/${tag}_${cfg}_fetch" > /dev/null
    cd /tmp
    grep -E "k_cpair" "$out/${tag}_${cfg}.txt" | cut -c1-60,108-
done
