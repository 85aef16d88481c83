#!/bin/bash
# Issue / occupancy / cache counter passes of bench.py on the GPU box.  One counter set per pass
# (SQ: 8 slots, TCC: 4 slots; --pmc alone, never with trace domains); the program directly after `--`.
#   gpurun --timeout 1100 -- 'bash scripts/pmc_issue.sh r02a'
# Results: gpurun_out/<tag>_pmc_issue.txt  (copy to profiles/ to have it judged)
set -o pipefail
tag=${1:-r02}
shift
extra="$@"
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$root/gpurun_out
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > "$out/${tag}_counters_avail.txt" 2>&1 || true
run_pass() {
    name=$1
    shift
    echo "pass $name: $*"
    timeout -k 10 240 rocprofv3 --pmc "$@" --output-format csv -d "$out/${tag}_pmc_$name" -- python3 "$root/bench.py" --no-cpu-baseline --steps 60 --warmup 10 $extra > "$out/${tag}_pmc_$name.json" 2> "$out/${tag}_pmc_$name.err" || echo "pass $name FAILED (see ${tag}_pmc_$name.err)"
}
run_pass sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY
run_pass sq2 SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS
run_pass lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_LDS_LOAD SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_INST_LEVEL_LDS
run_pass tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum
run_pass tcp TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum
run_pass ta TA_BUSY_avr TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum GRBM_GUI_ACTIVE
cd "$root"
python3 scripts/pmc_summary.py "$out/${tag}_pmc_issue.txt" "$out/${tag}_pmc_sq1" "$out/${tag}_pmc_sq2" "$out/${tag}_pmc_lds" "$out/${tag}_pmc_tcc" "$out/${tag}_pmc_tcp" "$out/${tag}_pmc_ta" > /dev/null
grep -E "k_pair_nlist|k_pair_tab|k_build_nlist|k_inner" "$out/${tag}_pmc_issue.txt" | cut -c1-75,108-
