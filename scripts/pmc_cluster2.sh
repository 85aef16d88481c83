#!/bin/bash
# second counter set for the molecule-row kernels: instruction fetch, queue levels, lane use
set -o pipefail
tag=${1:-r03}
shift
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$root/gpurun_out
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
EXTRA="$@"
run_pass() {
    name=$1
    shift
    timeout -k 10 240 rocprofv3 --pmc "$@" --output-format csv -d "$out/${tag}_pmc_$name" -- python3 "$root/scripts/probe_pair.py" --reps 20 $EXTRA > "$out/${tag}_pmc_$name.log" 2> "$out/${tag}_pmc_$name.err" || echo "pass $name FAILED"
}
run_pass if SQ_IFETCH SQ_IFETCH_LEVEL SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVE_CYCLES
run_pass fl SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_CVT SQ_INSTS_BRANCH SQ_INSTS_VALU_INT64
cd "$root"
python3 scripts/pmc_summary.py "$out/${tag}_pmc2.txt" "$out/${tag}_pmc_if" "$out/${tag}_pmc_fl" > /dev/null
python3 scripts/pmc_table.py "$out/${tag}_pmc2.txt"
