#!/bin/bash
# Counter passes of scripts/probe_pair.py (molecule-row kernels) on the GPU box; one counter set per pass, program directly
# after `--`.   gpurun --timeout 900 -- 'bash scripts/pmc_cluster.sh TAG [probe args]'   ->  gpurun_out/TAG_pmc.txt
set -o pipefail
tag=${1:-r03}
shift
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$root/gpurun_out
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
run_pass() {
    name=$1
    shift
    timeout -k 10 240 rocprofv3 --pmc "$@" --output-format csv -d "$out/${tag}_pmc_$name" -- python3 "$root/scripts/probe_pair.py" --reps 20 $EXTRA > "$out/${tag}_pmc_$name.log" 2> "$out/${tag}_pmc_$name.err" || echo "pass $name FAILED"
}
EXTRA="$@"
run_pass sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY
run_pass lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD
run_pass grbm GRBM_GUI_ACTIVE TA_TA_BUSY_sum
cd "$root"
python3 scripts/pmc_summary.py "$out/${tag}_pmc.txt" "$out/${tag}_pmc_sq1" "$out/${tag}_pmc_lds" "$out/${tag}_pmc_grbm" > /dev/null
python3 scripts/pmc_table.py "$out/${tag}_pmc.txt"
