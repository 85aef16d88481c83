#!/bin/bash
# PMC passes over scripts/probe_pair.py (static kernel timing): bash scripts/pmc_probe.sh TAG [env assignments...]
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$root/gpurun_out
for kv in "$@"; do export "$kv"; done
cd /tmp && export TMPDIR=/tmp
pass() { name=$1; shift; timeout -k 10 120 rocprofv3 --pmc "$@" --output-format csv -d "$out/${tag}_pp_$name" -- python3 "$root/scripts/probe_pair.py" --reps 20 > "$out/${tag}_pp_$name.log" 2>&1 || echo "pass $name failed"; }
pass sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY
pass sq2 SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INST_LEVEL_SMEM SQ_IFETCH
pass ta TA_BUSY_avr TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE
cd "$root"
python3 scripts/pmc_summary.py "$out/${tag}_pp.txt" "$out/${tag}_pp_sq1" "$out/${tag}_pp_sq2" "$out/${tag}_pp_ta" > /dev/null
python3 scripts/pmc_table.py "$out/${tag}_pp.txt"
