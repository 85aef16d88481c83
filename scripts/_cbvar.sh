#!/bin/bash
# k_cbuild variants: rebuild wall time of scripts/probe_pair.py per library, then issue counters of the product library
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$root/gpurun_out/$1
mkdir -p $out
cd $root
for v in product "$@"; do
    [ "$v" = "$1" ] && continue
    if [ "$v" = product ]; then unset AMM_LIB; else export AMM_LIB=$root/atomsmm_amd/exp/lib_$v.so AMM_ALLOW_TUNE=1; fi
    echo "== $v"
    timeout -k 10 200 python3 scripts/probe_pair.py --reps 20 2>&1 | grep -E "rebuild|near" | cut -c1-200
done
unset AMM_LIB
cd /tmp && export TMPDIR=/tmp
timeout -k 10 240 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU --output-format csv -d $out/pmc_sq -- python3 $root/scripts/probe_pair.py --reps 10 > $out/pmc_sq.log 2>&1
timeout -k 10 240 rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM SQ_LDS_BANK_CONFLICT --output-format csv -d $out/pmc_sq2 -- python3 $root/scripts/probe_pair.py --reps 10 > $out/pmc_sq2.log 2>&1
cd $root
python3 scripts/pmc_summary.py $out/pmc.txt $out/pmc_sq $out/pmc_sq2 > /dev/null
grep -E "k_cbuild<false" $out/pmc.txt | cut -c1-40,100-
