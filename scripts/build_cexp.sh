#!/bin/bash
# experimental variants of the molecule-row kernels: scripts/build_cexp.sh NAME "<flags>" [NAME "<flags>" ...]
#   -> atomsmm_amd/exp/lib_NAME.so  (cluster.hip compiled with -DAMM_CLUSTER_TUNE <flags>, the other objects as built)
set -e
cd "$(dirname "$0")/.."
O=atomsmm_amd/csrc/_obj
mkdir -p atomsmm_amd/exp
while [ $# -gt 1 ]; do
    name=$1; flags=$2; shift 2
    (
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DAMM_CLUSTER_TUNE $flags -c atomsmm_amd/csrc/cluster.hip -o /tmp/cexp_$name.o -Rpass-analysis=kernel-resource-usage 2> /tmp/ru_$name.txt || { grep -E "error" -A3 /tmp/ru_$name.txt; exit 1; }
    echo "== $name ($flags)"; grep -A9 "Function Name: _Z7k_cpair" /tmp/ru_$name.txt | grep -E "Name|VGPRs:|Scratch" | sed -e 's/.*remark: *//' -e 's/ \[-Rpass.*//' -e 's/Ev9CPairArgs.*//' | paste - - -
    hipcc --offload-arch=gfx950 -fPIC -shared -o atomsmm_amd/exp/lib_$name.so $O/abi.o $O/pair.o /tmp/cexp_$name.o $O/bonded.o $O/integrate.o $O/pme.o $O/expr.o $O/constraints.o $O/comm.o -lhipfft -ldl
    ) &
done
wait
ls atomsmm_amd/exp
