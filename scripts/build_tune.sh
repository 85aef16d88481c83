#!/bin/bash
# kernel tuning: recompile csrc/cluster.hip with only the bench's instantiations (-DAMM_CLUSTER_TUNE, seconds) and relink
set -e
cd "$(dirname "$0")/.."
O=atomsmm_amd/csrc/_obj
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DAMM_CLUSTER_TUNE "$@" -c atomsmm_amd/csrc/cluster.hip -o $O/cluster.o -Rpass-analysis=kernel-resource-usage 2> /tmp/ru_cluster.txt || { grep -E "error" -A3 /tmp/ru_cluster.txt; exit 1; }
grep -A9 "Function Name: _Z7k_cpair\|Function Name: _Z8k_cbuildILb0ELb0" /tmp/ru_cluster.txt | grep -E "Name|VGPRs:|Scratch|Occupancy|SGPRs Spill" | sed -e 's/.*remark: *//' -e 's/ \[-Rpass.*//'
hipcc --offload-arch=gfx950 -fPIC -shared -o atomsmm_amd/libatomsmm_hip.so $O/abi.o $O/pair.o $O/cluster.o $O/bonded.o $O/integrate.o $O/pme.o $O/expr.o $O/constraints.o $O/comm.o -lhipfft -ldl
echo "tune build linked (REMEMBER: python -m atomsmm_amd.build --force before committing results)"
