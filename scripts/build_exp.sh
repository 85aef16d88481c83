#!/bin/bash
# build experimental variants of the library: scripts/build_exp.sh NAME [NAME ...]  ->  atomsmm_amd/exp/lib_NAME.so  (-DAMM_EXP_NAME)
root=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $root/atomsmm_amd/exp
cd $root/atomsmm_amd/csrc
for v in "$@"; do
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DAMM_EXP_$v -o ../exp/lib_$v.so abi.hip pair.hip bonded.hip integrate.hip pme.hip expr.hip constraints.hip comm.hip -lhipfft -ldl 2>&1 | grep -E "error" &
done
wait
ls $root/atomsmm_amd/exp
