#!/bin/bash
# experimental variants of ONE kernel source: scripts/build_variant.sh FILE NAME "<flags>" [NAME "<flags>" ...]
#   -> atomsmm_amd/exp/lib_NAME.so  (csrc/FILE.hip compiled with <flags> into /tmp, every other object as built for the product)
# Select with AMM_LIB=atomsmm_amd/exp/lib_NAME.so.  Never touches atomsmm_amd/libatomsmm_hip.so or csrc/_obj.
set -e
cd "$(dirname "$0")/.."
O=atomsmm_amd/csrc/_obj
file=$1; shift
mkdir -p atomsmm_amd/exp
objs=""
for s in abi pair cluster group bonded integrate pme expr constraints comm; do
    [ "$s" = "$file" ] || objs="$objs $O/$s.o"
done
# the revision tag of a variant library ends in "-tune" (amm_kernel_revision): bench.py and the tests refuse it
printf 'extern "C" const char *amm_variant_tag(void) { return "variant"; }\n' > /tmp/var_tag.cpp
hipcc -fPIC -c /tmp/var_tag.cpp -o /tmp/var_tag.o
while [ $# -gt 1 ]; do
    name=$1; flags=$2; shift 2
    (
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $flags -c atomsmm_amd/csrc/$file.hip -o /tmp/var_$name.o -Rpass-analysis=kernel-resource-usage 2> /tmp/ru_$name.txt || { grep -E "error" -A3 /tmp/ru_$name.txt; exit 1; }
    hipcc --offload-arch=gfx950 -fPIC -shared -o atomsmm_amd/exp/lib_$name.so $objs /tmp/var_$name.o /tmp/var_tag.o -lhipfft -ldl
    echo "built atomsmm_amd/exp/lib_$name.so ($flags)"
    ) &
done
wait
