#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $root
export AMM_ALLOW_TUNE=1
for v in cbw5 cbw5b6; do
  for parts in 2 3 4 5 6; do
    export AMM_LIB=$root/atomsmm_amd/exp/lib_$v.so
    echo "== $v parts $parts: $(timeout -k 10 200 python3 scripts/probe_pair.py --reps 10 --option build_parts=$parts 2>&1 | grep -E "rebuild alone" | sed 's/.*with a list/with a list/')"
  done
done
