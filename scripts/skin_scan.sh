#!/bin/bash
# Verlet-buffer scan at C3: ns/day, rebuilds per step and kernel times per skin / dual-list setting
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $root
for cfg in "--skin 0.06" "--skin 0.08" "--skin 0.10" "--skin 0.12" "--skin 0.15"; do
    python3 bench.py --no-cpu-baseline --pme-steps 0 --steps 300 $cfg 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); t=d['detail']
print('%-34s %7.1f ns/day  %.4f ms/step  near %.1f us  boundary pass %.1f us  list (re)builds/step %.2f  outer builds/step %.3f  near list pairs %d  far %d' % ('$cfg', d['value'], d['ms_per_step'], t['near_kernel_us'], t['step_boundary_pass_us'], t['far_list_prunes_in_timed_region']/d['steps'], t['outer_list_builds_in_timed_region']/d['steps'], t['near_list_pairs'], t['far_list_pairs']))"
done
