#!/bin/bash
# Per-rank budget on ONE GPU: the slice of rank W/2 of a world of W = 1, 2, 4, 8 at the relaxed C3 configuration -- kernel
# durations from rocprofv3's kernel trace (real rebuilds only: the conditional launches that found no flag are left out)
#   gpurun --timeout 900 -- 'bash scripts/per_rank_table.sh TAG'
tag=${1:-ranks}
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $root
for w in 1 2 4 8; do
    bash scripts/kstats_probe.sh ${tag}_w$w --cluster 1 --world $w --rank $((w / 2)) > /dev/null
    python3 - <<PY
import csv, glob, collections
f = glob.glob('gpurun_out/${tag}_w$w' + '_stats/**/*kernel_trace.csv', recursive=True)[0]
d = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    d[r['Kernel_Name']].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
def real(prefix):
    v = [x for k, vs in d.items() if k.startswith(prefix) for x in vs]
    big = [x for x in v if x > 0.5 * max(v)]
    return sum(big) / len(big)
def mean(prefix):
    v = [x for k, vs in d.items() if k.startswith(prefix) for x in vs]
    return sum(v) / len(v)
line = open('gpurun_out/${tag}_w$w.txt').read()
import re
lanes = re.search(r'lanes/row (\d+)', line).group(1)
print('W %d  lanes/row %2s  near %.1f  outer alone %.1f  fused pass %.1f  | rebuild: assign %.1f  sort+copies %.1f  build %.1f  | sorted copies alone %.1f  (us)' % (
    $w, lanes, mean('void k_cpair<2, 0, -1'), mean('void k_cpair<3, 1, -1'), mean('void k_cpair<3, 1, 2'), real('k_cassign'), real('k_csort_gather'), real('void k_cbuild<false'), mean('k_csort_gather')))
PY
done
