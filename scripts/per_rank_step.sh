#!/bin/bash
# Per-rank step budget on ONE GPU: W ranks as threads (scripts/per_rank_step.py) under rocprofv3's kernel trace;
# every kernel's total duration / (W x steps) = its share of ONE rank's outer step.  The copy kernels that stand for the collectives
# are listed apart.   gpurun --timeout 1100 -- 'bash scripts/per_rank_step.sh TAG NSIDE STEPS "1 2 4 8" [STATE]'
tag=${1:-ranks}; nside=${2:-32}; steps=${3:-100}; worlds=${4:-"1 2 4 8"}; state=${5:-1}
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$root/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for w in $worlds; do
    rm -rf $out/w$w
    rocprofv3 --kernel-trace --output-format csv -d $out/w$w -- python3 $root/scripts/per_rank_step.py --world $w --nside $nside --steps $steps --warmup 20 --state $state > $out/w$w.json 2> $out/w$w.err || { tail -5 $out/w$w.err; exit 1; }
    python3 - <<PY
import csv, glob, json, collections
f = glob.glob('$out/w$w/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
info = json.loads([l for l in open('$out/w$w.json') if l.startswith('{')][-1])
W, K = info['world'], info['steps']
# the timed region: the last W x K x (launches per rank and step) kernels -- take the launches after the last warm-up by counting
# the pair kernels backwards: 2 per rank and step
pair = [i for i, r in enumerate(rows) if r['Kernel_Name'].startswith('void k_cpair<')]
first = pair[-2 * W * K] if len(pair) >= 2 * W * K else 0
d = collections.defaultdict(float)
n = collections.defaultdict(int)
for r in rows[first:]:
    name = r['Kernel_Name'].split('(')[0].replace('void ', '')
    d[name] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    n[name] += 1
copies = sum(v for k, v in d.items() if 'copy' in k.lower() or 'elementwise' in k)
work = sum(v for k, v in d.items() if not ('copy' in k.lower() or 'elementwise' in k))
print('W %d  atoms %d  lanes/row %s  state exchange %d  builds %d  | one rank, one outer step: %.1f us of kernels (+ %.1f us of copies standing for the collectives)' % (
    W, info['atoms'], info['lanes_per_row'], info['state'], info['builds'], work / (W * K), copies / (W * K)))
for k in sorted(d, key=lambda k: -d[k]):
    if d[k] / (W * K) >= 0.05:
        print('    %-58s %7.2f us/step  (%.2f launches/step, %.1f us each)' % (k[:58], d[k] / (W * K), n[k] / (W * K), d[k] / n[k]))
PY
    rm -rf $out/w$w
done
