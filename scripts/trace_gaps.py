"""Gaps between consecutive kernels of a rocprofv3 --kernel-trace csv (steady part of a bench run): per kernel name, how long the GPU
sat idle before it started.  usage: python scripts/trace_gaps.py <kernel_trace.csv> [skip_fraction]"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
rows = rows[int(len(rows) * skip):]
gap, dur, cnt = defaultdict(float), defaultdict(float), defaultdict(int)
for prev, cur in zip(rows, rows[1:]):
    name = cur['Kernel_Name'].split('(')[0][:60]
    gap[name] += max(0, int(cur['Start_Timestamp']) - int(prev['End_Timestamp']))
    dur[name] += int(cur['End_Timestamp']) - int(cur['Start_Timestamp'])
    cnt[name] += 1
total = (int(rows[-1]['End_Timestamp']) - int(rows[0]['Start_Timestamp'])) / 1e3
print('window %.1f us, kernels %d' % (total, len(rows)))
for name in sorted(cnt, key=lambda k: -dur[k]):
    print('%-62s n %6d  dur %8.2f us  gap before %6.2f us' % (name, cnt[name], dur[name] / cnt[name] / 1e3, gap[name] / cnt[name] / 1e3))
print('sum of durations %.1f us (%.1f %%), of gaps %.1f us' % (sum(dur.values()) / 1e3, 100 * sum(dur.values()) / 1e3 / total, sum(gap.values()) / 1e3))
