"""Tabulate a pmc_summary.py text file per kernel of interest (columns) and counter (rows)."""
import collections
import re
import sys

KERNELS = [('near', 'k_pair_nlist<2'), ('dual', 'k_pair_nlist<3'), ('tab_near', 'k_pair_tab<2'), ('tab_dual', 'k_pair_tab<3'),
           ('build(max)', 'k_build_nlist<false'), ('mol_near', 'k_cpair<2, 0, -1'), ('mol_outer', 'k_cpair<3, 1, -1'), ('mol_fused', 'k_cpair<3, 1, 2'),
           ('build_mol(max)', 'k_cbuild<false'), ('inner', 'k_inner_lanes')]
rows = collections.defaultdict(dict)
for line in open(sys.argv[1]):
    m = re.match(r'(\S+)\s+(.*?)\s+n=\s*(\d+) avg=\s*([\d.]+) max=\s*([\d.]+)', line)
    if not m:
        continue
    cn, kn, n, avg, mx = m.groups()
    for tag, pat in KERNELS:
        if pat in kn:
            rows[cn][tag] = float(mx if tag.startswith('build') else avg)
            break
cols = [t for t, _ in KERNELS if any(t in d for d in rows.values())]
print('%-30s' % 'counter' + ''.join('%15s' % c for c in cols))
for cn, d in rows.items():
    print('%-30s' % cn + ''.join('%15.0f' % d.get(c, 0) for c in cols))
