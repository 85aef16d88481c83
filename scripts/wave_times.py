#!/usr/bin/env python3
"""Per-wavefront wall clocks of the molecule-row traversal (a measurement build: scripts/build_variant.sh cluster wt
"-DAMM_CLUSTER_TUNE -DAMM_CPAIR_TIMING"; run e.g. scripts/probe_pair.py with AMM_ALLOW_TUNE=1 AMM_LIB=.../lib_wt.so
AMM_WAVE_TIMES=20 AMM_WAVE_TIMES_OUT=gpurun_out/wt/times: the 20th launch of each kernel writes [wavefront][4] u64 =
kernel entry, tables staged, tasks done (wall_clock64, 100 MHz), interior tasks << 8 | tasks).
    python scripts/wave_times.py gpurun_out/wt/times.fused
Files named *_epi come from the kernels that carry the inner RESPA loop as their epilogue: their second clock is "rows walked"."""
import sys

import numpy as np

for path in sys.argv[1:]:
    a = np.fromfile(path, dtype=np.uint64).reshape(-1, 4)
    t0 = a[:, 0].min()
    entry, staged, end = [(a[:, k] - t0).astype(float) / 100.0 for k in range(3)]
    work = end - staged
    slot = np.arange(len(a)) % 8
    print('%s: %d wavefronts, kernel %.1f us; tables staged after %.1f us (median); work per wavefront p10 / median / p90 / max = %.1f / %.1f / %.1f / %.1f us' % (
        path, len(a), end.max(), np.median(staged), np.percentile(work, 10), np.median(work), np.percentile(work, 90), work.max()))
    print('   mean wavefront lifetime / kernel time %.3f; wavefronts 0-3 of a block: median %.1f us, wavefronts 4-7: %.1f us' % (
        (end - entry).mean() / end.max(), np.median(work[slot < 4]), np.median(work[slot >= 4])))
    blk = np.arange(len(a)) // 8
    if path.endswith('_epi'):
        w3 = a[:, 3]
        parts = [((w3 >> np.uint64(sh)) & np.uint64(0xfffff)).astype(float) / 100.0 for sh in (0, 20, 40)]
        young = slot >= 4
        print('   epilogue of the wavefronts that run it alone (4-7), medians from "rows walked": own loads + this launch\'s forces visible %.1f us, '
              'preceding kicks done %.1f us, loop done %.1f us, stores issued / wavefront ends %.1f us' % (
                  np.median(parts[0][young]), np.median(parts[1][young]), np.median(parts[2][young]), np.median((end - staged)[young])))
        epi = end - staged
        print('   epilogue (inner RESPA loop of the rows\' molecules) per wavefront p10 / median / p90 / max = %.1f / %.1f / %.1f / %.1f us; '
              'rows walked: wavefronts 0-3 median %.1f us, 4-7 %.1f us, last %.1f us' % (
                  np.percentile(epi, 10), np.median(epi), np.percentile(epi, 90), epi.max(), np.median((staged - entry)[slot < 4]),
                  np.median((staged - entry)[slot >= 4]), staged.max()))
    print('   per XCD (block & 7), slowest wavefront: ' + ' '.join('%.1f' % end[(blk & 7) == x].max() for x in range(8)))
