#!/usr/bin/env python3
"""Time the list-free interaction-group kernel (csrc/group.hip) alone on the C5 system: softcore solute-solvent force, 30-atom solute
in ~249 000 atoms.  AMM_LIB selects an experimental build (scripts/build_variant.sh group ...)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    from atomsmm_amd import backend as B
    from atomsmm_amd.testing import solvated_chain
    case = solvated_chain()
    n = len(case['positions'])
    codes = np.full(n, 2.0)
    codes[case['solute']] = 1.0
    ctx = B.HipContext(n, case['box'])
    desc = B.pair_desc(B.SOFTCORE, 1.0, rswitch=0.9, alpha=0.6, flags=B.SWITCH, Kc=1.0)
    fid = ctx.pair_create(desc, codes, case['sigma'], case['epsilon'], case['exc_pairs'])
    pos = torch.as_tensor(case['positions'], device='cuda')
    f = torch.zeros((n, 3), dtype=torch.float64, device='cuda')
    for _ in range(5):
        ctx.force_eval(fid, pos, f)
    torch.cuda.synchronize()
    reps = 200
    t0 = time.perf_counter()
    for _ in range(reps):
        ctx.force_eval(fid, pos, f)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / reps
    ctx.profile_enable(True)
    for _ in range(reps):
        ctx.force_eval(fid, pos, f)
    nl, ms = ctx.profile_read(fid)
    print('%-40s wall %.1f us / launch, HIP events %.1f us (%d launches), |f|max %.6g' % (
        os.environ.get('AMM_LIB', 'product'), wall * 1e6, ms * 1e3 / max(nl, 1), nl, float(f.abs().max())))


if __name__ == '__main__':
    main()
