"""Pair-pass times of ONE rank's slice (rank 0 of world W emulated on one GPU, no collectives) against lanes per atom.
usage: python scripts/probe_slices.py"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import numpy as np
import torch
from atomsmm_amd import backend as B
from atomsmm_amd.testing import tip3p_box
from test_gpu_abi_parity import near, hip_pair, dev, O

c = tip3p_box(32)
n = len(c['positions'])
rng = np.random.default_rng(1)
c['positions'] = c['positions'] + rng.normal(0.0, 0.02, c['positions'].shape)
dn = near('force-switch', 0.7, 0.5)
dd = O.desc(O.DAMPED, rc=1.0, rswitch=0.9, alpha=2.9, degree=1)


def timed(fn, reps=20):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / reps


for world, parts in ((1, 3), (1, 4), (2, 3), (2, 5), (4, 3), (4, 5), (4, 8), (8, 3), (8, 5), (8, 8)):
    for lpa in (8 if world == 1 else 16,):
        os.environ['AMM_LPA'] = str(lpa)
        os.environ['AMM_PARTS'] = str(parts)
        ctx = B.HipContext(n, c['box'], rank=0, world=world)
        fn = hip_pair(B, ctx, dn, c)
        ff = hip_pair(B, ctx, dd, c)
        ctx.pair_share_list(fn, ff)
        x, v, m = dev(c['positions']), dev(c['velocities']), dev(c['mass'])
        f = [torch.zeros((n, 3), dtype=torch.float64, device='cuda') for _ in range(3)]
        ctx.bind_state(x, v, m)
        for slot, buf in enumerate(f):
            ctx.bind_buffer(slot, buf)
        ctx.group_define(1, 1, [fn])
        ctx.group_define(2, 2, [ff])
        E = B.OP_EVAL
        t_near = timed(lambda: ctx.run_ops([B.Op(E, 1, 0, 0, 0.0)], 1))
        t_dual = timed(lambda: ctx.run_ops([B.Op(E, 1, 0, 0, 0.0), B.Op(E, 2, 0, 0, 0.0)], 1))
        def rebuild():          # a uniform shift beyond skin/2 triggers the rebuild and keeps the geometry
            x.add_(0.06)
            ctx.run_ops([B.Op(E, 1, 0, 0, 0.0)], 1)
        t_rebuild = timed(rebuild) - t_near
        print('parts %d' % parts, 'world %d lpa %2d: near %.1f us, dual %.1f us (incl. cell chain no-ops + sorted copies), rebuild %.1f us' %
              (world, lpa, t_near, t_dual, t_rebuild), flush=True)
        ctx.close()
