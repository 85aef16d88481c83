#!/usr/bin/env python3
"""can the NEXT neighbour list be built on a second stream while the pair kernels run?  Two contexts over the
same relaxed C3 configuration, each on a stream of its own: A evaluates pair forces (the stand-alone near pass, or the fused
step-boundary pass) back to back; B rebuilds its molecule-row list at every evaluation (positions shifted beyond the Verlet buffer
each time: assign + sort + build, then its own near pass).  Timed: A alone, B alone, both interleaved -- if the two streams shared the
chip the interleaved time would approach max(A, B); if they cannot, it is A + B.

    python scripts/probe_overlap.py [--reps 40]"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CACHE = os.path.join(ROOT, 'scripts', '_cache', 'c3_relaxed.npz')


def make(B, torch, c, stream):
    n = len(c['positions'])
    dev = lambda a: torch.as_tensor(np.ascontiguousarray(a, dtype=np.float64), device='cuda')   # noqa: E731
    with torch.cuda.stream(stream):
        ctx = B.HipContext(n, c['box'], stream=stream.cuda_stream)
        fn = ctx.pair_create(B.pair_desc(B.NEAR_FSWITCH, 0.7, rc0=0.7, rs0=0.5), c['charge'], c['sigma'], c['epsilon'], c['exc_pairs'])
        ff = ctx.pair_create(B.pair_desc(B.DAMPED, 1.0, rswitch=0.9, alpha=2.9, degree=1), c['charge'], c['sigma'], c['epsilon'], c['exc_pairs'])
        ctx.pair_share_list(fn, ff)
        x, v, m = dev(c['positions']), dev(np.zeros((n, 3))), dev(c['mass'])
        f = [torch.zeros((n, 3), dtype=torch.float64, device='cuda') for _ in range(4)]
        ctx.bind_state(x, v, m)
        for slot, buf in enumerate(f):
            ctx.bind_buffer(slot, buf)
        ctx.group_define(1, 1, [fn])
        ctx.group_define(2, 2, [ff])
        near = [B.Op(B.OP_EVAL, 1, 0, 0, 0.0)]
        dual = [B.Op(B.OP_EVAL, 1, 0, 0, 0.0), B.Op(B.OP_EVAL, 2, 0, 0, 0.0)]
        ctx.run_ops(dual, 2)
        ctx.check()
    return dict(ctx=ctx, x=x, keep=(v, m, f), near=near, dual=dual, stream=stream)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--reps', type=int, default=40)
    args = ap.parse_args()
    import torch
    from atomsmm_amd import backend as B
    from atomsmm_amd.testing import tip3p_box
    c = tip3p_box(32)
    if os.path.exists(CACHE):
        c['positions'] = np.load(CACHE)['positions']
    else:
        c['positions'] = c['positions'] + np.random.default_rng(1).normal(0.0, 0.02, c['positions'].shape)
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    A, Bc = make(B, torch, c, sa), make(B, torch, c, sb)

    def run_a(which, k):
        with torch.cuda.stream(sa):
            A['ctx'].run_ops(A[which], k)

    def run_b():
        with torch.cuda.stream(sb):
            Bc['x'].add_(0.06)                      # beyond skin / 2: this evaluation rebuilds the list
            Bc['ctx'].run_ops(Bc['near'], 1)

    def timed(fn):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) * 1e6

    for which, per_b in (('near', 3), ('dual', 1)):
        def a_only():
            for _ in range(args.reps):
                run_a(which, per_b)

        def b_only():
            for _ in range(args.reps):
                run_b()

        def both():
            for _ in range(args.reps):
                run_b()
                run_a(which, per_b)
        ta, tb, tab = timed(a_only), timed(b_only), timed(both)
        print('A = %d x %-4s pass per round, B = one rebuild + near pass per round, %d rounds:  A alone %.1f us/round   B alone %.1f us/round   '
              'A and B on two streams %.1f us/round   (sum %.1f, max %.1f: overlap recovered %.0f %% of the smaller)' % (
                  per_b, which, args.reps, ta / args.reps, tb / args.reps, tab / args.reps, (ta + tb) / args.reps, max(ta, tb) / args.reps,
                  100.0 * (ta + tb - tab) / max(min(ta, tb), 1e-9)))
    A['ctx'].close()
    Bc['ctx'].close()


if __name__ == '__main__':
    main()
