#!/usr/bin/env python3
"""Config C5: the host's synchronising read of deriv(energy, lambda) -- how long the host is blocked in it per AFED step and how
long it then takes to hand the GPU its next launch (the GPU idles for that long: nothing is queued behind a blocking read)."""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=60)
    ap.add_argument('--profile-window', action='store_true', help='cProfile of the host code between the read and the next run_ops')
    args = ap.parse_args()
    import torch
    import bench
    from atomsmm_amd import backend as B
    from atomsmm_amd import engine as E
    sim, case = bench.build_simulation_c5((4, 2, 1), 2.0)
    sim.step(30)
    torch.cuda.synchronize()
    acc = dict(blocked=0.0, reads=0, after=0.0, t_read=None, launches=0)
    import cProfile
    import pstats
    pr = cProfile.Profile() if args.profile_window else None
    orig_settle = E.Engine._settle
    orig_run = B.HipContext.run_ops

    def settle(self, containers):
        pending = self._pending is not None and self._pending[1] > 0
        t0 = time.perf_counter()
        orig_settle(self, containers)
        if pending:
            acc['blocked'] += time.perf_counter() - t0
            acc['reads'] += 1
            acc['t_read'] = time.perf_counter()
            if pr:
                pr.enable()

    def run_ops(self, *a, **k):
        if acc['t_read'] is not None:
            if pr:
                pr.disable()
            acc['after'] += time.perf_counter() - acc['t_read']
            acc['t_read'] = None
        acc['launches'] += 1
        return orig_run(self, *a, **k)
    E.Engine._settle = settle
    B.HipContext.run_ops = run_ops
    t0 = time.perf_counter()
    sim.step(args.steps)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    n = args.steps
    print('wall %.3f ms / step; %.1f blocking reads / step, host blocked %.3f ms / step; read -> next run_ops %.1f us each; %.1f run_ops / step' % (
        1e3 * wall / n, acc['reads'] / n, 1e3 * acc['blocked'] / n, 1e6 * acc['after'] / max(acc['reads'], 1), acc['launches'] / n))


    if pr:
        pstats.Stats(pr).sort_stats('tottime').print_stats(28)


if __name__ == '__main__':
    main()
