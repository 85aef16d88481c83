#!/usr/bin/env python3
"""Config C5 at full size (249 075 atoms), 40 AFED steps (1 280 inner iterations, ~50 rebuilds of the molecule rows): the run with the
candidate walk of the list-free softcore force against the run that walks every atom every time -- positions, velocities and lambda
must agree bit for bit (development check, GPU; the test suite pins the same at 4 233 atoms and over 3 steps at full size)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(cand, steps):
    import bench
    sim, case = bench.build_simulation_c5((4, 2, 1), 2.0)
    eng = sim.context._engine
    eng.ctx.set_option('group_candidates', cand)
    sim.step(steps)
    eng._check()
    st = sim.context.getState(getPositions=True, getVelocities=True)
    soft = [s for s in (eng.ctx.pair_stats(p) for p in eng.pair_force_ids(0)) if s['list_kind'] == 3][0]
    builds = eng.ctx.pair_stats(eng.pair_force_ids(2)[0])['n_builds']
    return st.getPositions(asNumpy=True)._value, st.getVelocities(asNumpy=True)._value, sim.context.getParameter('lambda_vdw'), soft, builds


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    x1, v1, l1, s1, b1 = run(1, steps)
    x0, v0, l0, s0, b0 = run(0, steps)
    print('candidate walks %d (list of %d atoms), rebuilds %d / %d, lambda %.15g / %.15g' % (s1['n_candidate_walks'], s1['n_candidates'], b1, b0, l1, l0))
    same = np.array_equal(x1, x0) and np.array_equal(v1, v0) and l1 == l0
    print('bit for bit: %s   (max |dx| %.3e nm)' % (same, np.abs(x1 - x0).max()))
    return 0 if same and s1['n_candidate_walks'] > 0 and s0['n_candidate_walks'] == 0 else 1


if __name__ == '__main__':
    sys.exit(main())
