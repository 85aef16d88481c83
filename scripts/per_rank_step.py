#!/usr/bin/env python3
"""One GPU stands for W ranks: the bench workload's step program (C3, or a bigger box with --nside) run by W ranks as threads of this
process (atomsmm_amd.engine.LocalWorld: the real slices, exchange chunks and launches; the all-gathers are device-to-device copies), for
the per-rank budgets of DESIGN.md section 5.  Under `rocprofv3 --kernel-trace --stats` the kernels' durations divided by W x steps are
a rank's step, term by term (scripts/per_rank_step.sh); the copies that stand for the collectives are not a rank's work.

    python scripts/per_rank_step.py --world 8 [--nside 32] [--steps 100] [--state 1]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--world', type=int, default=8)
    ap.add_argument('--nside', type=int, default=32)
    ap.add_argument('--steps', type=int, default=100)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--relax', type=int, default=110, help='velocity-rescaling steps before the warm-up (0: lattice start)')
    ap.add_argument('--state', type=int, default=1, help='1: ranks integrate their own molecules and exchange positions + velocities; 0: forces')
    ap.add_argument('--skin', type=float, default=None)
    args = ap.parse_args()
    import torch
    import bench
    from atomsmm_amd.engine import LocalWorld
    bench.EXTRA_OPTIONS.append(('state_exchange', str(args.state)))

    def job(rank):
        sim, case = bench.build_simulation(args.nside, (4, 2, 1), 4.0, skin=args.skin)
        eng = sim.context._engine
        done = 0
        while done < args.relax:            # (every rank rescales alike: the velocities are whole on every rank)
            sim.step(10)
            done += 10
            T = bench.temperature(eng, torch)
            eng.v.mul_((300.0 / T) ** 0.5)
        sim.step(args.warmup)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        sim.step(args.steps)
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        st = eng.ctx.pair_stats(eng.pair_force_ids(2)[0])
        out = dict(rank=rank, world=eng.world, elapsed=el, run_stats=eng.ctx.run_stats(), builds=st['n_builds'], lanes=st['lanes_per_atom'],
                   slice_atoms=st['n_slice_atoms'], T=bench.temperature(eng, torch), x0=float(eng.x[0, 0]), rlist=st['rlist'])
        eng._check()
        return out

    if args.world > 1:
        res = LocalWorld(args.world).run(job)
    else:
        res = [job(0)]
    r0 = res[0]
    assert all(r['x0'] == r0['x0'] for r in res), 'the ranks disagree'
    print(json.dumps(dict(world=args.world, atoms=3 * args.nside ** 3, steps=args.steps, state=args.state,
                          wall_ms_per_step_all_ranks=round(1e3 * max(r['elapsed'] for r in res) / args.steps, 4), lanes_per_row=r0['lanes'],
                          slice_atoms=[r['slice_atoms'] for r in res], builds=r0['builds'], run_stats=r0['run_stats'], rlist=r0['rlist'],
                          T=round(r0['T'], 1))), flush=True)


if __name__ == '__main__':
    main()
