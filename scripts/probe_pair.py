#!/usr/bin/env python3
"""Static timing of the pair kernels at C3 size through the C-ABI (kernel tuning harness).

    python scripts/probe_pair.py --make-config            # relax the lattice start with the product library, save positions
    [AMM_LIB=atomsmm_amd/exp/lib_X.so] python scripts/probe_pair.py [--reps 50]

The positions are fixed (a relaxed liquid configuration, scripts/_cache/c3_relaxed.npz), so an experimental build whose
arithmetic is deliberately wrong (a look-up removed, a gather short-circuited) still walks the same lists.  Prints the
average duration (HIP events on the launch stream, amm_profile_enable) of the stand-alone near evaluation, of the outer
force alone and of the dual pass (near + outer in one traversal), and of a forced list rebuild.
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CACHE = os.path.join(ROOT, 'scripts', '_cache', 'c3_relaxed.npz')


def make_config(out):
    import torch
    import bench
    sim, case = bench.build_simulation(32, (4, 2, 1), 4.0)
    bench.relax(sim, torch)
    sim.step(100)
    eng = sim.context._engine
    os.makedirs(os.path.dirname(out), exist_ok=True)
    np.savez_compressed(out, positions=eng.x.cpu().numpy(), velocities=eng.v.cpu().numpy())
    print('saved', out, 'T =', bench.temperature(eng, torch))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--make-config', action='store_true')
    ap.add_argument('--out', default=os.path.join(ROOT, 'gpurun_out', 'c3_relaxed.npz'))
    ap.add_argument('--reps', type=int, default=50)
    ap.add_argument('--skin', type=float, default=-1.0)
    ap.add_argument('--outer', choices=['damped', 'ewald'], default='damped')
    ap.add_argument('--world', type=int, default=1, help='emulate rank --rank of this many ranks (its slice of the rows, no collectives)')
    ap.add_argument('--rank', type=int, default=0)
    ap.add_argument('--cluster', type=int, default=1, help='1: molecule rows (product default for water), 0: per-atom rows')
    ap.add_argument('--nside', type=int, default=32, help='waters per box edge (32: the relaxed C3 configuration; others: lattice start + jitter)')
    ap.add_argument('--no-lj', action='store_true', help='all epsilons zero (diagnostics)')
    ap.add_argument('--compare-fused', action='store_true', help='forces of the fused pass against the stand-alone launches, bit for bit')
    ap.add_argument('--option', action='append', default=[], help='name=value context option (amm_set_option), repeatable')
    args = ap.parse_args()
    if args.make_config:
        return make_config(args.out)
    import torch
    from atomsmm_amd import backend as B
    from atomsmm_amd.testing import tip3p_box
    c = tip3p_box(args.nside)
    n = len(c['positions'])
    if args.nside != 32:
        c['positions'] = c['positions'] + np.random.default_rng(1).normal(0.0, 0.02, c['positions'].shape)
    elif os.path.exists(CACHE):
        c['positions'] = np.load(CACHE)['positions']
    else:
        print('warning: no relaxed configuration (%s): timing the lattice start' % CACHE)
    if args.no_lj:
        c['epsilon'] = np.zeros_like(c['epsilon'])
    dev = lambda a: torch.as_tensor(np.ascontiguousarray(a, dtype=np.float64), device='cuda')   # noqa: E731
    ctx = B.HipContext(n, c['box'], rank=args.rank, world=args.world)
    ctx.set_option('cluster', args.cluster)
    for item in args.option:
        name, value = item.split('=')
        ctx.set_option(name, float(value))
    dn = B.pair_desc(B.NEAR_FSWITCH, 0.7, rc0=0.7, rs0=0.5)
    if args.outer == 'damped':
        dd = B.pair_desc(B.DAMPED, 1.0, rswitch=0.9, alpha=2.9, degree=1)
    else:
        dd = B.pair_desc(B.NONBONDED, 1.0, rswitch=0.9, alpha=2.628260884878466, flags=B.COULOMB_EWALD | B.SWITCH)
    fn = ctx.pair_create(dn, c['charge'], c['sigma'], c['epsilon'], c['exc_pairs'], skin=args.skin)
    ff = ctx.pair_create(dd, c['charge'], c['sigma'], c['epsilon'], c['exc_pairs'], skin=args.skin)
    ctx.pair_share_list(fn, ff)
    x, v, m = dev(c['positions']), dev(np.zeros((n, 3))), dev(c['mass'])
    f = [torch.zeros((n, 3), dtype=torch.float64, device='cuda') for _ in range(4)]
    ctx.bind_state(x, v, m)
    for slot, buf in enumerate(f):
        ctx.bind_buffer(slot, buf)
    ctx.group_define(1, 1, [fn])
    ctx.group_define(2, 2, [ff])
    E = B.OP_EVAL
    near_only = [B.Op(E, 1, 0, 0, 0.0)]
    far_only = [B.Op(E, 2, 0, 0, 0.0)]
    dual = [B.Op(E, 1, 0, 0, 0.0), B.Op(E, 2, 0, 0, 0.0)]
    ctx.run_ops(dual, 2)
    ctx.check()
    torch.cuda.synchronize()
    if args.compare_fused:
        fused = [f[1].cpu().numpy().copy(), f[2].cpu().numpy().copy()]
        ctx.run_ops(near_only, 1)
        ctx.run_ops(far_only, 1)
        torch.cuda.synchronize()
        for name, a, b in (('near', fused[0], f[1].cpu().numpy()), ('outer', fused[1], f[2].cpu().numpy())):
            d = np.abs(a - b)
            print('%s: fused vs stand-alone max |diff| %.3e (max |f| %.3e), differing components %d of %d; rode_along %s' % (
                name, d.max(), np.abs(b).max(), int((d != 0).sum()), d.size, ctx.pair_stats(fn).get('rode_along')))
            if (d != 0).any():
                i = np.argwhere(d != 0)[:5]
                print('   first differing (atom, component):', i.tolist(), [float(d[tuple(k)]) for k in i])
        return
    res = {}
    for label, ops, fid in (('near', near_only, fn), ('far', far_only, ff), ('dual', dual, ff)):
        ctx.run_ops(ops, 3)
        ctx.profile_enable(True, only=fid)
        ctx.run_ops(ops, args.reps)
        torch.cuda.synchronize()
        cnt, ms = ctx.profile_read(fid)
        ctx.profile_enable(False)
        res[label] = ms / max(cnt, 1) * 1e3
    # wall time of a whole evaluation (list check + sorted copies + kernel) and of one with a forced list rebuild: a uniform
    # shift beyond skin / 2 triggers the rebuild and keeps the geometry
    def wall(fn, reps=20):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return 1e3 * e0.elapsed_time(e1) / reps

    def moved(ops):
        x.add_(1e-7)                  # positions "changed": the displacement check and the sorted copies run again
        ctx.set_positions_changed() if hasattr(ctx, 'set_positions_changed') else None
        ctx.run_ops(ops, 1)

    def rebuilt(ops):
        x.add_(0.06)
        ctx.run_ops(ops, 1)
    w_near = wall(lambda: ctx.run_ops(near_only, 1))
    w_dual = wall(lambda: ctx.run_ops(dual, 1))
    w_rebuild = wall(lambda: rebuilt(near_only))
    print('   world %d rank %d: evaluation wall time near %.1f us, dual %.1f us; with a list rebuild %.1f us (rebuild alone ~%.1f us)' % (
        args.world, args.rank, w_near, w_dual, w_rebuild, w_rebuild - w_near))
    st = ctx.pair_stats(ff)
    print('lib=%s  near %.1f us  far %.1f us  dual %.1f us   (list pairs near/far: %d / %d, lanes/row %d, rows per %s)' % (
        os.path.basename(os.environ.get('AMM_LIB', 'product')), res['near'], res['far'], res['dual'],
        ctx.pair_stats(fn)['n_list_pairs'], st['n_list_pairs'], st['lanes_per_atom'], 'molecule' if st['list_kind'] else 'atom'))
    within = (ctx.pair_count_within(fn, x, 0.7), ctx.pair_count_within(ff, x, 1.0))
    print('   entries inside the cutoffs: near %d (%.1f %%)  outer %d (%.1f %%)' % (
        within[0], 100.0 * within[0] / max(ctx.pair_stats(fn)['n_list_pairs'], 1), within[1], 100.0 * within[1] / max(st['n_list_pairs'], 1)))
    fsum = [float(b.abs().sum()) for b in f[1:3]]
    print('   checksum |f1| = %.10e  |f2| = %.10e' % tuple(fsum))
    ctx.close()


if __name__ == '__main__':
    main()
