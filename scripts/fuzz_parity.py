#!/usr/bin/env python3
"""Wider sweeps of the randomised GPU parity checks than the test-suite runs (development tool; needs an MI355X):

    python scripts/fuzz_parity.py [--seeds 60] [--sliced 30]

* `tests/test_gpu_abi_parity.py::test_random_boxes_and_site_mixes_vs_oracle` over many more seeds (uniform mixes, density /
  site gradients, the ball-in-gas configuration);
* emulated multi-rank evaluations: W contexts with rank r of W each evaluate their slice of a random box with the force-only
  kernel; the sum of the slices against the oracle (1e-9 of the largest force), W in {2, 3, 5, 8}.
Prints the failing seeds; exit status 1 if any."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
os.chdir(ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--seeds', type=int, default=60)
    ap.add_argument('--sliced', type=int, default=30)
    args = ap.parse_args()
    import numpy as np
    import torch
    import test_gpu_abi_parity as T
    from test_gpu_abi_parity import near, hip_pair, dev, O
    B = T._backend()
    bad = []
    seeds = list(range(100, 100 + args.seeds)) + [31 + 6 * k for k in range(1, 12)] + [21 + 6 * k for k in range(1, 8)]
    for seed in seeds:
        try:
            T.test_random_boxes_and_site_mixes_vs_oracle(seed)
        except Exception as exc:          # noqa: BLE001  (a sweep: report and go on)
            bad.append(('box', seed))
            print('random box, seed', seed, 'FAILED:', repr(exc)[:200], flush=True)
    print('random boxes: %d configurations' % len(seeds), flush=True)
    for seed in range(200, 200 + args.sliced):
        rng = np.random.default_rng(seed)
        box = rng.uniform(2.4, 4.2, 3)
        n = int(rng.integers(900, 2400))
        m = int(np.ceil(n ** (1 / 3)))
        grid = np.stack(np.meshgrid(*[np.arange(m)] * 3, indexing='ij'), -1).reshape(-1, 3)[rng.permutation(m ** 3)[:n]]
        pos = (grid + 0.5 + rng.uniform(-0.28, 0.28, (n, 3))) / m * box
        has_site = rng.random(n) < rng.choice([0.15, 1 / 3, 0.6])
        q = rng.normal(0.0, 0.4, n)
        q -= q.mean()
        sigma = np.where(has_site, rng.uniform(0.25, 0.34, n), 0.1)
        eps = np.where(has_site, rng.uniform(0.2, 0.9, n), 0.0)
        case = dict(charge=q, sigma=sigma, epsilon=eps, exc_pairs=np.zeros((0, 2), int))
        dn = near('force-switch', 0.7, 0.5)
        dd = O.desc(O.DAMPED, rc=1.0, rswitch=0.9, alpha=2.9, degree=1)
        refs = [O.pair_eval(d, pos, box, q, sigma, eps, case['exc_pairs'])[1] for d in (dn, dd)]
        world = int(rng.choice([2, 3, 5, 8]))
        total = [np.zeros((n, 3)), np.zeros((n, 3))]
        for r in range(world):
            ctx = B.HipContext(n, box, rank=r, world=world)
            fn = hip_pair(B, ctx, dn, case)
            ff = hip_pair(B, ctx, dd, case)
            ctx.pair_share_list(fn, ff)
            x = dev(pos)
            for k, fid in enumerate((fn, ff)):
                out = torch.full((n, 3), float('nan'), dtype=torch.float64, device='cuda')
                ctx.force_eval(fid, x, out, accumulate=False)
                ctx.check()
                total[k] += out.cpu().numpy()
            ctx.close()
        for k in (0, 1):
            err = np.abs(total[k] - refs[k]).max() / max(np.abs(refs[k]).max(), 1.0)
            if not err <= 1e-9:
                bad.append(('sliced', seed, world, k))
                print('sliced, seed', seed, 'world', world, 'force', k, 'error', err, flush=True)
    print('sliced evaluations: %d configurations' % args.sliced)
    print('failures:', bad)
    return 1 if bad else 0


if __name__ == '__main__':
    sys.exit(main())
