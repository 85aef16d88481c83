"""Development probe: C3 water box through the raw C-ABI, per-phase timings (not the bench)."""
import sys, time
sys.path.insert(0, '.')
import numpy as np, torch
from atomsmm_amd import backend as B
from atomsmm_amd.testing import tip3p_box

nside = int(sys.argv[1]) if len(sys.argv) > 1 else 32
c = tip3p_box(nside)
n = len(c['positions'])
print('atoms', n, 'box', c['box'][0])
ctx = B.HipContext(n, c['box'])
dev = lambda a: torch.as_tensor(np.ascontiguousarray(a, dtype=np.float64), device='cuda')
bid = ctx.bonded_create()
ctx.bonded_add_terms(bid, B.BOND_HARMONIC, c['bonds'], np.stack([c['bond_r0'], c['bond_k']], 1))
ctx.bonded_add_terms(bid, B.ANGLE_HARMONIC, c['angles'], np.stack([c['angle_theta0'], c['angle_k']], 1))
ctx.bonded_finalize(bid)
nid = ctx.pair_create(B.pair_desc(B.NEAR_FSWITCH, 0.7, rc0=0.7, rs0=0.5), c['charge'], c['sigma'], c['epsilon'], c['exc_pairs'])
did = ctx.pair_create(B.pair_desc(B.DAMPED, 1.0, rswitch=0.9, alpha=2.9, degree=1), c['charge'], c['sigma'], c['epsilon'], c['exc_pairs'])
x, v, m = dev(c['positions']), dev(c['velocities']), dev(c['mass'])
bufs = [torch.zeros((n, 3), dtype=torch.float64, device='cuda') for _ in range(4)]
ctx.bind_state(x, v, m)
for k, b in enumerate(bufs): ctx.bind_buffer(k, b)
ctx.group_define(0, 0, [bid]); ctx.group_define(1, 1, [nid]); ctx.group_define(2, 2, [did])

def timeit(fn, reps=20):
    fn(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / reps * 1e6

t0 = time.perf_counter(); ctx.force_eval(nid, x, bufs[1]); ctx.synchronize(); print('near first build+eval %.1f ms' % ((time.perf_counter()-t0)*1e3))
t0 = time.perf_counter(); ctx.force_eval(did, x, bufs[2]); ctx.synchronize(); print('far first build+eval %.1f ms' % ((time.perf_counter()-t0)*1e3))
print('near stats', ctx.pair_stats(nid)); print('far stats', ctx.pair_stats(did))
print('near eval (list reuse) us', timeit(lambda: ctx.force_eval(nid, x, bufs[1])))
print('far  eval (list reuse) us', timeit(lambda: ctx.force_eval(did, x, bufs[2])))
print('bonded eval us', timeit(lambda: ctx.force_eval(bid, x, bufs[0])))
print('kick us', timeit(lambda: ctx.kick(v, bufs[0], m, 0.0)))
print('move us', timeit(lambda: ctx.move(x, v, 0.0)))
ctx.profile_enable(True)
for _ in range(20): ctx.force_eval(nid, x, bufs[1]); ctx.force_eval(did, x, bufs[2])
nl, ms = ctx.profile_read(nid); print('near pair kernel only: %.1f us' % (ms/nl*1e3))
nl, ms = ctx.profile_read(did); print('far  pair kernel only: %.1f us' % (ms/nl*1e3))
ctx.profile_enable(False)
dt = 0.004
E, K, M, CP = B.OP_EVAL, B.OP_KICK, B.OP_MOVE, B.OP_COPY
ops = [B.Op(E, 2, 0, 0, 0.0), B.Op(E, 1, 0, 0, 0.0), B.Op(CP, 3, 2, 0, 0.0), B.Op(K, 3, 1, 0, 0.5 * dt)]
for _n1 in range(2):
    ops += [B.Op(K, 1, -1, 0, 0.25 * dt), B.Op(E, 0, 0, 0, 0.0)]
    for _n0 in range(4):
        ops += [B.Op(K, 0, -1, 0, 0.0625 * dt), B.Op(M, 0, 0, 0, 0.125 * dt), B.Op(E, 0, 0, 0, 0.0), B.Op(K, 0, -1, 0, 0.0625 * dt)]
    ops += [B.Op(E, 1, 0, 0, 0.0), B.Op(K, 1, -1, 0, 0.25 * dt)]
ops += [B.Op(E, 2, 0, 0, 0.0), B.Op(CP, 3, 2, 0, 0.0), B.Op(K, 3, 1, 0, 0.5 * dt)]
for nsteps in (20, 100, 100):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ctx.run_ops(ops, repeat=nsteps); ctx.check()
    t = (time.perf_counter() - t0) / nsteps
    print('outer step %.1f us -> %.1f ns/day' % (t * 1e6, dt * 1e-3 * 86400 / t), ctx.pair_stats(nid)['n_builds'], ctx.pair_stats(did)['n_builds'])
out = torch.zeros(1, dtype=torch.float64, device='cuda'); ctx.mvv(v, m, out)
print('T = %.1f K' % (out.item() / (3 * n) / 0.0083144626))
