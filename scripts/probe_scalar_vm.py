"""Cost of the scalar interpreter (csrc/expr.hip: k_expr_scalar, amm_expr_eval_scalar): launches of programs of 10 ... 600 words, back to back.
Measured (round 5, MI355X): ~8 us for the shortest launch, 0.2 us per word whatever the words are -- a lone wavefront issues an instruction
every ~5 cycles and a word costs ~80 of them (fetch, dispatch, v_readlane pops, select-on-lane pushes)."""
import sys; sys.path.insert(0, '/root/repo')
import numpy as np, torch
from atomsmm_amd import backend as B, expr as X
ctx = B.HipContext(64, np.array([3.0, 3.0, 3.0]))
G = torch.zeros(2048, dtype=torch.float64, device='cuda')
O = X.OPCODES
def prog(kind, n):
    code, consts = [], [1.5, 0.25]
    if kind == 'addmul':
        code = [O['CONST'] | (0 << 8)]
        for k in range(n):
            code += [O['CONST'] | (1 << 8), O['MUL'], O['CONST'] | (0 << 8), O['ADD']]
        code += [O['OUT'] | (5 << 8)]
    elif kind == 'exp':
        code = [O['CONST'] | (1 << 8)]
        for k in range(n):
            code += [O['exp'], O['CONST'] | (1 << 8), O['MUL']]
        code += [O['OUT'] | (5 << 8)]
    elif kind == 'devg':
        code = []
        for k in range(n):
            code += [O['DEVG'] | (7 << 8), O['OUT'] | (8 << 8)]
    return code, consts
for kind, n in (('addmul', 5), ('addmul', 40), ('addmul', 150), ('exp', 5), ('exp', 40), ('devg', 5), ('devg', 40)):
    code, consts = prog(kind, n)
    for rep in range(3):
        ctx.expr_eval_scalar(code, consts, G)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ctx.synchronize()
    import time
    t0 = time.perf_counter()
    for rep in range(200):
        ctx.expr_eval_scalar(code, consts, G)
    ctx.synchronize()
    dt = (time.perf_counter() - t0) / 200 * 1e6
    print('%-7s n=%4d words=%4d  %.1f us per launch (back to back, wall)  -> %.3f us per word' % (kind, n, len(code), dt, dt / len(code)))
