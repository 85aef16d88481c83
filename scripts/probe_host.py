"""Host enqueue time vs GPU time of the steady-state RESPA op list (run on the GPU box)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

sim, case = bench.build_simulation(32, (4, 2, 1), 4.0)
eng = sim.context._engine
bench.relax(sim, torch)
sim.step(20)
key = eng._program_key(eng._valid, eng._mirror)
ops = [op for op in eng._programs[key][0] if not isinstance(op, tuple)]
print('ops per step', len(ops))
for rep in (100, 300):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng.ctx.run_ops(ops, rep)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print('repeat %d: enqueue %.3f ms/step, total %.3f ms/step' % (rep, (t1 - t0) / rep * 1e3, (t2 - t0) / rep * 1e3))
