"""Host-side cost of the multi-rank step path on ONE GPU: a 1-rank RCCL group, AMM_FORCE_COLLECTIVES=1.
Prints the time to ENQUEUE n outer steps (python + ctypes + torch.distributed calls) next to the GPU time."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ['AMM_FORCE_COLLECTIVES'] = '1'
os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
os.environ.setdefault('MASTER_PORT', '29544')
import torch
import torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
import bench
sim, case = bench.build_simulation(32, (4, 2, 1), 4.0, 'damped', -1.0)
eng = sim.context._engine
bench.relax(sim, torch)
sim.step(50)
torch.cuda.synchronize()
orig = eng._run
stamps = []


def timed(ops, repeat):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    orig(ops, repeat)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    stamps.append((repeat, t1 - t0, t2 - t0, sum(1 for op in ops if isinstance(op, tuple))))


eng._run = timed
sim.step(300)
for _ in range(3):
    sim.step(20)
os.environ['AMD_LOG_LEVEL'] = '0'
for rep, enq, tot, ncoll in stamps:
    print('steps %d: enqueue %.1f us/step, total %.1f us/step, %d collectives/step' % (rep, 1e6 * enq / rep, 1e6 * tot / rep, ncoll))
dist.destroy_process_group()
