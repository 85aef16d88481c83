#!/usr/bin/env python3
"""Where the host time of an AFED step at config C5 goes (cProfile over --steps steps after a warm-up)."""
import argparse
import cProfile
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--top', type=int, default=45)
    args = ap.parse_args()
    import torch
    import bench
    sim, case = bench.build_simulation_c5((4, 2, 1), 2.0)
    sim.step(5)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    sim.step(args.steps)
    torch.cuda.synchronize()
    print('wall %.2f ms / AFED step' % (1e3 * (time.perf_counter() - t0) / args.steps))
    pr = cProfile.Profile()
    pr.enable()
    sim.step(args.steps)
    torch.cuda.synchronize()
    pr.disable()
    st = pstats.Stats(pr)
    st.sort_stats('cumulative').print_stats(args.top)
    st.sort_stats('tottime').print_stats(25)


if __name__ == '__main__':
    main()
