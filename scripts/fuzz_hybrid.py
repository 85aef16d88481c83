"""Wider sweep of tests/test_gpu_hybrid.py::test_random_mixtures (development tool, GPU): python scripts/fuzz_hybrid.py [first=10] [n=40]"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import test_gpu_hybrid as T

first = int(sys.argv[1]) if len(sys.argv) > 1 else 10
count = int(sys.argv[2]) if len(sys.argv) > 2 else 40
failures = []
for seed in range(first, first + count):
    try:
        T.test_random_mixtures(seed)
    except AssertionError as exc:
        failures.append((seed, repr(exc)[:200]))
print('seeds %d..%d: failures %s' % (first, first + count - 1, failures))
