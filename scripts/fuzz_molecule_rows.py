"""Wider sweep of the randomised molecule-row checks of tests/test_gpu_abi_parity.py (development tool, GPU):
    python scripts/fuzz_molecule_rows.py [first_seed=10] [n=60]
Runs the body of test_random_molecule_boxes_vs_oracle for seeds beyond the nine the test suite pins."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import test_gpu_abi_parity as T

first = int(sys.argv[1]) if len(sys.argv) > 1 else 10
count = int(sys.argv[2]) if len(sys.argv) > 2 else 60
failures = []
for seed in range(first, first + count):
    try:
        T.random_molecule_case(seed, seed)
    except AssertionError as exc:
        failures.append((seed, repr(exc)[:200]))
print('seeds %d..%d: failures %s' % (first, first + count - 1, failures))
