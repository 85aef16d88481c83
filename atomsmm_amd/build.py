"""Build libatomsmm_hip.so for gfx950 with hipcc (cross-compiles without a GPU).

    python -m atomsmm_amd.build [--force]
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
SOURCES = ['abi.hip', 'pair.hip', 'bonded.hip', 'integrate.hip', 'pme.hip', 'expr.hip', 'constraints.hip', 'comm.hip']
HEADERS = ['amm_ctx.h', 'pair_math.h', 'erfcx_table.h', 'device_utils.h', 'expr_vm.h', os.path.join('..', '..', 'include', 'atomsmm_hip.h')]
LIB = os.path.join(HERE, 'libatomsmm_hip.so')
ARCH = 'gfx950'


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in SOURCES + HEADERS)


def build_hip(force=False, verbose=False):
    """Compile every HIP source into atomsmm_amd/libatomsmm_hip.so (in-tree, so it travels with the repo)."""
    if not force and not _stale():
        return LIB
    hipcc = os.environ.get('HIPCC', 'hipcc')
    cmd = [hipcc, '--offload-arch=' + ARCH, '-O3', '-std=c++17', '-fPIC', '-shared', '-o', LIB] + \
          [os.path.join(CSRC, s) for s in SOURCES] + ['-lhipfft', '-ldl']
    if verbose:
        print(' '.join(cmd))
    subprocess.check_call(cmd)
    return LIB


if __name__ == '__main__':
    print(build_hip(force='--force' in sys.argv, verbose=True))
