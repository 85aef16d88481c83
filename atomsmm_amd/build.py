"""Build libatomsmm_hip.so for gfx950 with hipcc (cross-compiles without a GPU).

    python -m atomsmm_amd.build [--force]

Every source is compiled to an object of its own (in parallel, and only when it or a header is newer than the object), then
linked: an edit of one kernel file costs one compilation, not eight.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
OBJ = os.path.join(CSRC, '_obj')
SOURCES = ['abi.hip', 'pair.hip', 'cluster.hip', 'group.hip', 'bonded.hip', 'integrate.hip', 'pme.hip', 'expr.hip', 'constraints.hip', 'comm.hip']
HEADERS = ['amm_ctx.h', 'pair_math.h', 'pair_tab.h', 'erfcx_table.h', 'device_utils.h', 'expr_vm.h', 'cluster.h', 'bonded_terms.h',
           os.path.join('..', '..', 'include', 'atomsmm_hip.h')]
LIB = os.path.join(HERE, 'libatomsmm_hip.so')
ARCH = 'gfx950'
FLAGS = ['--offload-arch=' + ARCH, '-O3', '-std=c++17', '-fPIC']


def _mtime(path):
    return os.path.getmtime(path) if os.path.exists(path) else 0.0


def _headers_time():
    return max(_mtime(os.path.join(CSRC, h)) for h in HEADERS)


def _object(src):
    return os.path.join(OBJ, os.path.splitext(src)[0] + '.o')


def _stale_sources(force=False):
    ht = _headers_time()
    return [s for s in SOURCES if force or _mtime(_object(s)) < max(_mtime(os.path.join(CSRC, s)), ht)]


def _stale():
    return bool(_stale_sources()) or any(_mtime(_object(s)) > _mtime(LIB) for s in SOURCES) or not os.path.exists(LIB)


def build_hip(force=False, verbose=False, jobs=None):
    """Compile every HIP source into atomsmm_amd/libatomsmm_hip.so (in-tree, so it travels with the repo)."""
    if not force and not _stale():
        return LIB
    hipcc = os.environ.get('HIPCC', 'hipcc')
    os.makedirs(OBJ, exist_ok=True)
    todo = _stale_sources(force)

    def compile_one(src):
        cmd = [hipcc] + FLAGS + ['-c', os.path.join(CSRC, src), '-o', _object(src)]
        if verbose:
            print(' '.join(cmd), flush=True)
        subprocess.check_call(cmd)

    if todo:
        with ThreadPoolExecutor(max_workers=jobs or min(len(todo), max(1, (os.cpu_count() or 2) - 1))) as pool:
            list(pool.map(compile_one, todo))
    cmd = [hipcc, '--offload-arch=' + ARCH, '-fPIC', '-shared', '-o', LIB] + [_object(s) for s in SOURCES] + ['-lhipfft', '-ldl']
    if verbose:
        print(' '.join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


if __name__ == '__main__':
    print(build_hip(force='--force' in sys.argv, verbose=True))
