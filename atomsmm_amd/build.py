"""Build libatomsmm_hip.so for gfx950 with hipcc (cross-compiles without a GPU).

    python -m atomsmm_amd.build [--force]

Every source is compiled to an object of its own (in parallel, and only when the content of the source, of a header or the flags
differ from what the object was built from), then linked: an edit of one kernel file costs one compilation, not ten.
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


def _digest(paths, extra=''):
    import hashlib
    h = hashlib.sha256(extra.encode())
    for path in paths:
        h.update(b'\0' + os.path.basename(path).encode() + b'\0')
        with open(path, 'rb') as fh:
            h.update(fh.read())
    return h.hexdigest()


def _object(src):
    return os.path.join(OBJ, os.path.splitext(src)[0] + '.o')


def _stamp(src):
    return _object(src) + '.stamp'


def _want(src):
    """What an up-to-date object of `src` was built from: the source, every header and the flags -- by CONTENT.  (Modification times
    do not survive a copy to another machine, and a tuning build with other flags left an object newer than its source that the
    old mtime test kept: ADVICE r3.)"""
    return _digest([os.path.join(CSRC, src)] + [os.path.join(CSRC, h) for h in HEADERS], ' '.join(FLAGS))


def _read(path):
    try:
        with open(path) as fh:
            return fh.read().strip()
    except OSError:
        return ''


def _stale_sources(force=False):
    return [s for s in SOURCES if force or not os.path.exists(_object(s)) or _read(_stamp(s)) != _want(s)]


def _lib_want():
    return _digest([], ' '.join(_read(_stamp(s)) for s in SOURCES))


def _stale():
    return bool(_stale_sources()) or not os.path.exists(LIB) or _read(LIB + '.stamp') != _lib_want()


def build_hip(force=False, verbose=False, jobs=None):
    """Compile every HIP source into atomsmm_amd/libatomsmm_hip.so (in-tree, so it travels with the repo).  Up to date = the stamps
    next to the objects and the library match the content of the sources, headers and flags."""
    if not force and not _stale():
        return LIB
    hipcc = os.environ.get('HIPCC', 'hipcc')
    os.makedirs(OBJ, exist_ok=True)
    todo = _stale_sources(force)

    def compile_one(src):
        cmd = [hipcc] + FLAGS + ['-c', os.path.join(CSRC, src), '-o', _object(src)]
        if verbose:
            print(' '.join(cmd), flush=True)
        want = _want(src)
        subprocess.check_call(cmd)
        with open(_stamp(src), 'w') as fh:
            fh.write(want)

    if todo:
        with ThreadPoolExecutor(max_workers=jobs or min(len(todo), max(1, (os.cpu_count() or 2) - 1))) as pool:
            list(pool.map(compile_one, todo))
    cmd = [hipcc, '--offload-arch=' + ARCH, '-fPIC', '-shared', '-o', LIB] + [_object(s) for s in SOURCES] + ['-lhipfft', '-ldl']
    if verbose:
        print(' '.join(cmd), flush=True)
    subprocess.check_call(cmd)
    with open(LIB + '.stamp', 'w') as fh:
        fh.write(_lib_want())
    return LIB


if __name__ == '__main__':
    print(build_hip(force='--force' in sys.argv, verbose=True))
