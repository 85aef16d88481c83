"""Error path of the library's own RCCL communicator (csrc/comm.hip; SURVEY.md section 5, failure detection): an asynchronous
error of the communicator -- a peer rank that died -- and a collective that never completes must surface as Python exceptions
(amm_last_error) from the call that meets them, not leave the rank in hipStreamSynchronize for ever.  RCCL is bound through a table
of function pointers (dlopen), so the tests hand the library a stand-in built from tests/stubs/fake_rccl.cpp: one rank, collectives
that move nothing, errors and stalls on request.  Each scenario runs in a process of its own (a process binds ONE librccl)."""
import os
import subprocess
import sys
import textwrap

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def fake_rccl(tmp_path_factory):
    out = tmp_path_factory.mktemp('fake_rccl') / 'libfake_rccl.so'
    subprocess.check_call(['hipcc', '-shared', '-fPIC', '-O1', '-o', str(out), os.path.join(ROOT, 'tests', 'stubs', 'fake_rccl.cpp')])
    return str(out)


def _run(script, fake, env):
    full = dict(os.environ, PYTHONPATH=ROOT, FAKE_RCCL_PATH=fake, **env)
    return subprocess.run([sys.executable, '-c', textwrap.dedent(script)], capture_output=True, text=True, env=full, timeout=180)


PRELUDE = """
    import os, time
    import numpy as np, torch
    from atomsmm_amd import backend as B
    ctx = B.HipContext(96, np.array([3.0, 3.0, 3.0]))
    ctx.comm_init(B.HipContext.comm_unique_id(os.environ['FAKE_RCCL_PATH']), os.environ['FAKE_RCCL_PATH'])
    buf = torch.ones(288, dtype=torch.float64, device='cuda')
"""


def test_async_error_of_the_communicator_raises(fake_rccl, tmp_path):
    log = tmp_path / 'log.txt'
    done = _run(PRELUDE + """
    ctx.comm_allreduce(buf)
    ctx.comm_allreduce(buf)                         # the third collective is the one "a peer died" under
    try:
        ctx.comm_allreduce(buf)
        print('NO ERROR')
    except B.HipError as exc:
        print('first:', exc)
    try:
        ctx.comm_allreduce(buf)
        print('NO ERROR')
    except B.HipError as exc:
        print('second:', exc)
    try:
        ctx.check()
        print('NO ERROR')
    except B.HipError as exc:
        print('check:', exc)
    ctx.close()
    print('closed')
    """, fake_rccl, {'FAKE_RCCL_ASYNC_ERROR_AFTER': '3', 'FAKE_RCCL_LOG': str(log)})
    assert done.returncode == 0, done.stderr
    out = done.stdout
    assert 'NO ERROR' not in out and 'closed' in out
    assert 'first: ' in out and 'asynchronous error' in out and 'remote process exited' in out
    assert 'second: ' in out and 'check: ' in out and out.count('was aborted') >= 3
    assert log.read_text().split() == ['abort']              # aborted once, never destroyed afterwards


def test_stuck_collective_times_out_and_aborts(fake_rccl, tmp_path):
    log = tmp_path / 'log.txt'
    done = _run(PRELUDE + """
    ctx.set_option('comm_timeout', 1.0)
    ctx.comm_allreduce(buf)                         # parks the stream for a minute: the peer never arrives
    t0 = time.time()
    try:
        ctx.check()
        print('NO ERROR')
    except B.HipError as exc:
        print('check:', exc)
    print('waited %.1f' % (time.time() - t0))
    ctx.close()
    print('closed')
    """, fake_rccl, {'FAKE_RCCL_STALL_MS': '60000', 'FAKE_RCCL_LOG': str(log)})
    assert done.returncode == 0, done.stderr
    out = done.stdout
    assert 'NO ERROR' not in out and 'closed' in out
    assert 'did not drain within 1 s' in out and 'was aborted' in out
    waited = float([ln for ln in out.splitlines() if ln.startswith('waited')][0].split()[1])
    assert 0.9 < waited < 20.0
    assert log.read_text().split() == ['abort']


def test_destroy_is_bounded_too(fake_rccl, tmp_path):
    log = tmp_path / 'log.txt'
    done = _run(PRELUDE + """
    ctx.set_option('comm_timeout', 1.0)
    ctx.comm_allreduce(buf)
    t0 = time.time()
    try:
        ctx.comm_destroy()
        print('NO ERROR')
    except B.HipError as exc:
        print('destroy:', exc)
    print('waited %.1f' % (time.time() - t0))
    ctx.close()
    print('closed')
    """, fake_rccl, {'FAKE_RCCL_STALL_MS': '60000', 'FAKE_RCCL_LOG': str(log)})
    assert done.returncode == 0, done.stderr
    assert 'destroy: amm_comm_destroy: the stream did not drain' in done.stdout and 'closed' in done.stdout
    assert log.read_text().split() == ['abort']


def test_healthy_communicator_is_destroyed_not_aborted(fake_rccl, tmp_path):
    log = tmp_path / 'log.txt'
    done = _run(PRELUDE + """
    for _ in range(5):
        ctx.comm_allreduce(buf)
    ctx.check()
    print(ctx.comm_stats())
    ctx.close()
    print('closed')
    """, fake_rccl, {'FAKE_RCCL_LOG': str(log)})
    assert done.returncode == 0, done.stderr
    assert "'calls': 5" in done.stdout and 'closed' in done.stdout
    assert log.read_text().split() == ['destroy']
