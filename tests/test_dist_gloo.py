"""CPU tests of the multi-rank path (world_size 2, gloo): the atom decomposition + all-reduce logic that the
engine uses over RCCL on the GPUs.  Pair forces of each rank's slice come from the oracle here (checker standing in
for the HIP kernels, which need a GPU); what is under test is slicing, zero-fill, the collective and lock-step."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from atomsmm_amd.engine import slice_bounds


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _worker(rank, world, port, fn, ret):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ.setdefault('GLOO_SOCKET_IFNAME', 'lo')      # no hostname resolution on the box
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        ret[rank] = fn(rank, world)
    finally:
        dist.destroy_process_group()


def run_ranks(fn, world=2):
    ctx = mp.get_context('spawn')
    with ctx.Manager() as manager:
        ret = manager.dict()
        port = _free_port()
        procs = [ctx.Process(target=_worker, args=(r, world, port, fn, ret)) for r in range(world)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(120)
        stuck = [p for p in procs if p.is_alive()]
        for p in stuck:
            p.kill()
        assert not stuck, 'a rank did not finish within 120 s'
        assert all(p.exitcode == 0 for p in procs)
        return dict(ret)


def _slice_forces_job(rank, world):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from oracle import oracle as O
    d = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'q-SPC-FW.npz'))
    n = len(d['positions'])
    desc = O.desc(O.NEAR_FSWITCH, rc=0.7, rc0=0.7, rs0=0.5)
    full = O.pair_eval(desc, d['positions'], d['box'], d['charge'], d['sigma'], d['epsilon'], d['exc_pairs'])[1]
    # a deterministic "cell-sorted" order shared by all ranks (here: sorted by z then index)
    order = np.lexsort((np.arange(n), np.floor(d['positions'][:, 2] / 0.4)))
    begin, end = slice_bounds(n, rank, world)
    mine = order[begin:end]
    buf = np.zeros((n, 3))
    buf[mine] = full[mine]                   # owner-computes: full neighbour rows for my atoms, zeros elsewhere
    t = torch.from_numpy(buf)
    dist.all_reduce(t)
    e = torch.tensor([float(rank + 1)], dtype=torch.float64)
    dist.all_reduce(e)
    return dict(n_mine=len(mine), equal=bool(np.array_equal(t.numpy(), full)), esum=e.item(),
                digest=float(np.abs(t.numpy()).sum()))


def test_slices_partition_the_atoms():
    for n, world in ((1536, 2), (98304, 8), (10, 3), (5, 8)):
        spans = [slice_bounds(n, r, world) for r in range(world)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        assert max(e - b for b, e in spans) - min(e - b for b, e in spans if e > b) <= (n + world - 1) // world


def test_allreduce_of_owner_computed_slices_is_exact():
    out = run_ranks(_slice_forces_job, world=2)
    assert out[0]['n_mine'] + out[1]['n_mine'] == 1536
    assert out[0]['equal'] and out[1]['equal']            # x + 0 == x: bit-exact on every rank
    assert out[0]['digest'] == out[1]['digest']           # ranks hold identical forces -> stay in lock-step
    assert out[0]['esum'] == out[1]['esum'] == 3.0


def _engine_job(rank, world):
    """The engine with a recording backend under a real (gloo) process group: collectives are issued in the
    same order on both ranks and only for groups that hold a pair force."""
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    sys.path.insert(0, os.path.join(root, 'tests'))
    import atomsmm_amd as atomsmm
    from atomsmm_amd import engine as E
    from atomsmm_amd import openmm, unit
    from atomsmm_amd.testing import system_from_arrays
    from fake_backend import RecordingContext
    made = []
    E._context_factory = lambda *a, **k: made.append(RecordingContext(*a, **k)) or made[-1]
    d = np.load(os.path.join(root, 'tests', 'golden', 'q-SPC-FW.npz'))
    case = {k: d[k] for k in d.files}
    system = system_from_arrays(case, nonbondedMethod='CutoffPeriodic')
    respa = atomsmm.RESPASystem(system, 0.7 * unit.nanometers, 0.5 * unit.nanometers)
    nb = atomsmm.hijackForce(respa, atomsmm.findNonbondedForce(respa))
    f = atomsmm.DampedSmoothedForce(2.9 / unit.nanometers, 1.0 * unit.nanometers, 0.9 * unit.nanometers).importFrom(nb)
    f.setForceGroup(2)
    f.addTo(respa)
    integ = atomsmm.RespaPropagator([4, 2, 1]).integrator(4 * unit.femtoseconds)
    ctx = openmm.Context(respa, integ)
    ctx.setPositions(case['positions'])
    eng = ctx._engine
    eng._buffer('f1').fill_(float(rank + 1))       # stand-in partial forces: the all-reduce must sum them
    eng._buffer('f2').fill_(float(10 * (rank + 1)))
    integ.step(2)
    rec = made[-1]
    return dict(rank=rec.rank, world=rec.world, f1=float(eng._buffers['f1'][0, 0]), f2=float(eng._buffers['f2'][0, 0]),
                n_runs=len(rec.runs), sliced=[b['sliced'] for b in rec.bonded])


def test_engine_collectives_under_gloo():
    out = run_ranks(_engine_job, world=2)
    assert (out[0]['rank'], out[0]['world']) == (0, 2) and (out[1]['rank'], out[1]['world']) == (1, 2)
    # step 1: f2 reduced twice, f1 three times; step 2 (steady state): f2 once, f1 twice.  Sum over 2 ranks each time:
    # f1: 1,2 -> x2 five times = x32 ; f2: x2 three times = x8 (values double at every all-reduce)
    assert out[0]['f1'] == out[1]['f1'] and out[0]['f2'] == out[1]['f2']
    assert out[0]['n_runs'] == out[1]['n_runs'] and out[0]['n_runs'] >= 6     # near+outer EVALs at the step boundary share a segment
    assert not any(out[0]['sliced'])


def _engine_gather_job(rank, world):
    """As _engine_job, with a recording backend that offers the exchange buffer: groups of one pair force exchange
    their slices by all-gather (host-driven here: the library hands back at each of their EVALs, the engine gathers the
    chunks over gloo and calls exchange_finish); nothing is all-reduced for them."""
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    sys.path.insert(0, os.path.join(root, 'tests'))
    import atomsmm_amd as atomsmm
    from atomsmm_amd import backend as B
    from atomsmm_amd import engine as E
    from atomsmm_amd import openmm, unit
    from atomsmm_amd.testing import system_from_arrays
    from fake_backend import RecordingContext

    class GatherContext(RecordingContext):
        def __init__(self, *a, **k):
            super().__init__(*a, **k)
            self.exchange, self.gather_groups, self.finishes, self.pending = None, set(), 0, False

        def bind_exchange(self, tensor):
            self.exchange = tensor

        def group_set_exchange(self, group, mode):
            assert mode == B.EXCHANGE_GATHER
            self.gather_groups.add(group)

        def run_ops_host_exchanges(self, ops, repeat, exchange):
            # (the library runs up to an exchanged EVAL, hands back, the engine gathers the chunks and calls exchange_finish, the
            # library goes on: include/atomsmm_hip.h, amm_run_ops_from)
            super().run_ops(ops, repeat)
            per = (self.n + self.world - 1) // self.world
            for _ in range(repeat):
                for o in ops:
                    if o.op == B.OP_EVAL and o.a in self.gather_groups:
                        assert not self.pending, 'an exchanged EVAL although the previous exchange was not finished'
                        # the pair kernel's part: this rank's chunk of the exchange buffer
                        self.exchange[self.rank * per * 3:(self.rank + 1) * per * 3] = float(self.rank + 1)
                        self.pending = True
                        exchange(1)

        def exchange_finish(self):
            assert self.pending
            self.pending = False
            self.finishes += 1

    made = []
    E._context_factory = lambda *a, **k: made.append(GatherContext(*a, **k)) or made[-1]
    d = np.load(os.path.join(root, 'tests', 'golden', 'q-SPC-FW.npz'))
    case = {k: d[k] for k in d.files}
    system = system_from_arrays(case, nonbondedMethod='CutoffPeriodic')
    respa = atomsmm.RESPASystem(system, 0.7 * unit.nanometers, 0.5 * unit.nanometers)
    nb = atomsmm.hijackForce(respa, atomsmm.findNonbondedForce(respa))
    f = atomsmm.DampedSmoothedForce(2.9 / unit.nanometers, 1.0 * unit.nanometers, 0.9 * unit.nanometers).importFrom(nb)
    f.setForceGroup(2)
    f.addTo(respa)
    integ = atomsmm.RespaPropagator([4, 2, 1]).integrator(4 * unit.femtoseconds)
    ctx = openmm.Context(respa, integ)
    ctx.setPositions(case['positions'])
    eng = ctx._engine
    eng._buffer('f1').fill_(float(rank + 1))
    integ.step(2)
    rec = made[-1]
    per = (rec.n + world - 1) // world
    chunks = [float(rec.exchange[r * per * 3]) for r in range(world)]
    return dict(gather=sorted(rec.gather_groups), finishes=rec.finishes, chunks=chunks, f1=float(eng._buffers['f1'][0, 0]),
                pending=rec.pending, n_runs=len(rec.runs))


def test_engine_all_gather_exchange_under_gloo():
    out = run_ranks(_engine_gather_job, world=2)
    for r in (0, 1):
        assert out[r]['gather'] == [1, 2]                  # near and outer force: one pair force each
        assert out[r]['finishes'] == 5 + 3                 # f1: 3 + 2 evaluations, f2: 2 + 1 (two steps)
        assert out[r]['chunks'] == [1.0, 2.0]              # every rank holds every rank's chunk
        assert out[r]['f1'] == float(r + 1)                # no all-reduce touched the group buffers
        assert not out[r]['pending']
    assert out[0]['n_runs'] == out[1]['n_runs']


def _consistency_job(rank, world):
    """Unseeded setVelocitiesToTemperature and a rank-local failed check under a real process group."""
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    sys.path.insert(0, os.path.join(root, 'tests'))
    import atomsmm_amd as atomsmm
    from atomsmm_amd import backend as B
    from atomsmm_amd import engine as E
    from atomsmm_amd import openmm, unit
    from atomsmm_amd.testing import system_from_arrays
    from fake_backend import RecordingContext

    class FailingOnRank1(RecordingContext):
        fail = False

        def check(self):
            if self.fail and self.rank == 1:
                raise B.HipError('neighbour row overflow (rank-local)')

    made = []
    E._context_factory = lambda *a, **k: made.append(FailingOnRank1(*a, **k)) or made[-1]
    d = np.load(os.path.join(root, 'tests', 'golden', 'q-SPC-FW.npz'))
    case = {k: d[k] for k in d.files}
    system = system_from_arrays(case, nonbondedMethod='CutoffPeriodic')
    respa = atomsmm.RESPASystem(system, 0.7 * unit.nanometers, 0.5 * unit.nanometers)
    integ = atomsmm.RespaPropagator([2, 1, 1]).integrator(1 * unit.femtoseconds)
    ctx = openmm.Context(respa, integ)
    ctx.setPositions(case['positions'])
    ctx.setVelocitiesToTemperature(300 * unit.kelvin)             # no seed: every process would draw its own
    v = ctx._engine.v.numpy().copy()
    gathered = [torch.zeros_like(torch.from_numpy(v)) for _ in range(world)]
    dist.all_gather(gathered, torch.from_numpy(v))
    same = all(bool(torch.equal(g, gathered[0])) for g in gathered)
    integ.step(1)                                                 # healthy step: no rank raises
    made[-1].fail = True
    raised = None
    try:
        integ.step(1)
    except B.HipError as exc:
        raised = str(exc)
    return dict(same=same, nonzero=bool(np.abs(v).max() > 0), raised=raised)


def test_unseeded_velocities_and_failed_checks_are_collective():
    out = run_ranks(_consistency_job, world=2)
    assert out[0]['same'] and out[1]['same'] and out[0]['nonzero']
    assert 'rank-local' in out[1]['raised']                       # the rank that detected it
    assert 'another rank' in out[0]['raised']                     # its peer raises too instead of hanging in the next collective


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus N` without torchrun (how the driver starts the 1-GPU run, and what a user types): the script must
    start N ranks itself, before anything touches a GPU, and relay rank 0's single JSON line.  AMM_BENCH_DRYRUN=1 replaces the GPU
    work by a gloo all-reduce of the rank numbers (there is no GPU here)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, AMM_BENCH_DRYRUN='1')
    env.pop('WORLD_SIZE', None)
    env.pop('RANK', None)
    out = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '3', '--steps', '1', '--warmup', '0'],
                         env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1
    assert json.loads(lines[0]) == {'dryrun': True, 'n_gpus': 3, 'rank_sum': 6.0}


def test_bench_launcher_ends_the_run_when_a_rank_dies():
    """A rank k > 0 that dies during start-up (import error, no device, failed rendezvous) must not leave rank 0 waiting in the
    rendezvous until somebody's time limit: the launcher polls all children, terminates the others and exits non-zero with the
    failing rank's stderr."""
    import subprocess
    import sys
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, AMM_BENCH_DRYRUN='1', AMM_BENCH_DRYRUN_FAIL_RANK='2')
    env.pop('WORLD_SIZE', None)
    env.pop('RANK', None)
    t0 = time.monotonic()
    out = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '3', '--steps', '1', '--warmup', '0'],
                         env=env, capture_output=True, text=True, timeout=240)
    assert out.returncode != 0
    assert 'rank 2 failed' in out.stderr and 'simulated start-up failure' in out.stderr
    assert not [ln for ln in out.stdout.splitlines() if ln.lstrip().startswith('{')]
    assert time.monotonic() - t0 < 120          # (the gloo rendezvous alone would wait for its 30-minute default)


def test_slices_are_whole_molecules_and_cover_every_atom():
    """The rule both list kinds slice by (backend.slice_per == amm_slice_per of csrc/amm_ctx.h): a rank owns `per` consecutive
    slots of the cell-sorted order, per a multiple of three (molecule rows slice by molecule), world * per >= n."""
    from atomsmm_amd import backend as B
    for n in (3, 24, 1536, 4233, 98304, 249075):
        for world in (1, 2, 3, 4, 5, 8, 16):
            per = B.slice_per(n, world)
            assert per % 3 == 0 and world * per >= n
            assert (world - 1) * per < n + 3 * world            # no more than a molecule of slack per rank
            assert per == 3 * -(-(-(-n // 3)) // world)


def test_local_world_runs_ranks_as_threads():
    """engine.LocalWorld: W ranks as threads of one process (what the 8-rank GPU test and scripts/per_rank_step.py run on) -- the
    collectives' semantics on host tensors, an engine that finds the world it runs in, and a rank that raises does not leave the others
    waiting."""
    import sys
    import numpy as np
    import torch
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
    import atomsmm_amd as atomsmm
    from atomsmm_amd import engine as E
    from atomsmm_amd import openmm, unit
    from atomsmm_amd.testing import system_from_arrays
    from fake_backend import RecordingContext

    world = E.LocalWorld(4)

    def collectives(rank):
        buf = torch.full((4 * 3,), -1.0, dtype=torch.float64)
        buf[rank * 3:(rank + 1) * 3] = float(rank + 1)
        world.all_gather(rank, buf, 3)
        total = torch.tensor([float(rank + 1)], dtype=torch.float64)
        world.all_reduce(rank, total)
        flag = torch.tensor([1 if rank == 2 else 0], dtype=torch.int32)
        world.all_reduce(rank, flag, op='max')
        return buf.tolist(), float(total), int(flag), world.broadcast(rank, 'from rank %d' % rank)

    out = world.run(collectives)
    for r in range(4):
        assert out[r][0] == [1.0] * 3 + [2.0] * 3 + [3.0] * 3 + [4.0] * 3
        assert out[r][1:] == (10.0, 1, 'from rank 0')

    made = []
    saved = E._context_factory
    E._context_factory = lambda *a, **k: made.append(RecordingContext(*a, **k)) or made[-1]
    try:
        d = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'q-SPC-FW.npz'))
        case = {k: d[k] for k in d.files}

        def job(rank):
            system = system_from_arrays(case, nonbondedMethod='CutoffPeriodic')
            respa = atomsmm.RESPASystem(system, 0.7 * unit.nanometers, 0.5 * unit.nanometers)
            integ = atomsmm.RespaPropagator([2, 2, 1]).integrator(2 * unit.femtoseconds)
            ctx = openmm.Context(respa, integ)
            ctx.setPositions(case['positions'])
            integ.step(2)
            eng = ctx._engine
            return eng.rank, eng.world, eng._local is not None

        assert E.LocalWorld(2).run(job) == [(0, 2, True), (1, 2, True)]
        assert sorted((c.rank, c.world) for c in made) == [(0, 2), (1, 2)]
    finally:
        E._context_factory = saved

    def failing(rank):
        if rank == 1:
            raise ValueError('rank 1 fails')
        world.all_reduce(rank, torch.zeros(1))

    with pytest.raises(ValueError, match='rank 1 fails'):
        world.run(failing)
