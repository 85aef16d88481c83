"""GPU test of the real multi-rank path on ONE MI355X: two processes share cuda:0 and talk over gloo (RCCL needs
one GPU per rank; the driver's 8-GPU run uses backend nccl).  Each rank evaluates the pair forces of its slice of
the cell-sorted order with the HIP kernels into its chunk of the exchange buffer, the engine all-gathers the chunks
(groups of one pair force) or all-reduces the group buffers (groups with further sliced terms), every rank integrates
all atoms: after 3 RESPA steps both ranks hold the SAME bits as a single-rank run."""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip('torch')


# Order of summation.  Water-like systems walk MOLECULE rows on the force-only path (csrc/cluster.hip): a row's partial sums depend
# on the row alone, so any decomposition gives the same bits -- the product default is what these tests run.  The per-atom rows
# (csrc/pair.hip: systems that do not qualify, and option "cluster" = 0) cut a wavefront's rows into stretches with / without
# Lennard-Jones arithmetic at positions taken from all the rows of the wavefront; their bit-identity variants switch that off
# with the context option "site_trips" (amm_set_option: an ABI option, not an environment variable).


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _simulate(nsteps, pme=False, state=1, chunks=1):
    import atomsmm_amd as atomsmm
    from atomsmm_amd import openmm, unit
    from atomsmm_amd.openmm import app
    from atomsmm_amd.testing import system_from_arrays, tip3p_box
    c = tip3p_box(10)          # 3000 atoms, L = 3.1 nm
    system = system_from_arrays(c, nonbondedMethod='PME' if pme else 'CutoffPeriodic')
    respa = atomsmm.RESPASystem(system, 0.7 * unit.nanometers, 0.5 * unit.nanometers)
    if not pme:      # SURVEY 8d C3 composition; with pme the group-2 force stays the PME NonbondedForce
        nb = atomsmm.hijackForce(respa, atomsmm.findNonbondedForce(respa))
        outer = atomsmm.DampedSmoothedForce(2.9 / unit.nanometers, 1.0 * unit.nanometers, 0.9 * unit.nanometers).importFrom(nb)
        outer.setForceGroup(2)
        outer.addTo(respa)
    integrator = atomsmm.RespaPropagator([4, 2, 1]).integrator(2 * unit.femtoseconds)
    # equal Verlet buffers on both sides of the comparison: the default grows with the number of ranks, and another buffer means
    # other (masked) entries between the same pairs in a row, i.e. another order of the partial sums
    sim = app.Simulation(app.Topology(), respa, integrator, openmm.Platform.getPlatformByName('HIP'),
                         {'Skin': '0.1', 'Option.state_exchange': str(state)})
    sim.context.setPositions(c['positions'] * unit.nanometers)
    sim.context.setVelocities(c['velocities'])
    e0 = sim.context.getState(getEnergy=True).getPotentialEnergy()._value
    eng = sim.context._engine
    before = eng.ctx.comm_stats() if eng._native_comm else None
    for _ in range(chunks):
        sim.step(nsteps)
    after = eng.ctx.comm_stats() if eng._native_comm else None
    run_stats = eng.ctx.run_stats()
    st = sim.context.getState(getPositions=True, getVelocities=True, getEnergy=True, getForces=True, groups={0, 1, 2})
    stats = eng.ctx.pair_stats(eng.pair_force_ids(2)[0])
    native = eng._native_comm
    world = eng.world
    eng.ctx.close()      # deterministic teardown: stream drained, ncclCommDestroy of the library's communicator, context freed -- now
    return dict(x=st.getPositions(asNumpy=True)._value, v=st.getVelocities(asNumpy=True)._value,
                f=st.getForces(asNumpy=True)._value, e=st.getPotentialEnergy()._value, e0=e0,
                slice_atoms=stats['n_slice_atoms'], world=world, native_comm=native, run_stats=run_stats, builds=stats['n_builds'],
                comm=None if before is None else {k: after[k] - before[k] for k in after})


def _worker(rank, world, port, ret, pme=False, state=1, nsteps=3, chunks=1):
    import faulthandler
    import sys
    faulthandler.dump_traceback_later(100, exit=True, file=sys.stderr)      # a stall names its stack instead of being 'did not finish'
    import torch.distributed as dist
    os.environ['RANK'] = str(rank)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ.setdefault('GLOO_SOCKET_IFNAME', 'lo')      # no hostname resolution on the box
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        ret[rank] = _simulate(nsteps, pme, state, chunks)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('pme', [False, True])
def test_two_ranks_match_single_rank_bit_for_bit(pme):
    """pme=True: the group-2 buffer also carries the sliced exclusion and reciprocal-space terms, whose owner rank
    (atom-index block) can differ from the owner of the atom's pair row (cell-sorted slice): the all-reduce then adds
    p + (b + m) where one rank adds (p + b) + m -- agreement to rounding instead of bit for bit."""
    import torch.multiprocessing as mp
    single = _simulate(3, pme)
    assert single['world'] == 1
    ctx = mp.get_context('spawn')
    with ctx.Manager() as manager:
        ret = manager.dict()
        port = _free_port()
        procs = [ctx.Process(target=_worker, args=(r, 2, port, ret, pme)) for r in range(2)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(120)
        stuck = [p for p in procs if p.is_alive()]
        for p in stuck:          # never leave a rank behind on the GPU box
            p.kill()
        assert not stuck, 'a rank did not finish within 120 s'
        assert all(p.exitcode == 0 for p in procs)
        out = dict(ret)
    for r in (0, 1):
        assert out[r]['world'] == 2 and out[r]['slice_atoms'] == 1500
        if pme:
            assert np.abs(out[r]['x'] - single['x']).max() < 1e-13
            assert np.abs(out[r]['v'] - single['v']).max() < 1e-11
            assert np.abs(out[r]['f'] - single['f']).max() < 1e-9 * np.abs(single['f']).max()
            assert np.array_equal(out[r]['x'], out[0]['x']) and np.array_equal(out[r]['v'], out[0]['v'])
        else:
            assert np.array_equal(out[r]['x'], single['x'])
            assert np.array_equal(out[r]['v'], single['v'])
            assert np.array_equal(out[r]['f'], single['f'])
        assert out[r]['e0'] == pytest.approx(single['e0'], rel=1e-13)
        assert out[r]['e'] == pytest.approx(single['e'], rel=1e-13)


@pytest.mark.parametrize('world,state', [(2, 1), (2, 0), (4, 1)])
def test_ranks_that_integrate_their_own_molecules_match_single_rank(world, state):
    """Owner-integrates: with several ranks the launch that walks a molecule's rows also runs its inner RESPA
    loop (csrc/cluster.hip: cepi_rows on the rank's slice) and the ranks all-gather positions and velocities -- not forces -- which
    k_state_scatter spreads to the atom-order arrays together with the next evaluation's sorted copies; the redundant inner loop over
    all atoms, the unsort and the gather are gone.  15 + 15 RESPA steps in two calls (rebuilds on the way, the calls' first and last
    evaluations take the force exchange) on 2 and 4 ranks sharing the card over gloo: the single rank's positions, velocities and
    forces bit for bit; option state_exchange = 0 is the force exchange of rounds 2-4, with the same bits."""
    import torch.multiprocessing as mp
    single = _simulate(15, chunks=2)
    assert single['world'] == 1 and single['run_stats']['state_exchanges'] == 0 and single['run_stats']['epilogues'] > 0
    ctx = mp.get_context('spawn')
    with ctx.Manager() as manager:
        ret = manager.dict()
        port = _free_port()
        procs = [ctx.Process(target=_worker, args=(r, world, port, ret, False, state, 15, 2)) for r in range(world)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(150)
        stuck = [p for p in procs if p.is_alive()]
        for p in stuck:          # never leave a rank behind on the GPU box
            p.kill()
        assert not stuck, 'a rank did not finish within 150 s'
        assert all(p.exitcode == 0 for p in procs)
        out = dict(ret)
    for r in range(world):
        assert out[r]['world'] == world and out[r]['builds'] == single['builds'] and single['builds'] >= 2
        # every evaluation of a call but its last step boundary: 2 x (15 x 2 - 1), one less when the engine runs the first step alone
        assert out[r]['run_stats']['state_exchanges'] in ((57, 58) if state else (0,)), out[r]['run_stats']
        assert np.array_equal(out[r]['x'], single['x'])
        assert np.array_equal(out[r]['v'], single['v'])
        assert np.array_equal(out[r]['f'], single['f'])


def test_eight_ranks_as_threads_match_single_rank():
    """World 8 on one card: the ranks are threads of this process (atomsmm_amd.engine.LocalWorld -- the real slices, chunks and launches;
    the all-gathers are device-to-device copies).  Owner-integrates with the state exchange, 12 + 12 RESPA steps: every rank ends with
    the single rank's positions, velocities and forces, bit for bit."""
    from atomsmm_amd.engine import LocalWorld
    single = _simulate(12, chunks=2)
    out = LocalWorld(8).run(lambda rank: _simulate(12, chunks=2))
    for r in range(8):
        assert out[r]['world'] == 8 and out[r]['run_stats']['state_exchanges'] >= 45
        assert out[r]['builds'] == single['builds']
        assert np.array_equal(out[r]['x'], single['x'])
        assert np.array_equal(out[r]['v'], single['v'])
        assert np.array_equal(out[r]['f'], single['f'])


def _phase(name):
    """One line per phase of a spawned rank on its stderr (the parent shows it when the rank fails or stalls)."""
    import sys
    import time
    print('[rank %s] %.1f s: %s' % (os.environ.get('RANK', '0'), time.monotonic() - _T0, name), file=sys.stderr, flush=True)


_T0 = __import__('time').monotonic()


def _rccl_worker(port, ret, pme):
    """Phases are announced and a watchdog dumps every thread's stack (and ends the process) after 100 s: a stall names its phase
    instead of being found as 'did not finish' (round 3: one such stall, cause discussed in DESIGN.md section 4).  The
    bootstrap sockets of c10d and RCCL are pinned to the loopback interface: the box has no name resolution, and an unpinned
    bootstrap walks the interfaces with reverse look-ups that can wait for resolver time-outs."""
    import faulthandler
    import sys
    faulthandler.dump_traceback_later(100, exit=True, file=sys.stderr)
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    os.environ.setdefault('NCCL_SOCKET_IFNAME', 'lo')
    os.environ.setdefault('GLOO_SOCKET_IFNAME', 'lo')
    os.environ['AMM_FORCE_COLLECTIVES'] = '1'       # a 1-rank job takes the multi-rank code path
    torch.cuda.set_device(0)
    _phase('rendezvous + torch NCCL group')
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
    try:
        _phase('context, library communicator, 3 steps')
        ret[0] = _simulate(3, pme)
        _phase('teardown: library communicator first (its Context is closed), then the process group')
    finally:
        dist.destroy_process_group()
    _phase('done')
    faulthandler.cancel_dump_traceback_later()


@pytest.mark.parametrize('pme', [False, True])
def test_library_owned_rccl_communicator(pme):
    """The step program with AMM_OP_ALLREDUCE ops on the library's own RCCL communicator (csrc/comm.hip), over a 1-rank
    group (RCCL refuses two ranks on one GPU): ncclCommInitRank from a broadcast id, ncclAllReduce on the context's
    stream inside amm_run_ops, energies through amm_comm_allreduce -- same bits as the plain single-process run."""
    import torch.multiprocessing as mp
    single = _simulate(3, pme)
    assert not single['native_comm']
    ctx = mp.get_context('spawn')
    with ctx.Manager() as manager:
        ret = manager.dict()
        p = ctx.Process(target=_rccl_worker, args=(_free_port(), ret, pme))
        p.start()
        p.join(120)          # a healthy run: process start + imports ~15 s, RCCL init ~5 s, the steps < 1 s
        if p.is_alive():
            p.kill()
            raise AssertionError('the RCCL rank did not finish within 120 s (its watchdog should have named the phase at 100 s)')
        assert p.exitcode == 0
        out = dict(ret)[0]
    assert out['native_comm'] and out['slice_atoms'] == 3000
    for key in ('x', 'v', 'f'):
        assert np.array_equal(out[key], single[key]), key
    assert out['e'] == single['e'] and out['e0'] == single['e0']
    # [4,2,1]: the outer force once and the near force twice per step = 3 x 3N doubles; the first step also evaluates
    # both at its start.  Groups of one pair force exchange their slices by all-gather, and the two evaluated in one
    # pass at the step boundary share a message; with PME the group-2 buffer also carries sliced exclusion and
    # reciprocal-space terms and is all-reduced on its own
    assert out['comm']['doubles'] == (3 * 3 + 2) * 9000, out['comm']
    assert out['comm']['calls'] == (2 * 3 + 1 if not pme else 3 * 3 + 2), out['comm']


@pytest.mark.parametrize('cluster', [1, 0])
def test_exchange_chunks_of_five_uneven_slices(cluster):
    """The all-gather exchange without any collective library: five contexts in ONE process stand for ranks 0..4 of a
    world of 5 (1536 atoms = 512 molecules: slices of 309, 309, 309, 309, 300 slots -- whole molecules).  cluster = 1: molecule
    rows (the product default for water); 0: per-atom rows with the stretch cutting off.  Every 'rank' evaluates the near and the outer
    force of its slice in one pass into its chunk of its exchange buffer (no communicator: the EVAL leaves the exchange
    to the host), the chunks are copied between the buffers as an all-gather would, amm_exchange_finish spreads them:
    every rank then holds the forces of a single-context evaluation, bit for bit."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from atomsmm_amd import backend as B
    from atomsmm_amd.testing import tip3p_box
    from test_gpu_abi_parity import near, hip_pair, dev, O
    c = tip3p_box(8)
    n = len(c['positions'])
    assert n == 1536
    dn = near('force-switch', 0.7, 0.5)
    dd = O.desc(O.DAMPED, rc=1.0, rswitch=0.9, alpha=2.9, degree=1)
    world = 5
    per = B.slice_per(n, world)
    assert per == 309
    E = B.OP_EVAL

    def make(rank, w):
        ctx = B.HipContext(n, c['box'], rank=rank, world=w)
        ctx.set_option('cluster', cluster)
        ctx.set_option('site_trips', 0)
        assert ctx.exchange_per() == (per if w == world else B.slice_per(n, w))
        fn = hip_pair(B, ctx, dn, c, skin=0.1)         # equal buffers on both sides: the default grows with the world
        ff = hip_pair(B, ctx, dd, c, skin=0.1)
        ctx.pair_share_list(fn, ff)
        x, v, m = dev(c['positions']), dev(c['velocities']), dev(c['mass'])
        f = [torch.zeros((n, 3), dtype=torch.float64, device='cuda') for _ in range(3)]
        ctx.bind_state(x, v, m)
        for slot, buf in enumerate(f):
            ctx.bind_buffer(slot, buf)
        ctx.group_define(1, 1, [fn])
        ctx.group_define(2, 2, [ff])
        return ctx, f, (x, v, m)

    ref, fref, keep_ref = make(0, 1)
    ref.run_ops([B.Op(E, 1, 0, 0, 0.0), B.Op(E, 2, 0, 0, 0.0)], 1)
    ref.check()
    ranks = []
    for r in range(world):
        ctx, f, keep = make(r, world)
        xchg = torch.full((world * 2 * per * 3,), float('nan'), dtype=torch.float64, device='cuda')
        ctx.bind_exchange(xchg)
        ctx.group_set_exchange(1, B.EXCHANGE_GATHER)
        ctx.group_set_exchange(2, B.EXCHANGE_GATHER)
        ctx.run_ops([B.Op(E, 1, 0, 0, 0.0), B.Op(E, 2, 0, 0, 0.0)], 1)       # dual pass: chunk = [2][per][3]
        ranks.append((ctx, f, xchg, keep))
    chunk = 2 * per * 3
    for r, (ctx, f, xchg, keep) in enumerate(ranks):           # the all-gather, by hand
        for q, other in enumerate(ranks):
            if q != r:
                xchg[q * chunk:(q + 1) * chunk].copy_(other[2][q * chunk:(q + 1) * chunk])
    for ctx, f, xchg, keep in ranks:
        ctx.exchange_finish()
        ctx.check()
        assert torch.equal(f[1], fref[1]) and torch.equal(f[2], fref[2])
        assert ctx.pair_stats(1)['list_kind'] == cluster
        with pytest.raises(B.HipError):
            ctx.exchange_finish()                               # nothing is waiting any more
    # a single (not dual) exchanged evaluation uses chunks of [per][3]
    ctx, f, xchg, keep = ranks[2]
    f[1].zero_()
    ctx.run_ops([B.Op(E, 1, 0, 0, 0.0)], 1)
    lo = 2 * per * 3
    mine = xchg[lo:lo + per * 3].clone()
    assert torch.isfinite(mine).all() and torch.count_nonzero(mine) > 0
    with pytest.raises(B.HipError):                             # the previous exchange still waits for its finish
        ctx.run_ops([B.Op(E, 1, 0, 0, 0.0)], 1)
    for ctx, f, xchg, keep in ranks:
        ctx.close()
    ref.close()


def test_exchange_with_empty_slices_on_a_tiny_box():
    """More ranks than work: 24 atoms = 8 molecules over a world of 16: per = 3 slots (one molecule), ranks 8..15 own nothing.
    Small box: the minimum-image (RINT) list build and the per-pair periodic image of the traversal.  Same hand-made
    all-gather as above; every rank, the empty ones included, ends with the single-context force, bit for bit."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from atomsmm_amd import backend as B
    from atomsmm_amd.testing import tip3p_box
    from test_gpu_abi_parity import near, hip_pair, dev
    c = tip3p_box(2)
    n = len(c['positions'])
    assert n == 24
    dn = near('force-switch', 0.3, 0.25)
    world = 16
    per = B.slice_per(n, world)
    assert per == 3 and (world - 1) * per >= n            # the last ranks are empty
    E = B.OP_EVAL

    def make(rank, w):
        ctx = B.HipContext(n, c['box'], rank=rank, world=w)
        fn = hip_pair(B, ctx, dn, c, skin=0.1)         # equal buffers on both sides: the default grows with the world
        x, v, m = dev(c['positions']), dev(c['velocities']), dev(c['mass'])
        f = torch.zeros((n, 3), dtype=torch.float64, device='cuda')
        ctx.bind_state(x, v, m)
        ctx.bind_buffer(1, f)
        ctx.group_define(1, 1, [fn])
        return ctx, f, (x, v, m)

    ref, fref, keep_ref = make(0, 1)
    ref.run_ops([B.Op(E, 1, 0, 0, 0.0)], 1)
    ref.check()
    assert float(fref.abs().max()) > 0.0
    ranks = []
    for r in range(world):
        ctx, f, keep = make(r, world)
        xchg = torch.zeros(world * 2 * per * 3, dtype=torch.float64, device='cuda')
        ctx.bind_exchange(xchg)
        ctx.group_set_exchange(1, B.EXCHANGE_GATHER)
        ctx.run_ops([B.Op(E, 1, 0, 0, 0.0)], 1)
        ranks.append((ctx, f, xchg, keep))
    chunk = per * 3                                       # a single evaluation: chunks of [per][3]
    for r, (ctx, f, xchg, keep) in enumerate(ranks):
        for q, other in enumerate(ranks):
            if q != r:
                xchg[q * chunk:(q + 1) * chunk].copy_(other[2][q * chunk:(q + 1) * chunk])
    for ctx, f, xchg, keep in ranks:
        ctx.exchange_finish()
        ctx.check()
        assert torch.equal(f, fref)
        ctx.close()
    ref.close()


@pytest.mark.parametrize('cluster', [1, 0])
def test_c3_size_eight_slices_all_gather_bit_identical(cluster):
    """Config C4 of BASELINE.json at full size, emulated on one GPU: eight contexts stand for the ranks of a world of 8 over
    the 98 304-atom TIP3P box (slices of 4 096 molecules of the cell-sorted order).  Each 'rank' runs the dual pass (near + outer
    force) on its slice into its chunk of the exchange buffer, the chunks are copied as an all-gather would,
    amm_exchange_finish unsorts them: every rank then holds the single-context forces bit for bit -- before and after a
    displacement that makes every rank rebuild its part of the list.  cluster = 1: molecule rows, the product default (a row's
    order of summation depends on the row alone).  cluster = 0: per-atom rows; there a slice is walked with 16 lanes per atom
    and the whole box with 8, and the wavefront-wide stretch cutting couples a row's order to its wave-mates, so the reference
    context is pinned to 16 lanes and the cutting is off on both sides (options of the ABI); against the product's own 8-lane
    walk with the cutting on the forces agree to 1e-12 of the largest, checked too."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from atomsmm_amd import backend as B
    from atomsmm_amd.testing import tip3p_box
    from test_gpu_abi_parity import near, hip_pair, dev, O
    c = tip3p_box(32)
    n = len(c['positions'])
    assert n == 98304
    dn = near('force-switch', 0.7, 0.5)
    dd = O.desc(O.DAMPED, rc=1.0, rswitch=0.9, alpha=2.9, degree=1)
    world = 8
    per = B.slice_per(n, world)
    E = B.OP_EVAL
    dual = [B.Op(E, 1, 0, 0, 0.0), B.Op(E, 2, 0, 0, 0.0)]

    def make(rank, w, options):
        ctx = B.HipContext(n, c['box'], rank=rank, world=w)
        for name, value in options.items():
            ctx.set_option(name, value)
        fn = hip_pair(B, ctx, dn, c, skin=0.1)         # equal buffers on both sides: the default grows with the world
        ff = hip_pair(B, ctx, dd, c, skin=0.1)
        ctx.pair_share_list(fn, ff)
        x, v, m = dev(c['positions']), dev(c['velocities']), dev(c['mass'])
        f = [torch.zeros((n, 3), dtype=torch.float64, device='cuda') for _ in range(3)]
        ctx.bind_state(x, v, m)
        for slot, buf in enumerate(f):
            ctx.bind_buffer(slot, buf)
        ctx.group_define(1, 1, [fn])
        ctx.group_define(2, 2, [ff])
        return ctx, f, x, (v, m), ff

    # (the number of lanes that share a row fixes the order of its partial sums: the library picks it from the number of rows a
    # context owns -- 8 for the whole box, 32 for an eighth -- so both sides of a bit-for-bit comparison pin it)
    pinned = {'cluster': 1, 'lanes_per_row': 8} if cluster else {'cluster': 0, 'lanes_per_row': 16, 'site_trips': 0}
    ref, fref, xref, keep_ref, ff_ref = make(0, 1, pinned)
    ref8, fref8, xref8, keep_ref8, ff_ref8 = make(0, 1, {'cluster': 0})      # the per-atom walk as shipped: 8 lanes, stretches cut
    ref8.run_ops(dual, 1)
    ref8.check()
    assert ref8.pair_stats(ff_ref8)['lanes_per_atom'] == 8 and ref8.pair_stats(ff_ref8)['list_kind'] == 0
    ranks = []
    for r in range(world):
        ctx, f, x, keep, ff = make(r, world, pinned)
        xchg = torch.full((world * 2 * per * 3,), float('nan'), dtype=torch.float64, device='cuda')
        ctx.bind_exchange(xchg)
        ctx.group_set_exchange(1, B.EXCHANGE_GATHER)
        ctx.group_set_exchange(2, B.EXCHANGE_GATHER)
        ranks.append((ctx, f, xchg, x, keep, ff))
    chunk = 2 * per * 3
    rng = np.random.default_rng(3)
    # whole molecules move (beyond skin / 2 for many of them: rebuild) and every atom a little on top
    shift = torch.as_tensor(np.repeat(rng.normal(0.0, 0.04, (n // 3, 3)), 3, axis=0) + rng.normal(0.0, 0.004, (n, 3)), device='cuda')
    for stage in range(2):
        ref.run_ops(dual, 1)
        ref.check()
        for ctx, f, xchg, x, keep, ff in ranks:
            ctx.run_ops(dual, 1)
        for r, (ctx, f, xchg, x, keep, ff) in enumerate(ranks):                # the all-gather, by hand
            for q, other in enumerate(ranks):
                if q != r:
                    xchg[q * chunk:(q + 1) * chunk].copy_(other[2][q * chunk:(q + 1) * chunk])
        slices = 0
        for ctx, f, xchg, x, keep, ff in ranks:
            ctx.exchange_finish()
            ctx.check()
            assert torch.equal(f[1], fref[1]) and torch.equal(f[2], fref[2])
            st = ctx.pair_stats(ff)
            assert st['list_kind'] == cluster and st['lanes_per_atom'] == (8 if cluster else 16) and st['n_builds'] == stage + 1
            slices += st['n_slice_atoms']
        assert slices == n and ref.pair_stats(ff_ref)['n_builds'] == stage + 1
        if stage == 0:
            for k in (1, 2):
                assert (fref[k] - fref8[k]).abs().max() <= 1e-12 * fref8[k].abs().max()
            ref8.close()
        xref.add_(shift)
        for ctx, f, xchg, x, keep, ff in ranks:
            x.add_(shift)
    for ctx, f, xchg, x, keep, ff in ranks:
        ctx.close()
    ref.close()


def _simulate_c5(nsteps):
    """The 4 233-atom instance of config C5 (solvated chain with 1-4 exceptions + softcore solute + AFED variable): per-group
    forces, deriv(energy, lambda_vdw), then AFED steps through the host-walked program (interaction-group force, term-parallel
    / sliced bond lists, the derivative's reduction over ranks, deferred globals)."""
    import atomsmm_amd as atomsmm
    from atomsmm_amd import openmm, unit
    from atomsmm_amd.testing import build_c5_system, solvated_chain
    case = solvated_chain(nside=12, n_chain=300, n_solute=30)
    respa = build_c5_system(case)
    inner = atomsmm.RespaPropagator([2, 2, 1]).integrator(1 * unit.femtoseconds)
    var = atomsmm.ExtendedSystemVariable('lambda_vdw', 50, 2.5, 20 * unit.femtoseconds)
    integrator = atomsmm.AdiabaticDynamicsIntegrator(inner, 2, [var])
    context = openmm.Context(respa, integrator, openmm.Platform.getPlatformByName('HIP'), {'Skin': '0.1'})
    context.setPositions(case['positions'] * unit.nanometers)
    context.setVelocities(case['velocities'])
    context.setParameter('lambda_vdw', 0.8)
    eng = context._engine
    out = dict(world=eng.world, n=len(case['positions']))
    for g in (0, 1, 2):
        st = context.getState(getForces=True, getEnergy=True, groups={g})
        out['e%d' % g] = st.getPotentialEnergy()._value
        out['f%d' % g] = st.getForces(asNumpy=True)._value
        out['fo%d' % g] = context.getState(getForces=True, groups={g}).getForces(asNumpy=True)._value     # force-only kernels
    out['dEdl'] = eng.energy_derivative('lambda_vdw')
    integrator.step(0)
    integrator.setGlobalVariableByName('_v_lambda_vdw', 0.05)
    integrator.setGlobalVariableByName('_v_eta_lambda_vdw', 0.0)
    integrator.step(nsteps)
    st = context.getState(getPositions=True, getVelocities=True)
    out['x'] = st.getPositions(asNumpy=True)._value
    out['v'] = st.getVelocities(asNumpy=True)._value
    out['lam'] = context.getParameter('lambda_vdw')
    out['v_lam'] = integrator.getGlobalVariableByName('_v_lambda_vdw')
    return out


def _worker_c5(rank, world, port, ret):
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ.setdefault('GLOO_SOCKET_IFNAME', 'lo')
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        ret[rank] = _simulate_c5(2)
    finally:
        dist.destroy_process_group()


def test_c5_four_ranks_match_single_rank():
    """Config C5's path on several ranks: four processes share the card over gloo, each evaluates the
    pair rows / bond-list terms of its slice; group energies, forces (energy-carrying and force-only kernels) and
    deriv(energy, lambda_vdw) after the reductions equal the single-rank values to rounding (the all-reduce adds the
    ranks' partial sums in another order than one rank does), and two AFED steps end in the same state on every rank."""
    import torch.multiprocessing as mp
    single = _simulate_c5(2)
    assert single['world'] == 1 and single['n'] == 4233
    world = 4
    ctx = mp.get_context('spawn')
    with ctx.Manager() as manager:
        ret = manager.dict()
        port = _free_port()
        procs = [ctx.Process(target=_worker_c5, args=(r, world, port, ret)) for r in range(world)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(240)
        stuck = [p for p in procs if p.is_alive()]
        for p in stuck:
            p.kill()
        assert not stuck, 'a rank did not finish within 240 s'
        assert all(p.exitcode == 0 for p in procs)
        out = dict(ret)
    for r in range(world):
        o = out[r]
        assert o['world'] == world
        for g in (0, 1, 2):
            scale = np.abs(single['f%d' % g]).max()
            assert o['e%d' % g] == pytest.approx(single['e%d' % g], rel=1e-11)
            assert np.abs(o['f%d' % g] - single['f%d' % g]).max() <= 1e-11 * scale
            assert np.abs(o['fo%d' % g] - single['fo%d' % g]).max() <= 1e-11 * scale
        assert o['dEdl'] == pytest.approx(single['dEdl'], rel=1e-10)
        assert np.abs(o['x'] - single['x']).max() < 1e-11
        assert np.abs(o['v'] - single['v']).max() < 1e-9
        assert o['lam'] == pytest.approx(single['lam'], abs=1e-12)
        assert o['v_lam'] == pytest.approx(single['v_lam'], rel=1e-9, abs=1e-12)
        # every rank holds the same state, bit for bit (no rank-dependent arithmetic after the reductions)
        assert np.array_equal(o['x'], out[0]['x']) and np.array_equal(o['v'], out[0]['v'])
