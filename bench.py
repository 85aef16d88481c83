#!/usr/bin/env python3
"""bench.py -- headline benchmark of the hot path (BASELINE.json): ns/day of a ~100k-atom flexible TIP3P box
under RESPA (near/far split + multiple-timescale inner loop) on N MI355X, plus the near-nonbonded
kernel's roofline figure and a CPU baseline timed in the same run.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload: 32 768 flexible TIP3P waters (98 304 atoms, L = 9.94 nm) built by
atomsmm_amd.testing.tip3p_box (seeded, synthetic); RESPASystem(rcutIn = 0.7 nm, rswitchIn = 0.5 nm,
'force-switch'); outer force DampedSmoothedForce(alpha = 2.9/nm, 1.0 nm, 0.9 nm) in group 2;
RespaPropagator([4,2,1]), outer step 4 fs (0.5 / 2 / 4 fs).  Everything goes through the AtomsMM-shaped API
(atomsmm_amd as atomsmm) and therefore through the C-ABI of libatomsmm_hip.so.  One "step" = one outer
RESPA step = 1 far + 2 near + 8 bonded evaluations, 22 kicks, 8 moves.  The lattice start is relaxed
before timing (velocity rescaling to 300 K, untimed) so that neighbour-list rebuild frequency is that of
liquid water.  Inputs are resident in HBM when the timed region starts.

N > 1: one process per GPU; every rank integrates all atoms, evaluates the pair-force rows of its slice of the
cell-sorted order and the slices are all-gathered (RCCL, the library's own communicator; groups with sliced bond
lists or reciprocal space all-reduce their buffer instead).  Strong scaling (fixed box).

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s peak (spec)
FP64_VECTOR_PEAK_TF = 78.6     # public datasheet value; not in the container's guide
BYTES_PER_ATOM = 72            # fp64: read x,y,z + q,sigma,eps, write Fx,Fy,Fz
BYTES_PER_ATOM_DUAL = 96       # dual pass: the same read set, two force arrays written
EPILOGUE_BYTES_PER_ATOM = 200  # inner RESPA loop inside the launch: x, v, f0 read + written (144), mass (8), next sorted copy (48)
FLOP_PER_PAIR_NEAR = 60        # force-switch near, force only
FLOP_PER_PAIR_FAR = 80         # DampedSmoothedForce (erfc + exp), force only
FP64_SUSTAINED_TF = 60.7       # measured: v_fma_f64, 8 wavefronts per SIMD, 2.16 ns per wave-instruction per SIMD (DVFS clock)
TRAFFIC_FILE = 'r05_traffic.json'
KB = 0.0083144626181532


EXTRA_OPTIONS = []          # --option name=value: context options of the library (amm_set_option), tuning only


def build_simulation(nside, loops, dt_fs, outer_kind='damped', skin=None, outer_skin=None):
    import atomsmm_amd as atomsmm
    from atomsmm_amd import openmm, unit
    from atomsmm_amd.openmm import app
    from atomsmm_amd.testing import system_from_arrays, tip3p_box
    case = tip3p_box(nside)
    if outer_kind == 'pme':      # the group-2 force stays the source PME NonbondedForce
        system = system_from_arrays(case, nonbondedMethod='PME', cutoff=1.0, switch=0.9)
        respa = atomsmm.RESPASystem(system, 0.7 * unit.nanometers, 0.5 * unit.nanometers)
    else:                        # DampedSmoothedForce outer force
        system = system_from_arrays(case, nonbondedMethod='CutoffPeriodic', cutoff=1.0, switch=0.9)
        respa = atomsmm.RESPASystem(system, 0.7 * unit.nanometers, 0.5 * unit.nanometers)
        nb = atomsmm.hijackForce(respa, atomsmm.findNonbondedForce(respa))
        outer = atomsmm.DampedSmoothedForce(2.9 / unit.nanometers, 1.0 * unit.nanometers, 0.9 * unit.nanometers)
        outer.importFrom(nb)
        outer.setForceGroup(2)
        outer.addTo(respa)
    integrator = atomsmm.RespaPropagator(list(loops)).integrator(dt_fs * unit.femtoseconds)
    simulation = app.Simulation(app.Topology(len(case['positions'])), respa, integrator,
                                openmm.Platform.getPlatformByName('HIP'),
                                dict(([('Skin', str(skin))] if skin is not None else []) +
                                     ([('OuterSkin', str(outer_skin))] if outer_skin is not None else []) +
                                     [('Option.' + k, v) for k, v in EXTRA_OPTIONS]) or None)
    simulation.context.setPositions(case['positions'] * unit.nanometers)
    simulation.context.setVelocities(case['velocities'])
    return simulation, case


def build_simulation_c5(loops, dt_fs, afed_substeps=2, skin=None):
    """Config C5: ~249 000 atoms -- 82 015 flexible TIP3P waters, a 3 000-atom bonded chain with 1-4 exceptions
    (NonbondedExceptionsForce in group 0) and a 30-atom solute coupled by the AFED extended variable lambda_vdw;
    AdiabaticDynamicsIntegrator(RespaPropagator(loops).integrator(dt), n) -> one step covers 2 n dt."""
    import atomsmm_amd as atomsmm
    from atomsmm_amd import openmm, unit
    from atomsmm_amd.openmm import app
    from atomsmm_amd.testing import build_c5_system, solvated_chain
    case = solvated_chain()
    respa = build_c5_system(case)
    inner = atomsmm.RespaPropagator(list(loops)).integrator(dt_fs * unit.femtoseconds)
    variable = atomsmm.ExtendedSystemVariable('lambda_vdw', 50, 2.5, 20 * unit.femtoseconds)
    integrator = atomsmm.AdiabaticDynamicsIntegrator(inner, afed_substeps, [variable])
    integrator.setRandomNumberSeed(7)
    simulation = app.Simulation(app.Topology(len(case['positions'])), respa, integrator, openmm.Platform.getPlatformByName('HIP'),
                                {'Skin': str(skin)} if skin is not None else None)
    simulation.context.setPositions(case['positions'] * unit.nanometers)
    simulation.context.setVelocities(case['velocities'])
    simulation.context.setParameter('lambda_vdw', 0.6)
    return simulation, case


C2_SKIN_NM = 0.2


def bench_c2(args, torch):
    """Config C2 of BASELINE.json: 32 768-atom Lennard-Jones fluid (atomsmm_amd.testing.lj_fluid, rho sigma^3 = 0.8), the ONLY force a
    NearNonbondedForce(2.5 sigma, 0.9 x that, 'force-switch') imported from the NonbondedForce (forces.py:655-670), fp64, one GPU;
    velocity Verlet at 4 fs through the AtomsMM-shaped API.  A step = one force evaluation + the two half kicks and the move.  The
    atoms are not molecules: the traversal is the per-atom-row kernel k_pair_tab<NEAR_FSWITCH> (radial table, analytic
    Lennard-Jones).  Roofline as for C3: algorithmic bytes 72 N per launch against the HBM peak, and the fp64 fraction."""
    import atomsmm_amd as atomsmm
    from atomsmm_amd import backend, openmm, unit
    from atomsmm_amd.openmm import app
    from atomsmm_amd.testing import lj_fluid, system_from_arrays
    case = lj_fluid(32)
    n = len(case['positions'])
    sigma = float(case['sigma'][0])
    rc, rs, dt_fs = 2.5 * sigma, 0.9 * 2.5 * sigma, 4.0
    rng = np.random.default_rng(7)
    case['velocities'] = rng.normal(size=(n, 3)) * np.sqrt(KB * 100.0 / case['mass'])[:, None]
    system = system_from_arrays(case, nonbondedMethod='CutoffPeriodic', cutoff=rc)
    nb = atomsmm.hijackForce(system, atomsmm.findNonbondedForce(system))
    near = atomsmm.NearNonbondedForce(rc * unit.nanometers, rs * unit.nanometers, 'force-switch').importFrom(nb)
    near.addTo(system)
    integrator = atomsmm.UnconstrainedVelocityVerletPropagator().integrator(dt_fs * unit.femtoseconds)
    # Verlet buffer: the slow atoms of a 100 K Lennard-Jones fluid want a wider one than the library's default of 0.1 nm (made for
    # water's hydrogens at 4 fs): 0.2 nm -- 40 instead of 16 steps between list builds -- measured on the scan 0.06 ... 0.30 nm
    # (profiles/r05_c2_skin_scan.txt: 6690 / 7490 at 0.1 / 8110 at 0.18-0.26 / 7890 ns/day); `--skin` overrides
    skin = args.skin if args.skin is not None else C2_SKIN_NM
    simulation = app.Simulation(app.Topology(n), system, integrator, openmm.Platform.getPlatformByName('HIP'), {'Skin': str(skin)})
    simulation.context.setPositions(case['positions'] * unit.nanometers)
    simulation.context.setVelocities(case['velocities'])
    eng = simulation.context._engine
    relaxed = 0 if args.no_relax else relax(simulation, torch, target=100.0)
    simulation.step(args.warmup)
    fid = eng.pair_force_ids(0)[0]
    st0 = eng.ctx.pair_stats(fid)
    eng.ctx.profile_enable(True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    simulation.step(args.steps)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    launches, ms = eng.ctx.profile_read(fid)
    eng.ctx.profile_enable(False)
    st1 = eng.ctx.pair_stats(fid)
    pairs = eng.ctx.pair_count_within(fid, eng.x, rc) / 2
    t_k = ms * 1e-3 / max(launches, 1)
    alg = BYTES_PER_ATOM * n
    achieved = alg / max(t_k, 1e-12) / 1e9
    tf = FLOP_PER_PAIR_NEAR * pairs / max(t_k, 1e-12) / 1e12
    stored = stored_traffic(backend)
    result = {
        # (NOT the headline metric of BASELINE.json -- that is the default run's, config C3; this is its config C2)
        'metric': 'ns/day on the 32k-atom Lennard-Jones fluid of BASELINE config C2 (NearNonbondedForce only); near-nonbonded HBM GB/s vs 8 TB/s peak',
        'value': round(dt_fs * 1e-6 * 86400.0 / (elapsed / args.steps), 3), 'unit': 'ns/day', 'n_gpus': 1, 'steps': args.steps,
        'warmup': args.warmup, 'ms_per_step': round(elapsed / args.steps * 1e3, 4), 'higher_is_better': True, 'scaling': 'strong',
        'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
        'config': {'workload': 'C2: %d-atom Lennard-Jones fluid (L = %.3f nm, rho sigma^3 = 0.8), NearNonbondedForce(%.3f, %.3f, force-switch) '
                               'only, velocity Verlet at %.0f fs; Verlet buffer %.2f nm' % (n, case['box'][0], rc, rs, dt_fs, skin),
                   'atoms': n, 'step_fs': dt_fs, 'verlet_buffer_nm': skin, 'relax_steps': relaxed, 'parallelism': 'single GPU',
                   'temperature_K_end': round(temperature(eng, torch), 1)},
        'roofline': {'bound': 'hbm', 'achieved': round(achieved, 3), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                     'frac': round(achieved / HBM_PEAK_GBS, 6), 'traffic': stored.get('c2_near', {}).get('hbm_bytes_per_launch'),
                     'traffic_raw': (int((stored['c2_near']['fetch_size_kb_avg'] + stored['c2_near']['write_size_kb_avg']) * 1024)
                                     if 'c2_near' in stored else None),
                     'kernel': 'k_pair_tab<NEAR_FSWITCH%s> (per-atom rows, force only)' % (', no Coulomb table: all charges zero' if st1.get('chargeless') else ''),
                     'avg_launch_us': round(t_k * 1e6, 2),
                     'launches': launches, 'algorithmic_bytes_per_launch': alg, 'fp64_tflops': round(tf, 3),
                     'fp64_frac_of_vector_peak': round(tf / FP64_VECTOR_PEAK_TF, 4),
                     'note': 'FP64-VALU / latency bound, not HBM bound (SURVEY.md 8d)'},
        'detail': {'list_builds_in_timed_region': st1['n_builds'] - st0['n_builds'], 'list_pairs': st1['n_list_pairs'],
                   'pairs_within_cutoff_counted': int(pairs), 'lanes_per_atom': st1['lanes_per_atom'], 'rlist_nm': st1['rlist'],
                   'list_kind': st1['list_kind'], 'kernel_revision': backend.kernel_revision()},
    }
    if not args.no_cpu_baseline:
        # CPU leg: the oracle's OpenMP cell-list traversal of the same force at the final configuration (one evaluation = one step's
        # force work; kicks and moves are negligible next to it), all host threads OpenMP gives it
        try:
            # (the box shows all of the host's logical CPUs under a cgroup quota: as many OpenMP threads as the quota grants)
            from oracle import cpu_port
            quota = cpu_port.cpu_quota()
            threads = max(1, int(min(os.cpu_count() or 1, quota if quota else (os.cpu_count() or 1))))
            os.environ['OMP_NUM_THREADS'] = str(threads)
            from oracle import oracle as O
            d = O.desc(O.ADJ['force-switch'], rc=rc, rc0=rc, rs0=rs)
            x = eng.x.cpu().numpy()
            O.pair_eval(d, x, case['box'], case['charge'], case['sigma'], case['epsilon'], None, use_cells=True)
            t0 = time.perf_counter()
            reps = 5
            for _ in range(reps):
                O.pair_eval(d, x, case['box'], case['charge'], case['sigma'], case['epsilon'], None, use_cells=True)
            sec = (time.perf_counter() - t0) / reps
            result['cpu_baseline'] = {'value': round(dt_fs * 1e-6 * 86400.0 / sec, 4), 'unit': 'ns/day', 'cores': threads, 'kind': 'port',
                                      'sample': '%d force evaluations of the same %d-atom configuration by the oracle (oracle/amm_oracle.c, OpenMP '
                                                'cell traversal, every pair twice), %.1f ms each' % (reps, n, sec * 1e3)}
        except Exception as exc:
            result['cpu_baseline'] = {'value': None, 'unit': 'ns/day', 'cores': 0, 'kind': 'port', 'sample': 'failed: %r' % (exc,)}
    print(json.dumps(result), flush=True)


def stored_traffic(backend):
    """HBM bytes per launch from the PMC passes of scripts/measure_round.sh (rocprofv3 cannot run inside this process), kept in
    profiles/ and quoted only while the kernels are the ones they were measured on."""
    path = os.path.join(ROOT, 'profiles', TRAFFIC_FILE)
    try:
        stored = json.load(open(path))
        return stored if stored.get('kernel_revision') == backend.kernel_revision() else {}
    except Exception:
        return {}


def run_leg(config, steps, warmup, timeout_s=280):
    """Another BASELINE config (c2 / c5) as a short timed region of its own, in a child process (`bench.py --config ...`: a fresh
    interpreter and context; the parent's GPU work is over and it only waits), condensed for the default line's `detail`: ns/day,
    ms/step, the dominant kernel's roofline entry, the CPU leg where the config has one.  A failure costs the leg, not the line."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), '--config', config, '--steps', str(steps), '--warmup', str(warmup), '--no-legs']
    try:
        out = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout_s)
    except subprocess.TimeoutExpired:
        return {'failed': 'no result after %d s' % timeout_s}
    lines = [ln for ln in out.stdout.splitlines() if ln.lstrip().startswith('{')]
    if out.returncode != 0 or not lines:
        return {'failed': 'exit code %s: %s' % (out.returncode, out.stderr.strip()[-300:])}
    line = json.loads(lines[-1])
    leg = {'ns_day': line['value'], 'ms_per_step': line['ms_per_step'], 'steps': line['steps'], 'warmup': line['warmup'],
           'workload': line['config']['workload'], 'roofline': line.get('roofline')}
    if line.get('roofline_dominant'):
        leg['roofline_dominant'] = line['roofline_dominant']
    if line.get('cpu_baseline'):
        leg['cpu_baseline'] = line['cpu_baseline']
    for key in ('list_kind', 'rows', 'kernel_revision', 'step_boundary_pass_us', 'near_kernel_us'):
        if key in line.get('detail', {}):
            leg[key] = line['detail'][key]
    return leg


def temperature(engine, torch):
    out = torch.zeros(1, dtype=torch.float64, device=engine.x.device)
    engine.ctx.mvv(engine.v, engine.mass, out)
    return out.item() / (3 * engine.n * KB)


def relax(simulation, torch, target=300.0, max_steps=1500, block=10, log=None):
    """Untimed: take the synthetic lattice start to liquid-like water at ~300 K by rescaling velocities."""
    eng = simulation.context._engine
    calm = 0
    done = 0
    while done < max_steps:
        simulation.step(block)
        done += block
        T = temperature(eng, torch)
        if log:
            log('relax step %d: T = %.1f K' % (done, T))
        eng.v.mul_((target / T) ** 0.5)
        calm = calm + 1 if T < 1.04 * target else 0
        if calm >= 5:
            break
    return done


def cpu_baseline(nside, loops, dt_fs, sample_steps=40, state=None):
    """The same system and step program on this host's cores: oracle/cpu_port.c -- the whole RESPA step loop in C with OpenMP over
    rows / molecules / atoms, one cell-sorted Verlet list (rebuilt on displacement) shared by the near and the outer force, every
    pair evaluated once (Newton's third law, per-thread force copies), one force cache per group (the outer force and the last near
    force of a step in one traversal), first-touch allocation; the same arithmetic as oracle/amm_oracle.c
    (tests/test_oracle_golden.py).  OpenMM is not installable here: this is a port, and labelled so.  `state` = (positions,
    velocities) at the end of the GPU run: the sample continues the relaxed liquid the GPU was timed on, not the lattice start.
    Thread count and Verlet buffer are calibrated on the host (a few timed steps each): the box's CPU quota, not its logical CPU
    count, decides the first; a CPU list build is dear next to its pair loop, so its best buffer is larger than the GPU's."""
    from atomsmm_amd.testing import tip3p_box
    from oracle import cpu_port
    case = tip3p_box(nside)
    if state is not None:
        case = dict(case, positions=state[0], velocities=state[1])
    kw = dict(loops=tuple(loops), dt=dt_fs * 1e-3)
    # every candidate over 12 steps: two or more rebuild intervals of its Verlet buffer (five steps were fewer than one)
    threads, timing = cpu_port.best_thread_count(case, steps=12, skin=0.2, **kw)
    skins = {}
    for skin in (0.1, 0.2, 0.3, 0.4):
        skins[skin] = cpu_port.time_port(case, warmup=1, steps=12, skin=skin, **kw)[0]
    best_skin = min(skins, key=skins.get)
    sec, st = cpu_port.time_port(case, warmup=3, steps=sample_steps, skin=best_skin, **kw)
    quota = cpu_port.cpu_quota()
    return {'value': round(dt_fs * 1e-6 * 86400.0 / sec, 4), 'unit': 'ns/day', 'cores': threads, 'kind': 'port',
            'threads': threads, 'cgroup_cpu_quota': None if quota is None else round(quota, 2), 'logical_cpus': os.cpu_count(),
            'thread_scan_ms_per_step': {str(t): round(v * 1e3, 1) for t, v in sorted(timing.items())},
            'skin_scan_ms_per_step': {'%.1f' % k: round(v * 1e3, 1) for k, v in sorted(skins.items())},
            'sample': '%d outer RESPA steps (after 3 warm-up) of the same %d-atom workload continued from the state the GPU run ended in, '
                      '%.1f ms/step, %d list builds, Verlet buffer %.1f nm; CPU port in C + OpenMP (oracle/cpu_port.c: step loop, shared '
                      'cell-sorted Verlet list, each pair once), not OpenMM' % (sample_steps, len(case['positions']), sec * 1e3, st['builds'], best_skin)}


def cpu_baseline_in_child(nside, x, v, timeout_s=600):
    """cpu_baseline() in a child process of its own (fresh interpreter, never touches the GPU): the baseline is a reported figure
    and must never cost the GPU result -- an exception, a crash of the C port (an illegal instruction on another host CPU cannot
    be caught in-process) or a hang all leave the bench line intact with the reason in `sample`."""
    import subprocess
    import tempfile
    failed = {'value': None, 'unit': 'ns/day', 'cores': 0, 'kind': 'port'}
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, 'state.npz')
        np.savez(path, x=x, v=v)
        try:
            out = subprocess.run([sys.executable, os.path.abspath(__file__), '--cpu-baseline-child', path, '--nside', str(nside)],
                                 capture_output=True, text=True, timeout=timeout_s)
        except subprocess.TimeoutExpired:
            return dict(failed, sample='failed: no result after %d s' % timeout_s)
    lines = [ln for ln in out.stdout.splitlines() if ln.lstrip().startswith('{')]
    if out.returncode != 0 or not lines:
        return dict(failed, sample='failed: child exit code %s: %s' % (out.returncode, out.stderr.strip()[-300:]))
    return json.loads(lines[-1])


def launch_ranks(n, poll_s=0.2, timeout_s=None):
    """Self-launch: N child processes of this script with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set (what
    torch.distributed.run would do), rank 0's stdout relayed.  The parent never initialises the GPU.

    All children are polled: the first one that exits non-zero (an import error, no device, a failed RCCL rendezvous) ends the
    run -- the others, which would otherwise sit in the rendezvous or in a collective until the driver's time limit, are
    terminated (fresh child processes only: nothing is ever re-executed), and the parent exits non-zero with the tail of the
    failing rank's stderr.  AMM_BENCH_TIMEOUT (seconds) bounds the whole run the same way."""
    import socket
    import subprocess
    import tempfile
    with socket.socket() as sock:
        sock.bind(('127.0.0.1', 0))
        port = sock.getsockname()[1]
    if timeout_s is None:
        timeout_s = float(os.environ.get('AMM_BENCH_TIMEOUT', '0')) or None
    procs, logs = [], []
    out_file = tempfile.TemporaryFile()
    for rank in range(n):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(n), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
        env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        logs.append(tempfile.TemporaryFile())
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=out_file if rank == 0 else subprocess.DEVNULL, stderr=logs[-1]))

    def tail(f, n_bytes=4000):
        f.seek(0, 2)
        size = f.tell()
        f.seek(max(0, size - n_bytes))
        return f.read().decode(errors='replace')

    t0 = time.monotonic()
    failed = None
    while True:
        codes = [p.poll() for p in procs]
        bad = [r for r, c in enumerate(codes) if c not in (None, 0)]
        if bad:
            failed = (bad[0], 'exit code %s' % codes[bad[0]])
            break
        if all(c == 0 for c in codes):
            break
        if timeout_s and time.monotonic() - t0 > timeout_s:
            running = [r for r, c in enumerate(codes) if c is None]
            failed = (running[0], 'no result after %.0f s (AMM_BENCH_TIMEOUT)' % timeout_s)
            break
        time.sleep(poll_s)
    if failed is not None:
        for p in procs:
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(timeout=10)
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()
    out_file.seek(0)
    for line in out_file.read().decode(errors='replace').splitlines():    # the JSON line to stdout, library chatter (if any) to stderr
        print(line, file=sys.stdout if line.lstrip().startswith('{') else sys.stderr, flush=True)
    if failed is not None:
        rank, why = failed
        sys.stderr.write('bench.py: rank %d failed (%s); the other ranks were terminated.  Its stderr (tail):\n%s\n' % (rank, why, tail(logs[rank])))
        raise SystemExit(1)
    if os.environ.get('AMM_BENCH_VERBOSE_RANKS'):
        for rank, f in enumerate(logs):
            sys.stderr.write('--- rank %d stderr ---\n%s\n' % (rank, tail(f)))


def dry_run():
    """AMM_BENCH_DRYRUN=1 (CPU tests of the launcher): the ranks rendezvous over gloo, all-reduce their rank numbers and rank 0
    prints a line -- everything bench.py does around the GPU work, none of the GPU work."""
    import torch
    import torch.distributed as dist
    rank, world = int(os.environ.get('RANK', '0')), int(os.environ.get('WORLD_SIZE', '1'))
    if os.environ.get('AMM_BENCH_DRYRUN_FAIL_RANK') == str(rank):       # (test of the launcher: a rank that dies during start-up)
        raise SystemExit('rank %d: simulated start-up failure' % rank)
    os.environ.setdefault('GLOO_SOCKET_IFNAME', 'lo')
    dist.init_process_group('gloo', rank=rank, world_size=world)
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    dist.all_reduce(t)
    dist.barrier()
    if rank == 0:
        print(json.dumps({'dryrun': True, 'n_gpus': world, 'rank_sum': t.item()}), flush=True)
    dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=500)
    ap.add_argument('--warmup', type=int, default=50)
    ap.add_argument('--nside', type=int, default=32, help='waters per box edge (32 -> 98 304 atoms)')
    ap.add_argument('--outer', choices=['damped', 'pme'], default='damped',
                    help="group-2 force: DampedSmoothedForce (headline, SURVEY 8d C3 i) or the PME NonbondedForce (C3 ii)")
    ap.add_argument('--config', choices=['c3', 'c5', 'c2'], default='c3',
                    help='c3: the headline 98 304-atom TIP3P RESPA box; c5: ~249 000-atom solvated chain, RESPA + exceptions + AFED (2 fs inner step); '
                         'c2: 32 768-atom Lennard-Jones fluid, NearNonbondedForce only (one GPU)')
    ap.add_argument('--outer-skin', type=float, default=None, help='dual Verlet list: buffer of the cell-built outer list in nm (default: single list)')
    ap.add_argument('--skin', type=float, default=None, help='Verlet buffer in nm (default: the library default, 0.1)')
    ap.add_argument('--option', action='append', default=[], help='name=value: a context option of the library (tuning; not for the headline)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-relax', action='store_true')
    ap.add_argument('--verbose', action='store_true')
    ap.add_argument('--pme-steps', type=int, default=100,
                    help='also time this many steps with the PME NonbondedForce as the outer force (what RESPASystem leaves in group 2 for a '
                         'PME source, systems.py:74-75) and report them under detail.pme_outer; 0 skips it')
    ap.add_argument('--no-legs', action='store_true',
                    help='the default line also times BASELINE configs 2 and 5 (short regions, child processes) under detail.c2 / detail.c5; this skips them')
    ap.add_argument('--leg-steps', type=int, nargs=2, default=[2000, 60], metavar=('C2', 'C5'), help='timed steps of the two legs')
    ap.add_argument('--cpu-baseline-child', default=None, help=argparse.SUPPRESS)      # internal: cpu_baseline_in_child
    args = ap.parse_args()
    EXTRA_OPTIONS.extend(tuple(item.split('=', 1)) for item in args.option)
    if args.cpu_baseline_child:          # (no GPU in this process)
        state = np.load(args.cpu_baseline_child)
        print(json.dumps(cpu_baseline(args.nside, (4, 2, 1), 4.0, state=(state['x'], state['v']))), flush=True)
        return

    # `python bench.py --gpus N` without a launcher: start the N ranks ourselves (one process per GPU), BEFORE anything touches
    # the GPU -- this process only waits and relays rank 0's JSON line
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        return launch_ranks(args.gpus)

    if os.environ.get('AMM_BENCH_DRYRUN') == '1':
        return dry_run()

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X: the HIP path has no CPU fallback')
    # one rank per GPU over RCCL; AMM_BENCH_BACKEND=gloo lets several ranks share one card to REHEARSE the
    # multi-rank path on a 1-GPU box (numbers from such a run mean nothing)
    backend = os.environ.get('AMM_BENCH_BACKEND', 'nccl')
    if backend != 'nccl':
        local = local % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local)
    if world > 1 or os.environ.get('AMM_FORCE_COLLECTIVES') == '1':
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29533')
        # one node: the bootstrap sockets of c10d / RCCL stay on the loopback interface (no interface walk, no name look-ups on
        # hosts without a resolver); the data path over xGMI is not affected
        os.environ.setdefault('NCCL_SOCKET_IFNAME', 'lo')
        if backend == 'nccl':
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device('cuda', local))
        else:
            os.environ.setdefault('GLOO_SOCKET_IFNAME', 'lo')
            dist.init_process_group(backend, rank=rank, world_size=world)
    if args.gpus != world and rank == 0:
        print('warning: --gpus %d but WORLD_SIZE %d; using WORLD_SIZE' % (args.gpus, world), file=sys.stderr)

    def log(msg):
        if args.verbose and rank == 0:
            print('[bench] ' + msg, file=sys.stderr, flush=True)

    if args.config == 'c2':
        if world > 1:
            raise SystemExit('bench.py --config c2 is a one-GPU line')
        return bench_c2(args, torch)
    loops, dt_fs = (4, 2, 1), 4.0
    t_setup = time.perf_counter()
    if args.config == 'c5':
        # AFED wraps the RESPA integrator: one integrator step = 2 n = 4 RESPA steps of 2 fs (the chain's bonds are stiff)
        inner_fs, substeps = 2.0, 2
        simulation, case = build_simulation_c5(loops, inner_fs, substeps, args.skin)
        dt_fs = 2 * substeps * inner_fs
        args.no_cpu_baseline = True
    else:
        simulation, case = build_simulation(args.nside, loops, dt_fs, args.outer, args.skin, args.outer_skin)
    eng = simulation.context._engine
    n = eng.n
    log('system built in %.1f s: %d atoms' % (time.perf_counter() - t_setup, n))
    relaxed = 0 if args.no_relax else relax(simulation, torch, log=log)
    log('relaxed for %d steps, T = %.1f K' % (relaxed, temperature(eng, torch)))

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    simulation.step(args.warmup)
    near_id = eng.pair_force_ids(1)[0]
    far_id = eng.pair_force_ids(2)[0]
    st0 = {fid: eng.ctx.pair_stats(fid) for fid in (near_id, far_id)}
    # HIP events around every pair-kernel launch, on the launch stream: the stand-alone near evaluation (the kernel the
    # metric names) under the near force's id, the dual pass (outer + near force in one traversal: the dominant kernel)
    # under the outer force's id
    eng.ctx.profile_enable(True)
    rs0 = eng.ctx.run_stats() if hasattr(eng.ctx, 'run_stats') else {}
    fence()
    t0 = time.perf_counter()
    simulation.step(args.steps)
    fence()
    elapsed = time.perf_counter() - t0
    n_near, ms_near = eng.ctx.profile_read(near_id)
    n_dual, ms_dual = eng.ctx.profile_read(far_id)
    eng.ctx.profile_enable(False)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device='cuda')
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()
    st1 = {fid: eng.ctx.pair_stats(fid) for fid in (near_id, far_id)}
    T_end = temperature(eng, torch)
    ms_per_step = elapsed / args.steps * 1e3
    ns_day = dt_fs * 1e-6 * 86400.0 / (elapsed / args.steps)
    # pairs actually inside the cutoffs at the final configuration: one counting launch per force over the neighbour rows
    # (fp64, outside the timed region); directed entries / 2 = pairs counted once (this rank's slice of the rows)
    pairs_near = eng.ctx.pair_count_within(near_id, eng.x, 0.7) / 2
    pairs_far = eng.ctx.pair_count_within(far_id, eng.x, 1.0) / 2
    # molecule rows: lane-trips the walks execute against the entries they hold (a wavefront's rows run to the longest of them)
    padding = {}
    if hasattr(eng.ctx, 'pair_row_padding'):
        for name, fid in (('near', near_id), ('outer', far_id)):
            slots, entries = eng.ctx.pair_row_padding(fid)
            if slots:
                padding[name] = {'lane_trips': slots, 'entries': entries, 'used': round(entries / slots, 4)}

    # the same box with the PME NonbondedForce as the outer force (what RESPASystem leaves in group 2 for a PME source,
    # systems.py:74-75; SURVEY 8d C3 ii): continued from the state the headline run ended in, its own short timed region
    pme_outer = None
    if args.pme_steps > 0 and args.outer == 'damped' and args.config == 'c3' and world == 1:      # (a second context + communicator only on the 1-GPU line)
        from atomsmm_amd import unit
        sim2, _ = build_simulation(args.nside, loops, dt_fs, 'pme', args.skin, args.outer_skin)
        sim2.context.setPositions(eng.x.cpu().numpy() * unit.nanometers)
        sim2.context.setVelocities(eng.v.cpu().numpy())
        sim2.step(max(10, args.warmup // 2))
        fence()
        t1 = time.perf_counter()
        sim2.step(args.pme_steps)
        fence()
        el2 = time.perf_counter() - t1
        if world > 1:
            t = torch.tensor([el2], dtype=torch.float64, device='cuda')
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el2 = t.item()
        # reciprocal space alone: stand-alone evaluations of the mesh force (bin, spread, two FFTs, convolution, gather), HIP events
        recip_us = None
        eng2 = sim2.context._engine
        rec = eng2.recip_force_ids(2)
        if rec:
            scratch = torch.empty_like(eng2.x)
            for _ in range(3):
                eng2.ctx.force_eval(rec[0], eng2.x, scratch)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(20):
                eng2.ctx.force_eval(rec[0], eng2.x, scratch)
            e1.record()
            torch.cuda.synchronize()
            recip_us = round(1e3 * e0.elapsed_time(e1) / 20, 1)
        pme_outer = {'ms_per_step': round(el2 / args.pme_steps * 1e3, 4), 'ns_day': round(dt_fs * 1e-6 * 86400.0 / (el2 / args.pme_steps), 2),
                     'recip_us': recip_us, 'steps': args.pme_steps,
                     'workload': 'the same box and RESPA split, outer force = PME NonbondedForce (rc 1.0, switch 0.9, Ewald tolerance 5e-4: '
                                 'direct space + reciprocal space on an 80^3 mesh), continued from the final state of the headline run'}
        del sim2

    if rank == 0:
        from atomsmm_amd import backend
        t_near = ms_near / max(n_near, 1) * 1e-3          # s per launch (this rank's slice)
        t_dual = ms_dual / max(n_dual, 1) * 1e-3
        atoms_per_launch = st1[near_id]['n_slice_atoms']
        near_stats = st1[near_id]
        flop_dual = FLOP_PER_PAIR_FAR * pairs_far + FLOP_PER_PAIR_NEAR * pairs_near
        # HBM traffic from the PMC passes (rocprofv3 cannot run inside this process: scripts/measure_round.sh makes them on
        # this same command and commits the summary); quoted only while the kernels are the ones it was measured on
        traffic = {}
        tfile = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'profiles', TRAFFIC_FILE)
        if world == 1 and args.nside == 32 and args.config == 'c3' and os.path.exists(tfile):
            try:
                stored = json.load(open(tfile))
                if stored.get('kernel_revision') == backend.kernel_revision():
                    traffic = stored
            except Exception:
                pass
        if world == 1 and args.config == 'c5':
            # a hybrid list's evaluation is two launches (molecule rows + per-atom part): their traffic added up
            stored = stored_traffic(backend)
            for tag in ('near', 'dual'):
                a_, b_ = stored.get('c5_' + tag), stored.get('c5_%s_rest' % tag)
                if a_ and b_:
                    traffic[tag] = {'hbm_bytes_per_launch': a_['hbm_bytes_per_launch'] + b_['hbm_bytes_per_launch'],
                                    'fetch_size_kb_avg': a_['fetch_size_kb_avg'] + b_['fetch_size_kb_avg'],
                                    'write_size_kb_avg': a_['write_size_kb_avg'] + b_['write_size_kb_avg']}
            if traffic:
                traffic['kernel_revision'] = stored.get('kernel_revision')

        molecule_rows = bool(near_stats.get('list_kind'))
        # molecule rows with a fused kernel for the two families: the near force of the step boundary rides on the outer force's launch
        # (its own launches are the stand-alone evaluations of the middle RESPA loop); without one it is a second launch over the rows
        fused_rows = molecule_rows and bool(near_stats.get('rode_along'))
        two_launches = molecule_rows and not fused_rows
        kname = 'k_cpair' if molecule_rows else 'k_pair_tab'
        outer_name = 'DAMPED' if args.outer == 'damped' else 'NONBONDED/Ewald'

        rs = eng.ctx.run_stats() if hasattr(eng.ctx, 'run_stats') else {}
        fused_share = (rs.get('epilogues', 0) - rs0.get('epilogues', 0)) / max(n_near + n_dual, 1)
        epilogues_per_launch = {'near': fused_share, 'dual': fused_share} if fused_share > 0 else {}

        def roofline(kernel, seconds, launches, alg_bytes, flops, tag, npairs, listed):
            achieved = alg_bytes / max(seconds, 1e-12) / 1e9
            tf = flops / max(seconds, 1e-12) / 1e12
            entry = {'bound': 'hbm', 'achieved': round(achieved, 3), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                     'frac': round(achieved / HBM_PEAK_GBS, 6), 'traffic': traffic.get(tag, {}).get('hbm_bytes_per_launch'),
                     'kernel': kernel, 'avg_launch_us': round(seconds * 1e6, 2), 'launches': launches,
                     'algorithmic_bytes_per_launch': alg_bytes, 'fp64_tflops': round(tf, 3),
                     'fp64_frac_of_vector_peak': round(tf / FP64_VECTOR_PEAK_TF, 4),
                     'note': 'FP64-VALU / latency bound, not HBM bound (SURVEY.md 8d); fp64 fraction against the %.1f TF datasheet peak '
                             '(a bare v_fma_f64 loop sustains %.1f TF on this chip: scripts/micro/fp64_rate.hip)'
                             % (FP64_VECTOR_PEAK_TF, FP64_SUSTAINED_TF)}
            if epilogues_per_launch.get(tag):
                # the launch also ran the inner RESPA loop of its molecules (csrc/cluster.hip: cepi_rows): the kicks, moves and bond-list
                # evaluations that were launches of their own until round 4.  `frac` keeps SURVEY 8d's bytes of the PAIR evaluation (what
                # earlier rounds quoted) over the whole launch; with the loop's compulsory traffic -- read + write x, v, f0 (144 B), mass
                # (8 B), the next evaluation's sorted copy (48 B) per atom -- the launch moves this many algorithmic bytes:
                with_epi = alg_bytes + EPILOGUE_BYTES_PER_ATOM * atoms_per_launch
                entry['epilogue'] = {'share_of_launches': round(epilogues_per_launch[tag], 3), 'algorithmic_bytes_per_launch': with_epi,
                                     'achieved': round(with_epi / max(seconds, 1e-12) / 1e9, 3),
                                     'frac': round(with_epi / max(seconds, 1e-12) / 1e9 / HBM_PEAK_GBS, 6),
                                     'note': 'inner RESPA loop of the rows\' molecules (4 x {kick, move, bonds + angle, kick} + the '
                                             'preceding kicks) inside this launch: ~12 us of its duration'}
            if entry['traffic'] is not None:
                t_ = traffic.get(tag, {})
                # FETCH_SIZE halves WIDE streaming reads on gfx950 (MI355X_MICROARCH.md); this kernel's reads are 32-byte gathers,
                # for which the correction is uncalibrated: the truth lies between the raw and the doubled figure
                entry['traffic_raw'] = int((t_.get('fetch_size_kb_avg', 0) + t_.get('write_size_kb_avg', 0)) * 1024)
                entry['traffic_corrected'] = entry['traffic']
                if t_.get('valu_insts_per_launch') and npairs:
                    # lane-instructions (64 x SQ_INSTS_VALU) per in-cutoff pair counted once (the pass evaluates both directions)
                    entry['valu_insts_per_pair'] = round(64.0 * t_['valu_insts_per_launch'] / npairs, 1)
                    if listed:   # ... and per pair evaluation (a listed pair, each direction, inside the cutoff or not)
                        entry['valu_insts_per_listed_pair'] = round(64.0 * t_['valu_insts_per_launch'] / listed, 1)
                entry['traffic_source'] = 'profiles/%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, bytes per launch, kernels %s)' % (
                    TRAFFIC_FILE, traffic.get('kernel_revision'))
            return entry
        result = {
            'metric': 'ns/day on 100k-atom TIP3P RESPA box; near-nonbonded HBM GB/s vs 8 TB/s peak',
            'value': round(ns_day, 3), 'unit': 'ns/day', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(ms_per_step, 4), 'higher_is_better': True, 'scaling': 'strong', 'vs_baseline': None,
            'dtype': 'f64', 'data': 'synthetic',
            'config': {'workload': ('C5: %d atoms (%d TIP3P waters + 3000-atom bonded chain with 1-4 exceptions + 30-atom solute, L = %.3f nm), '
                                    'SolvationSystem -> RESPASystem(0.7, 0.5) + DampedSmoothedForce outer force, AdiabaticDynamicsIntegrator('
                                    'RespaPropagator([4,2,1]) at 2 fs, n = 2, lambda_vdw): 8 fs per step' % (n, case['n_waters'], case['box'][0]))
                       if args.config == 'c5' else
                       'C3: %d-atom flexible TIP3P box (L = %.3f nm), RESPASystem(0.7, 0.5, force-switch) + '
                       '%s outer force, RespaPropagator([4,2,1]), 4 fs outer step'
                       % (n, case['box'][0], 'DampedSmoothedForce(2.9/nm, 1.0, 0.9)' if args.outer == 'damped'
                          else 'PME NonbondedForce (rc 1.0, switch 0.9, tol 5e-4: direct + reciprocal space)'),
                       'atoms': n, 'loops': list(loops), 'outer_step_fs': dt_fs, 'relax_steps': relaxed,
                       'parallelism': ('atom decomposition x%d, %s' % (world, 'owner-integrates: all-gather of positions + velocities of the molecules a rank walked (force slices at the first and last evaluation of a call); RCCL, library-owned communicator'
                                                                     if getattr(eng, '_native_comm', False) else 'collectives through torch.distributed')) if world > 1 else 'single GPU',
                       'temperature_K_end': round(T_end, 1)},
            # the kernel the metric names: the near-force traversal (group 1, force only).  Molecule rows: the same kernel serves the
            # stand-alone evaluation and the near force's launch of the step-boundary pass (front parts of the shared rows)
            'roofline': roofline('%s<NEAR_FSWITCH> (group-1 near force, force only)' % kname, t_near, n_near,
                                 BYTES_PER_ATOM * atoms_per_launch, FLOP_PER_PAIR_NEAR * pairs_near, 'near', pairs_near, near_stats['n_list_pairs']),
            # the dominant kernel of the step: the outer force over the whole rows.  Molecule rows: one force per launch (72 B per
            # atom); per-atom rows (--option cluster=0): outer + near force in one traversal (96 B per atom, both forces' flops)
            'roofline_dominant': (roofline('%s<%s> (group-2 outer force, force only)' % (kname, outer_name), t_dual, n_dual,
                                           BYTES_PER_ATOM * atoms_per_launch, FLOP_PER_PAIR_FAR * pairs_far, 'outer', pairs_far, st1[far_id]['n_list_pairs'])
                                  if two_launches else
                                  roofline('%s<%s, guest NEAR_FSWITCH> (outer + near force in one pass)' % (kname, outer_name),
                                           t_dual, n_dual, BYTES_PER_ATOM_DUAL * atoms_per_launch, flop_dual, 'dual', pairs_far, st1[far_id]['n_list_pairs'])),
            'detail': {'near_kernel_us': round(t_near * 1e6, 2), 'near_launches': n_near,
                       'outer_kernel_us' if two_launches else 'dual_kernel_us': round(t_dual * 1e6, 2), 'outer_launches' if two_launches else 'dual_launches': n_dual,
                       'step_boundary_pass_us': round((t_dual + (t_near if two_launches else 0.0)) * 1e6, 2),
                       'near_list_prunes_in_timed_region': st1[near_id]['n_builds'] - st0[near_id]['n_builds'],
                       'far_list_prunes_in_timed_region': st1[far_id]['n_builds'] - st0[far_id]['n_builds'],
                       'outer_list_builds_in_timed_region': st1[far_id]['n_outer_builds'] - st0[far_id]['n_outer_builds'],
                       'outer_rlist_nm': st1[far_id]['rlist_outer'], 'shared_list': bool(near_stats['shares_list']),
                       'near_lanes_per_atom': near_stats['lanes_per_atom'], 'near_rlist_nm': near_stats['rlist'],
                       'near_list_pairs': near_stats['n_list_pairs'], 'far_list_pairs': st1[far_id]['n_list_pairs'],
                       'pairs_within_0.7nm_counted': int(pairs_near), 'pairs_within_1.0nm_counted': int(pairs_far),
                       'kernel_revision': backend.kernel_revision(), 'rows': {0: 'one per atom', 1: 'one per molecule',
                                2: 'hybrid: one per three-site molecule for the pairs of two molecules + per-atom rows for every pair with one of the '
                                   '%d other atoms (kernel times: both launches of an evaluation)' % near_stats.get('n_rest_atoms', 0),
                                3: 'none'}.get(near_stats.get('list_kind'), '?'),
                       'row_padding': padding or None, 'pme_outer': pme_outer,
                       'run_stats': {k: rs.get(k, 0) - rs0.get(k, 0) for k in rs} or None},
        }
        if world == 1 and not args.no_cpu_baseline and args.outer == 'damped':
            result['cpu_baseline'] = cpu_baseline_in_child(args.nside, eng.x.cpu().numpy(), eng.v.cpu().numpy())
        if world == 1 and args.config == 'c3' and args.outer == 'damped' and args.nside == 32 and not args.no_legs:
            # BASELINE configs 2 and 5 on the same line: short timed regions after the headline's, each in a child
            # process; configs[1] = the 32k-atom Lennard-Jones fluid, configs[4] = the ~250k-atom solvated chain under AFED
            result['detail']['c2'] = run_leg('c2', args.leg_steps[0], 200)
            result['detail']['c5'] = run_leg('c5', args.leg_steps[1], 20)
        print(json.dumps(result), flush=True)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
