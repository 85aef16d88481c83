/*
 * include/atomsmm_hip.h -- C-ABI of libatomsmm_hip.so (MI355X / gfx950).
 *
 * Drop-in boundary for ONE hot path of AtomsMM: the RESPA-split nonbonded pair forces and the
 * multiple-timescale propagator inner loop.  AtomsMM has no FFI of its own: it subclasses OpenMM's
 * SWIG classes and OpenMM's C++ does the arithmetic.  Each entry point below
 * therefore names the OpenMM call *as made by the reference* (file:line in /root/reference) whose
 * work it takes over.  INTEGRATION.md shows the ctypes binding a maintainer adds.
 *
 * Conventions
 *   - return 0 = OK; non-zero = error, message via amm_last_error().
 *   - `h_` pointers are HOST memory (copied during the call, caller keeps ownership);
 *     `d_` pointers are DEVICE memory owned by the caller (e.g. torch tensors), fp64, C-contiguous;
 *     per-atom vectors are AoS [n_atoms][3].
 *   - units: nm, ps, dalton, kJ/mol, elementary charge (OpenMM's unit system).
 *   - all work is enqueued on the context's HIP stream; nothing synchronises unless stated.
 *   - one context per process per GPU; not thread-safe.
 *   - atom decomposition: amm_set_slice(rank, world) makes pair forces compute only the rows of this
 *     rank's slice of the cell-sorted order (owner-computes: every force row has one producer).  The
 *     default exchange of a group that holds one pair force is an ALL-GATHER of those slices
 *     (AMM_EXCHANGE_GATHER: each rank leaves its rows, in sorted order, in its chunk of the exchange
 *     buffer; 1/world of the bytes of an all-reduce and no additions); groups that also hold sliced
 *     bond lists or reciprocal space, and hybrid lists, write zeros for the rows they do not own and
 *     all-reduce(sum) the buffer (AMM_OP_ALLREDUCE).  Either is issued by the library itself on its
 *     own RCCL communicator (amm_comm_init), which keeps a whole multi-rank step program inside one
 *     amm_run_ops call, or by the host (torch.distributed) between amm_run_ops calls.
 */
#ifndef ATOMSMM_HIP_H
#define ATOMSMM_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AMM_ABI_VERSION 1

/* ---- pair-energy families -------------------------------------------------------------------- */
enum {
    AMM_NEAR_NONE = 0,      /* S(u) V_LJC                      NearForce._expressions forces.py:541-543 */
    AMM_NEAR_SHIFT = 1,     /* S(u) (V_LJC(r) - V_LJC(rc0))    forces.py:544-548                        */
    AMM_NEAR_FSWITCH = 2,   /* force-switched LJC              forces.py:549-563 (V' = S V'_LJC, :628)  */
    AMM_DAMPED = 3,         /* DampedSmoothedForce             forces.py:448-455                        */
    AMM_NONBONDED = 4,      /* _AtomsMM_NonbondedForce direct space  forces.py:134-190, 723             */
    AMM_SOFTCORE = 5,       /* SolvationSystem's solute-solvent softcore LJ, 4 lambda eps (1-x)/x^2 with
                               x = (r/sigma)^6 + (1-lambda)/2, restricted to an interaction group
                               (systems.py:266-272).  lambda = desc.alpha; the charge array carries the set of
                               each atom (1, 2, 0 = neither; use Kc = 1): a pair counts iff the codes multiply
                               to 2; AMM_SWITCH = OpenMM's built-in switch imported with the cutoff            */
    AMM_LJ_VIRIAL = 6       /* ComputingSystem's dispersion virial as an energy, 24 eps (2 (sigma/r)^12 - (sigma/r)^6)
                               (systems.py:894), AMM_SWITCH as imported                                        */
};
enum {
    AMM_GUARD_RC0 = 1,      /* energy *= step(rc0 - r)         forces.py:661, 714 ; systems.py:73      */
    AMM_COULOMB_EWALD = 2,  /* NONBONDED: Kc qq erfc(alpha r)/r                                         */
    AMM_COULOMB_RF = 4,     /* NONBONDED: reaction field (parity unpinned)              */
    AMM_SWITCH = 8,         /* NONBONDED / SOFTCORE: OpenMM built-in switch rswitch -> rc                */
    AMM_NO_SHIFT = 16,      /* NEAR_FSWITCH: energy without the constant -V*(rc0) (AlchemicalRespaSystem's
                               force-switched potentials, systems.py:823-846)                            */
    AMM_GROUP_LJ = 32,      /* interaction group for Lennard-Jones-only forces: the charge array carries the set
                               of each atom (1, 2, 0; Kc = 1), a pair counts iff the codes multiply to 2, no
                               Coulomb term (systems.py:739-772)                                         */
    AMM_GROUP_Q = 64        /* interaction group for Coulomb-only forces (the force-switched solute-solvent
                               electrostatics of Coulomb scaling, systems.py:696-708, 848-856): the sigma array
                               carries TWICE the set code of each atom (2, 4, 0): the mixed sigma (sigma_i +
                               sigma_j)/2 = 3 selects a (set 1, set 2) pair; epsilon is ignored         */
};

typedef struct {
    int32_t family;
    int32_t flags;
    int32_t degree;   /* DAMPED: u = (r^d - rs^d)/(rc^d - rs^d); d = 1 is OpenMM's built-in switch */
    int32_t pad_;
    double sign;      /* +1 / -1 (subtract=True forces.py:662; discount forces.py:714)            */
    double rc;        /* cutoff actually used: pairs with r >= rc skipped                          */
    double rswitch;   /* DAMPED / NONBONDED switch start                                           */
    double rc0, rs0;  /* near-force cutoff / switch start                                          */
    double alpha;     /* erfc damping, 1/nm                                                        */
    double Kc;        /* 138.935456  forces.py:407                                                 */
    double krf, crf;  /* reaction field constants                                                  */
} amm_pair_desc;

/* ---- bonded term kinds (owner-computes, no atomics) ------------------------------------------ */
enum {
    AMM_BOND_HARMONIC = 0,   /* idx[2], params (r0, k)            OpenMM HarmonicBondForce (group 0)       */
    AMM_ANGLE_HARMONIC = 1,  /* idx[3], params (theta0, k)        OpenMM HarmonicAngleForce                */
    AMM_BOND_LJC = 2,        /* idx[2], params (qq, sigma, eps)   NonbondedExceptionsForce forces.py:400-407 */
    AMM_BOND_NEAR = 3,       /* idx[2], params (qq, sigma, eps)   NearExceptionForce forces.py:673-680     */
    AMM_TORSION_PERIODIC = 4,/* idx[4], params (n, phase, k)      OpenMM PeriodicTorsionForce              */
    AMM_BOND_EWALD_EXCL = 5, /* idx[2], params (qi*qj)            -Kc qi qj erf(alpha r)/r, NonbondedForce exclusion term */
    AMM_BOND_VIRIAL_HARMONIC = 6, /* idx[2], params (r0, k)       -k r (r - r0): bond-stretching virial, ComputingSystem systems.py:914 */
    AMM_BOND_VIRIAL_LJ = 7   /* idx[2], params (qq, sigma, eps)   24 eps (2 x^2 - x), x = (sigma/r)^6: exception virial systems.py:898 */
};

/* ---- step-program ops (one flat, unrolled outer step; built by the host from the step program
 *      that RespaPropagator.addSteps emits, propagators.py:933-973) ------------------------------ */
enum {
    AMM_OP_EVAL = 1,   /* a = group: buffer[group_slot(a)] <- sum of the group's forces at current x */
    AMM_OP_KICK = 2,   /* v <- v + coef*(buf[a] -/+ buf[b])/m  (b = -1: single buffer; c = 1: plus)  propagators.py:271,
                          force expressions (f0), (f2-f1), (f0+fm1) ... of propagators.py:917-928 */
    AMM_OP_MOVE = 3,   /* x <- x + coef*v                                                propagators.py:249 */
    AMM_OP_COPY = 4,   /* buf[a] <- buf[b]                          integrators.py:139-144 (`_f2_ <- f2`)    */
    AMM_OP_COMBINE = 5,/* buf[a] <- buf[b] + coef*buf[c]            propagators.py:951 (`fm2 <- f2-f1`)       */
    AMM_OP_EXPR = 6,   /* buf[b] <- per-DOF expression a (amm_expr_define): bath steps inside a RESPA loop, e.g. a
                          thermostat update on a per-DOF variable (NHL_R, integrators.py:272-318); each execution draws from the
                          next random-stream counter                                                                  */
    AMM_OP_BATH = 7,   /* v <- z v + sqrt(kT (1 - z^2)/m) gaussian with (z, kT) = bath a (amm_bath_define): the Ornstein-Uhlenbeck
                          step of OrnsteinUhlenbeckPropagator on (v, m) without force (propagators.py:727-741), as a native op so
                          that the inner-loop kernel can carry it (Langevin_R 'middle' scheme)                        */
    AMM_OP_SAVE_REF = 8,    /* reference positions of the constraint solver <- x (start of a step)                   */
    AMM_OP_CONSTRAIN_X = 9, /* addConstrainPositions (propagators.py:250, 1129): SHAKE along the reference bond vectors,
                               then reference <- x                                                                    */
    AMM_OP_CONSTRAIN_V = 10,/* addConstrainVelocities (propagators.py:272, 1131): RATTLE                              */
    AMM_OP_ALLREDUCE = 11   /* buf[a] <- sum over ranks of buf[a] (RCCL, the context's own communicator: amm_comm_init):
                               the exchange of atom decomposition after the EVAL of a sliced group      */
};
typedef struct {
    int32_t op, a, b, c;
    double coef;
} amm_op;

typedef struct amm_ctx amm_ctx;

int amm_abi_version(void);
const char *amm_last_error(void);

/* Context.__init__ / Context.setPeriodicBoxVectors  (utils.py:153-155 builds the Simulation/Context).
 * stream: hipStream_t as void* (NULL = default stream). */
int amm_create(int32_t n_atoms, const double h_box[3], int32_t device, void *stream, amm_ctx **out);
int amm_destroy(amm_ctx *ctx);
int amm_set_stream(amm_ctx *ctx, void *stream);
int amm_set_slice(amm_ctx *ctx, int32_t rank, int32_t world);
/* The library's own RCCL communicator (one rank per process per GPU).  Rank 0 makes an id (amm_comm_unique_id), the
 * host distributes its 128 bytes to all ranks by any means (torch.distributed broadcast, MPI, a file) and every rank
 * calls amm_comm_init (collective, blocks until all ranks joined).  rccl_path: the librccl to bind (a torch process
 * passes torch/lib/librccl.so so that one copy serves both), NULL = the dynamic loader's default.
 * No reference counterpart: AtomsMM/OpenMM are single-device. */
#define AMM_COMM_ID_BYTES 128
int amm_comm_unique_id(const char *rccl_path, uint8_t id[AMM_COMM_ID_BYTES]);
int amm_comm_init(amm_ctx *ctx, const char *rccl_path, const uint8_t id[AMM_COMM_ID_BYTES], int32_t rank, int32_t world);
/* releases the communicator (amm_destroy does it too); collective, like ncclCommDestroy.  Errors of the communicator: an enqueue that
 * fails returns non-zero at once; an ASYNCHRONOUS error (a peer rank died, a link failed: ncclCommGetAsyncError) is polled after every
 * collective the library enqueues and by amm_check / amm_synchronize / amm_comm_destroy, whose waits for the stream are bounded by the
 * option "comm_timeout" -- in either case the communicator is aborted (ncclCommAbort), the call returns non-zero with the reason in
 * amm_last_error, and every later collective of the context fails the same way (failure detection). */
int amm_comm_destroy(amm_ctx *ctx);
/* in-place sum over ranks of count doubles in device memory, on the context's stream */
int amm_comm_allreduce(amm_ctx *ctx, double *d_buf, int64_t count);
/* Exchange of owner-computed force slices by ALL-GATHER instead of all-reduce (1/world of the bytes, no additions).
 * A group in AMM_EXCHANGE_GATHER mode must hold exactly one pair force.  Its EVAL op leaves the rows of the rank's slice
 * of the cell-sorted order in chunk `rank` of the exchange buffer (caller-owned, world x 2 x ceil(n/world) x 3 doubles;
 * a dual evaluation fills [2][ceil(n/world)][3] per chunk, a single one [ceil(n/world)][3]), all-gathers the chunks on
 * the library's communicator and spreads them to the group's buffer in atom order (every rank holds the same
 * permutation).  Without a communicator the EVAL must be the last op of its amm_run_ops call: the host gathers the
 * chunks (count per rank = forces x ceil(n/world) x 3 doubles) and calls amm_exchange_finish. */
enum { AMM_EXCHANGE_REDUCE = 0, AMM_EXCHANGE_GATHER = 1 };
int amm_group_set_exchange(amm_ctx *ctx, int32_t group, int32_t mode);
int amm_bind_exchange(amm_ctx *ctx, double *d_buf, int64_t n_doubles);
int amm_exchange_finish(amm_ctx *ctx);
/* *nf = 0: no exchange is waiting; else every rank's chunk of the exchange buffer holds nf x amm_exchange_per() x 3 doubles to all-gather
 * (1: one force; 2: the two forces of a dual evaluation, or positions + velocities of a state exchange -- see amm_run_stats) */
int amm_exchange_pending(amm_ctx *ctx, int32_t *nf);
/* out[0] = collectives issued so far, out[1] = doubles per rank they carried (AMM_OP_ALLREDUCE ops on buffers that are
 * neighbours in memory are merged into one message) */
int amm_comm_stats(amm_ctx *ctx, int64_t out[2]);
int amm_synchronize(amm_ctx *ctx);
/* Raises pending device-side errors (neighbour-list overflow, NaN guard): returns non-zero + message. Synchronises. */
int amm_check(amm_ctx *ctx);

/* CustomNonbondedForce(energy) + addParticle + addExclusion  (forces.py:225, 299-312; systems.py:97-111).
 * h_excl: E pairs (i,j) -- every exception of the source NonbondedForce becomes an exclusion.
 * skin: Verlet buffer (nm) for the cell-list-built neighbour list; <0 = default. */
int amm_pair_create(amm_ctx *ctx, const amm_pair_desc *desc, const double *h_q, const double *h_sigma,
                    const double *h_eps, const int32_t *h_excl, int32_t n_excl, double skin,
                    int32_t *force_id);
/* setParticleParameters + updateParametersInContext / Context.setParameter for offset parameters
 * (forces.py:247-258, 292-309: charge+lambda*chargeScale ...). Effective values are passed. */
int amm_pair_set_params(amm_ctx *ctx, int32_t force_id, const double *h_q, const double *h_sigma,
                        const double *h_eps);
/* RESPASystem puts a short-ranged copy (group 1, rcutIn) and the full force (group 2) over the SAME particles and
 * exclusions (systems.py:71-77): let `force_id` traverse the front part of `host_id`'s neighbour rows instead of
 * building its own list.  Call before the first evaluation; exclusions must be identical. */
int amm_pair_share_list(amm_ctx *ctx, int32_t force_id, int32_t host_id);

/* CustomBondForce / HarmonicBondForce / HarmonicAngleForce term lists of one force group. */
/* context.setParameter('lambda_vdw', value) for a softcore pair force (systems.py:267: global parameter). */
int amm_pair_set_lambda(amm_ctx *ctx, int32_t force_id, double value);
/* The same with lambda left on the device (an AFED step moves it with amm_expr_eval_scalar): the force's kernels read *d_lambda at
 * launch time from now on; NULL, or a later amm_pair_set_lambda, returns to the host's number.  Only the list-free evaluation of a
 * softcore force with a small set (csrc/group.hip: what SolvationSystem's solute gets) supports it; others return an error. */
int amm_pair_set_lambda_dev(amm_ctx *ctx, int32_t force_id, const double *d_lambda);
/* Overall factor of a pair force (desc.sign): a global parameter that multiplies the whole energy -- `respa_switch`, or
 * the coupling function of a CustomCVForce over this force (systems.py:738-772) -- changed by setParameter. */
int amm_pair_set_scale(amm_ctx *ctx, int32_t force_id, double scale);
/* deriv(energy, lambda) of a softcore pair force (addEnergyParameterDerivative, systems.py:712-718; used by the AFED
 * kicks, integrators.py:735-737): *d_out (device) += sum over pairs of dE/dlambda at d_pos. */
int amm_pair_energy_derivative(amm_ctx *ctx, int32_t force_id, const double *d_pos, double *d_out);

int amm_bonded_create(amm_ctx *ctx, int32_t *force_id);
int amm_bonded_add_terms(amm_ctx *ctx, int32_t force_id, int32_t kind, const int32_t *h_idx,
                         const double *h_params, int32_t n_terms, int32_t periodic,
                         const amm_pair_desc *desc_or_null);
int amm_bonded_finalize(amm_ctx *ctx, int32_t force_id);
/* world > 1: evaluate only this rank's block of atoms (use for bonded sets living in an all-reduced group). */
int amm_bonded_set_sliced(amm_ctx *ctx, int32_t force_id, int32_t on);
/* Free a bond-list set the host has replaced: Context.setParameter on a parameter offset rebuilds the exception terms of the
 * NonbondedForce (forces.py:292-309, systems.py:303-311), which OpenMM does in place.  The id is retired (never reused) and
 * leaves every group definition. */
int amm_bonded_release(amm_ctx *ctx, int32_t force_id);

/* Reciprocal space of a NonbondedForce with nonbondedMethod PME / Ewald, as RESPASystem and FarNonbondedForce
 * keep it in group 2 with the source force's Ewald tolerance / PME parameters (systems.py:74-75,
 * forces.py:185-188; setReciprocalSpaceForceGroup: utils.py:147-152).  Smooth PME, B-spline order 5, grid
 * nx x ny x nz; the energy includes the Ewald self term (no neutralising-background term for a charged box: the
 * reference's literals have none, tests/test_systems.py:173).
 * Evaluated through amm_force_eval like the other force objects. */
int amm_pme_create(amm_ctx *ctx, double alpha, int32_t nx, int32_t ny, int32_t nz, double Kc, const double *h_q,
                   int32_t *force_id);
int amm_pme_set_charges(amm_ctx *ctx, int32_t force_id, const double *h_q);   /* parameter offsets on charges */
/* world > 1: gather forces only for this rank's block of atoms (use inside an all-reduced group). */
int amm_pme_set_sliced(amm_ctx *ctx, int32_t force_id, int32_t on);

/* Context.getState(getForces=True, getEnergy=True, groups=...)  (utils.py:159-164).
 * d_force [n][3]: overwritten (accumulate=0) or added to; d_energy: *d_energy += E (skipped if NULL). */
int amm_force_eval(amm_ctx *ctx, int32_t force_id, const double *d_pos, double *d_force,
                   int32_t accumulate, double *d_energy);

/* CustomIntegrator per-DOF steps as the reference emits them (propagators.py:249, 271; integrators.py:113). */
int amm_kick(amm_ctx *ctx, double *d_v, const double *d_f, const double *d_f2, int32_t plus, const double *d_mass, double coef);
int amm_move(amm_ctx *ctx, double *d_x, const double *d_v, double coef);
int amm_copy(amm_ctx *ctx, double *d_dst, const double *d_src);
int amm_mvv(amm_ctx *ctx, const double *d_v, const double *d_m, double *d_out); /* addComputeSum('mvv','m*v*v') */

/* CustomIntegrator.step(n)  (integrators.py:153-163): bind state buffers, define groups, run ops. */
/* CustomIntegrator.addComputePerDof(variable, expression) / addComputeSum(variable, expression) for expressions
 * beyond kick and move -- the thermostat / stochastic propagators (propagators.py:276-827, 1108-2172), and
 * integrators.py:113 `addComputeSum('mvv', 'm*v*v')`.  `code` is a postfix program (opcode | arg << 8; opcodes in
 * atomsmm_amd/expr.py and csrc/expr.hip) over constants, global variables, the bound per-DOF buffers (by slot), the
 * masses and `gaussian` / `uniform` draws (Philox-4x32-10 keyed by seed; counter = launch index).  d_dst ([n][3],
 * may be one of the bound buffers) receives the per-DOF values, *d_sum (device) their sum; either may be NULL. */
int amm_expr_eval(amm_ctx *ctx, const int32_t *code, int32_t n_code, const double *consts, int32_t n_consts,
                  const double *globals, int32_t n_globals, uint64_t seed, uint64_t counter, double *d_dst, double *d_sum);

/* CustomIntegrator.addComputeGlobal(variable, expression) whose operands wait on device results -- the extended variable's scalar
 * block of AdiabaticDynamicsIntegrator (integrators.py:701-737: lambda moves, reflects, is thermostatted, all on sums of
 * deriv(energy, lambda) that kernels already enqueued will leave in device memory).  The postfix program is a sequence of
 * assignments, each closed by X_OUT dst: d_scalars[dst] <- the value on the stack; X_DEVG operands are other entries of d_scalars
 * (n_scalars doubles, device; later assignments see earlier ones); constants as in amm_expr_eval, no per-DOF operands, no random
 * draws (the host makes those); <= 640 words, <= 96 constants.  One thread on the context's stream; the host never waits. */
int amm_expr_eval_scalar(amm_ctx *ctx, const int32_t *code, int32_t n_code, const double *consts, int32_t n_consts, double *d_scalars,
                         int32_t n_scalars);

/* System.addConstraint(i, j, distance) x n (forcefield.createSystem(constraints=HBonds, rigidWater=True) in the
 * reference's tests, tests/test_propagators.py:11-18).  Clusters of coupled constraints (<= 8 atoms, <= 16 constraints)
 * are solved by one thread each; tolerance as CustomIntegrator.getConstraintTolerance() (<= 0: 1e-5). */
int amm_constraints_create(amm_ctx *ctx, const int32_t *h_pairs, const double *h_dist, int32_t n_constraints, double tolerance);

/* Register a per-DOF expression with fixed globals for AMM_OP_EXPR; amm_expr_seed sets the random stream used by the
 * ops (integrator.setRandomNumberSeed, integrators.py:149-151) and restarts its counter. */
int amm_expr_define(amm_ctx *ctx, const int32_t *code, int32_t n_code, const double *consts, int32_t n_consts,
                    const double *globals, int32_t n_globals, int32_t *expr_id);
int amm_expr_seed(amm_ctx *ctx, uint64_t seed);
/* z = exp(-gamma * fraction * dt), kT in kJ/mol: the constants of one Ornstein-Uhlenbeck bath step for AMM_OP_BATH. */
int amm_bath_define(amm_ctx *ctx, double z, double kT, int32_t *bath_id);
/* The Nose-Hoover-Langevin bath block of NHL_R_Integrator (integrators.py:272-330; MassiveNoseHooverLangevinPropagator,
 * propagators.py:1362-1449) as ONE AMM_OP_BATH: v <- v exp(-h w) ; w <- z w + sqrt(kT (1 - z^2)/Q) gaussian +
 * (m v^2 - kT)(1 - z)/(Q friction) ; v <- v exp(-h w), with the per-DOF thermostat velocities w in buffer slot `slot`
 * (h = fraction * dt of the two scalings, z = exp(-2 h friction)). */
int amm_bath_define_nhl(amm_ctx *ctx, double h, double z, double kT, double Q, double friction, int32_t slot, int32_t *bath_id);
/* SIN(R) with L = 1 (SIN_R_Integrator, integrators.py:358-416; MassiveIsokineticPropagator / SIN_R_Propagator,
 * propagators.py:276-355, 1045-1105).  amm_iso_define(on = 1) puts the context in isokinetic mode: every AMM_OP_KICK becomes
 *   v <- v cosh(z) + sqrt(LkT/m) sinh(z), z = coef F / sqrt(m LkT) ; H = sqrt(LkT/(m v^2 + Q1 v1^2/2)) ; v <- H v ; v1 <- H v1
 * with the per-DOF thermostat velocities v1 in buffer slot `slot_v1`.  amm_bath_define_sin registers the bath block between
 * the two half moves -- v1 <- v1 exp(-h v2) ; rescale ; v2 <- z v2 + sqrt(kT (1 - z^2)/Q2) gaussian + (Q1 v1^2 - kT)(1 - z)/
 * (Q2 friction) ; v1 <- v1 exp(-h v2) ; rescale -- as ONE AMM_OP_BATH (v2 in slot_v2; needs the isokinetic mode). */
int amm_iso_define(amm_ctx *ctx, int32_t on, double LkT, double Q1, int32_t slot_v1);
int amm_bath_define_sin(amm_ctx *ctx, double h, double z, double kT, double Q2, double friction, int32_t slot_v2, int32_t *bath_id);

int amm_bind_state(amm_ctx *ctx, double *d_x, double *d_v, const double *d_mass);
#define AMM_MAX_SLOTS 64
#define AMM_SLOT_X 62   /* positions and velocities are addressable as buffers too (`x0 <- x`) */
#define AMM_SLOT_V 63
int amm_bind_buffer(amm_ctx *ctx, int32_t slot, double *d_buf);              /* per-DOF buffers f0.., _f2_, fm1 */
int amm_group_define(amm_ctx *ctx, int32_t group, int32_t slot, const int32_t *force_ids, int32_t n_forces);
int amm_run_ops(amm_ctx *ctx, const amm_op *ops, int32_t n_ops, int32_t repeat);
/* Dual Verlet list: an OUTER list (radius rc + skin_out, built from the cell list, rare) is PRUNED to the inner list
 * (rc + skin) that the traversal walks whenever an atom moved more than skin/2.  Applies to pair forces created later. */
int amm_set_outer_skin(amm_ctx *ctx, double skin_out);
/* Tuning and test options of a context (set before the first evaluation; never read from the environment).  Names:
 * "cluster" (1: molecule rows for three-site molecules on the force-only path, 0: per-atom rows everywhere), "hybrid" (1: molecule
 * rows also for waters that share the box with other atoms, the rest through per-atom rows), "small_group" (1: interaction-group
 * forces with a set of <= 128 atoms are evaluated without a neighbour list), "rest_skin_factor" (Verlet buffer of a hybrid list's
 * per-atom part as a multiple of its molecule rows' buffer, default 2; set before amm_pair_create), "mixed_terms", "tab" (tabulated
 * force-only kernels), "site_trips", "lanes_per_row", "build_parts", "build_split" (molecule rows: the split-stream
 * list build -- a block per (cell, part), both passes shared out over its four wavefronts; 0 = off (default: not faster than the
 * one-wavefront build on the slices measured), -1 = for a rank's slice of the rows, k = k blocks per cell; the rows are the same), "unroll", "dual_unroll", "tab_block", "tab_dual_block",
 * "no_dual", "no_defer", "terms_from", "no_term_lanes", "row_phases" (1: the rows a molecule-row traversal cannot deal out in whole
 * rounds of wavefront tasks go out in smaller tasks), "group_candidates" (1: a list-free group force on a fused inner loop walks only
 * the atoms near its small set while a neighbour list of the context vouches for them), "positions_private" (1: the caller promises
 * to call amm_positions_changed after writing the bound position buffer itself; amm_run_ops then trusts the displacement checks its
 * own launches made at the end of the previous call instead of assuming that anything may have moved), "fuse_epilogue" (1: a molecule-row
 * pair kernel runs the kicks and the inner RESPA loop that follow its EVAL in the step program as its epilogue when the innermost group
 * is one bond-list set of three-site molecules -- propagators.py:933-973 unrolled; 0: launches of their own), "comm_timeout" (seconds
 * amm_check / amm_synchronize / amm_comm_destroy wait for the stream while the context owns an RCCL communicator before they abort it and
 * fail; default 0 = no deadline, the asynchronous error alone is polled).  Unknown names are an error. */
int amm_set_option(amm_ctx *ctx, const char *name, double value);
/* What amm_run_ops fused so far (statistics for tests and bench.py): out[0] = pair-kernel launches that carried the inner RESPA loop
 * of their molecules as an epilogue (the reference runs it as CustomIntegrator steps, propagators.py:933-973), out[1] = pair
 * evaluations that found their sorted copies written by the launch that moved the atoms (no gather launch), out[2] = of out[0], the
 * launches on a rank's slice that were followed by an exchange of positions and velocities instead of forces, out[3] = 0. */
int amm_run_stats(amm_ctx *ctx, int64_t out[4]);
/* amm_run_ops, resumable (several ranks whose collectives the HOST makes -- torch.distributed over gloo, or RCCL outside the library):
 * starts at op *cursor of the unrolled program (repetition * n_ops + index; 0 at first) and runs to the end (*cursor = repeat * n_ops)
 * or to the first exchanged evaluation that now waits for its exchange: the caller all-gathers the chunks of the exchange buffer,
 * calls amm_exchange_finish and calls again with the cursor it was given.  With a communicator of the library's own nothing is left
 * to the host and one call runs everything, as amm_run_ops does. */
int amm_run_ops_from(amm_ctx *ctx, const amm_op *ops, int32_t n_ops, int32_t repeat, int64_t *cursor);
/* The bound position buffer was written by the caller (needed only with option "positions_private"; harmless otherwise). */
int amm_positions_changed(amm_ctx *ctx);
/* slots of the cell-sorted order per rank (whole molecules of three: 3 ceil(ceil(n/3) / world)): the exchange buffer holds
 * world x 2 x per x 3 doubles and a chunk of a single / dual evaluation is [1 or 2][per][3] */
int amm_exchange_per(amm_ctx *ctx, int32_t *per);
/* amm_run_ops fuses KICK;MOVE;EVAL(bond-list group);KICK into one launch (bit-identical results); 0 disables. */
int amm_set_fuse_inner(amm_ctx *ctx, int32_t on);

/* ---- measurement ----------------------------------------------------------------------------- */
typedef struct {
    int64_t n_builds;       /* neighbour-list (re)builds so far                       */
    int64_t n_evals;        /* force evaluations so far                               */
    int64_t n_list_pairs;   /* directed pairs currently in the list (this slice)      */
    int64_t n_slice_atoms;  /* atoms owned by this rank's slice                       */
    int32_t capacity;       /* neighbour slots per atom                               */
    int32_t max_neighbors;  /* longest list                                           */
    int32_t lanes_per_atom; /* lanes of a wavefront that share one i-atom             */
    int32_t n_cells;
    double rlist;
    int32_t shares_list;    /* 1 if this force traverses another force's list */
    int32_t list_kind;      /* rows walked by the last evaluation: 0 one per atom, 1 one per molecule (three-site molecules only,
                               force-only evaluations: n_list_pairs then counts nine atom pairs per entry, capacity and
                               max_neighbors are molecule partners per row, lanes_per_atom is lanes per row), 2 hybrid: one per
                               three-site molecule for the pairs of two molecules + one per atom, filtered to the pairs with an
                               atom outside the molecules, for the rest (an ion, a solute, a chain next to the waters), 3 none:
                               an interaction-group force whose smaller set has <= 128 atoms is evaluated without a list */
    int64_t n_outer_builds; /* cell-based builds of the outer list (n_builds counts prunes of the inner list) */
    int64_t n_outer_pairs;
    double rlist_outer;
    double tab_error;       /* largest relative interpolation error of the force's radial Coulomb table at its check points;
                               0 without a table (analytic kernels: family without one, or a table that missed 1e-13) */
    int32_t has_table;
    int32_t rode_along;     /* 1: this force's last force-only evaluation was computed inside its list owner's launch (molecule rows:
                               the fused step-boundary pass), so it has no launch and no profile time of its own */
    int32_t has_site_table; /* 1: pairs of two Lennard-Jones sites read a radial table of their own in the molecule-row kernels (the
                               force's sites share one sigma, eps and charge: water) instead of Lennard-Jones arithmetic */
    int32_t n_rest_atoms;   /* hybrid lists: atoms outside the three-site molecules (0: none, or the force keeps per-atom rows) */
    double site_tab_error;  /* its largest relative interpolation error (bound 3e-13: the r^-14 wall) */
    int32_t n_candidates;   /* list_kind 3 on a fused inner loop: atoms within cutoff + buffer of the small set at the companion
                               list's last build (the later evaluations walk these only); 0: every evaluation walks every atom */
    int32_t n_candidate_walks;  /* evaluations that walked the candidates only */
    int32_t chargeless;         /* 1: the last force-only evaluation ran the per-atom-row kernel's instantiation without the Coulomb
                                   table (every charge of the force is zero: the Lennard-Jones fluid of BASELINE config C2) */
    int32_t build_split;        /* molecule rows: blocks per cell of the split-stream list build (a rank's slice of the rows: the four
                                   wavefronts of a block share one cell's candidate stream); 0: one wavefront per (cell, part) */
} amm_pair_stats;
int amm_pair_get_stats(amm_ctx *ctx, int32_t force_id, amm_pair_stats *out);   /* synchronises */
/* Measurement helper (bench.py): the number of directed list entries of this force with r < r_within at d_pos, counted in
 * fp64 by one launch over the neighbour rows (an evaluation must have built them).  Half of it is the number of pairs the
 * reference's expression is evaluated for when r_within is the force's cutoff (forces.py:659-661).  Synchronises. */
int amm_pair_count_within(amm_ctx *ctx, int32_t force_id, const double *d_pos, double r_within, int64_t *count);
/* Measurement helper (bench.py): molecule rows -- out[0] = lane-trips the traversal of this force executes (a wavefront walks
   64 / lanes_per_row rows until the longest is done), out[1] = row entries; their ratio is the padding of the walk.  Zeros for
   per-atom rows.  Synchronises. */
int amm_pair_row_padding(amm_ctx *ctx, int32_t force_id, int64_t out[2]);
/* Revision tag of the pair-traversal kernels: measurements kept under profiles/ carry it, and bench.py drops a stored
 * figure (the PMC traffic) once the kernels it was measured on have changed. */
const char *amm_kernel_revision(void);
/* HIP-event timing of the dominant kernel (pair traversal) on the context stream. */
/* on = 1: HIP events around every pair-kernel launch; on = -(force_id + 1): only around that force's launches
 * (two event packets per timed launch cost ~4 us of stream time each: time what you report); 0: off. */
int amm_profile_enable(amm_ctx *ctx, int32_t on);
int amm_profile_read(amm_ctx *ctx, int32_t force_id, int64_t *n_launches, double *total_ms); /* synchronises, resets */

#ifdef __cplusplus
}
#endif
#endif
