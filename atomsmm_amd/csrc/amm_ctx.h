// atomsmm_amd/csrc/amm_ctx.h -- internal structures of libatomsmm_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <string>
#include <vector>

#include "../../include/atomsmm_hip.h"

#define AMM_WAVE 64
#define AMM_MAX_PRE 6            // kicks that can ride on one inner-loop launch
#define AMM_DERIV_LAMBDA 1024   // internal PairConsts flag: energy output = dE/dlambda (softcore family)
#define AMM_MAX_GROUPS 33   // 0..31 = OpenMM force groups, 32 = all forces (`f`)

void amm_set_error(const std::string &msg);
#define AMM_HIP(call)                                                                            \
    do {                                                                                         \
        hipError_t e_ = (call);                                                                  \
        if (e_ != hipSuccess) {                                                                  \
            amm_set_error(std::string(#call) + ": " + hipGetErrorString(e_));                    \
            return 1;                                                                            \
        }                                                                                        \
    } while (0)

// Radial Coulomb table of a pair force (pair_tab.h): abscissa w = r^2 * scale, interval = (hi32(w) >> 13) - base.
struct PairTab {
    double scale, r2min;                // r2 < r2min: below the table (analytic path)
    int baseA, nA;                      // zone A (w < 1): interval = (hi32(w) >> 13) - baseA, nA intervals
    int shiftB, rawB_minus_nA, halfB;   // zone B (w >= 1): interval = (hi32(w) >> shiftB) - rawB_minus_nA
    int nint;                           // all intervals; 0: the family has no table
    int ss_first;                       // site-site table (pair_tab.h): first interval it covers; -1: none
    double ss_r2min;                    // site pairs closer than this are below it (analytic path)
};

// Constants of one pair force, precomputed on the host and passed to kernels by value.
struct PairConsts {
    int family, flags, degree, cmode;   // cmode: 0 plain coulomb, 1 ewald(erfc), 2 reaction field
    double sign, rc2, rc0, rs0, inv_dr0, rswitch, rc;
    double alpha, two_alpha_over_sqrtpi, Kc, krf, crf;
    double inv_rc0, inv_rc0_2;          // shift / force-switch constants
    double b, f12c, f6c, f1c;           // force-switch (forces.py:559-563)
    double sw_den, inv_sw_dr;           // DAMPED: rc^d - rs^d ; NONBONDED: 1/(rc - rswitch)
    double inv_sw_den, rswitch_d;       // DAMPED: 1/(rc^d - rs^d), rs^d
    PairTab tab;
};

struct Box {
    double L[3], invL[3];
};

// force rows to clear (hybrid lists: the atoms outside the molecules), served by the molecule-row chain's sort / gather launch
struct CZeroRows {
    int n;
    const int *idx;
    double *f0, *f1;
};

struct CellGrid {
    int nc[3];
    int nstencil[3];   // number of unique neighbour cells visited per axis (2h+1, or all nc cells)
    int h[3];          // stencil half-width in cells
    int ncell;
    double cw[3], inv_cw[3];
};

struct ClusterList;   // cluster.h: molecule-row list of the force-only traversal
struct SmallGroup;    // group.hip: interaction-group force with one small set, evaluated without a neighbour list

struct PairForce {
    amm_pair_desc desc;
    ClusterList *cl = nullptr;     // molecule rows (built on the first force-only evaluation of a qualifying force; list owners only)
    bool cluster_ok = false;       // the force has three-site molecules (their three pairs their only exclusions) and walks molecule rows
    // Hybrid list: molecule rows for the pairs of two such molecules + per-atom rows, kept by a hidden child force (`rest`, filtered to
    // the pairs that involve an atom outside the molecules), for everything else -- an ion, a solute, a chain next to the waters.
    bool hybrid = false;
    SmallGroup *small = nullptr;   // interaction-group force whose smaller set has <= 128 atoms (group.hip): no list at all
    std::vector<int> h_excl_ptr, h_excl_idx;    // host copy of the exclusion CSR (interaction-group forces)
    PairForce *rest = nullptr;     // the child (registered in ctx->forces behind its parent, in no group)
    bool hybrid_rest = false;      // this force IS such a child: its list keeps the pairs with at least one rest atom (code 2)
    int profile_id = -1;           // a child is timed when its parent is (amm_profile_enable with a force id): the parent's id
    std::vector<int> h_mol_first;  // first atom of every molecule; empty: molecule m = atoms 3 m .. 3 m + 2 (no rest atoms)
    int n_mol = 0, n_rest = 0;
    int *d_mol_first = nullptr, *d_rest_idx = nullptr;
    bool force_rebuild_c = false;  // the molecule rows carry site bits of another site pattern
    bool one_site_class = false;   // all atoms with eps != 0 share ONE (sigma, eps): the molecule-row kernels carry them as constants
    double site_hsig = 0, site_seps2 = 0;
    bool site_one_charge = false;  // ... and ONE charge (site_q): the site-site radial table of pair_tab.h can stand for their pairs
    double site_q = 0;
    int site_atoms = 7;            // bit a set when atom 3 m + a of some molecule m has a site (molecule-row forces)
    double *d_tab_ss = nullptr;    // site-site table [nint - ss_first][6] (molecule-row kernels); null: none
    double ss_error = 0;           // its largest relative interpolation error
    double ss_built_for[3] = {0, 0, 0};     // (sigma/2, 2 sqrt(eps), q) the tables were last built for
    const double *d_lambda_dev = nullptr;      // softcore family: lambda is read from this device scalar at launch time (amm_pair_set_lambda_dev)
    bool all_q_zero = false;       // every charge is zero (set_params): the per-atom-row kernel has an instantiation without the Coulomb table
    int last_chargeless = 0;       // ... and the last force-only evaluation ran it (statistics)
    int last_fused = 0;            // 1: the last force-only evaluation rode on the list owner's launch (molecule rows, fused pass)
    int last_kind = 0;             // list walked by the last evaluation: 0 per-atom rows, 1 molecule rows, 2 hybrid (statistics)
    PairConsts pc;
    int n = 0;
    int id = -1;                   // force id within the context
    double skin = 0, rlist = 0;
    double rlist_build = 0;        // rlist + fp32 safety margin used by the prune pass (inner list)
    double skin_out = 0;           // outer Verlet buffer (cell-built list, radius rc + skin_out)
    double rlist_out_build = 0;
    // per-atom parameters, original order: q, sigma/2, 2*sqrt(eps)
    double *d_q = nullptr, *d_hsig = nullptr, *d_seps2 = nullptr;
    int *d_excl_ptr = nullptr, *d_excl_idx = nullptr;
    CellGrid grid;
    int *d_cell_of = nullptr, *d_cell_count = nullptr, *d_cell_start = nullptr;
    int *d_cell_members = nullptr;   // [ncell][capc] atoms of each cell in arrival order (k_cell_assign)
    int capc = 0;
    int *d_perm = nullptr, *d_inv_perm = nullptr;
    double4 *d_posq_s = nullptr;   // sorted: wrapped x,y,z and charge
    double2 *d_lj_s = nullptr;     // sorted: sigma/2, 2*sqrt(eps)
    // per-atom rows: sorted copies written ahead of time by the launch that moved the atoms (pair.hip: the kernel's epilogue)
    PairForce *a_sorted_for = nullptr;
    long a_sorted_epoch = -1;
    const double *a_sorted_pos = nullptr;
    double4 *d_posq_alt = nullptr; // molecule rows: second copy of d_posq_s, for an epilogue that writes the sorted positions of the NEXT evaluation of this same force
    float4 *d_pos4f_s = nullptr;   // sorted fp32 positions at the last list build
    double *d_xref = nullptr;      // positions at the last prune (original order)
    double *d_xref_out = nullptr;  // positions at the last outer (cell-based) build
    int s_begin = 0, s_end = 0;    // sorted-slot range owned by this rank
    int cap = 0;
    int *d_nl = nullptr, *d_nnb = nullptr, *d_nnb_near = nullptr;   // inner list (traversed)
    int *d_nl_out = nullptr, *d_nnb_out = nullptr, *d_nnb_scratch = nullptr;   // outer list (pruned from)
    int cap_out = 0;
    bool dual = false;             // outer list + prune (skin_out > skin) or a single list built from the cells
    PairForce *host = nullptr;     // owner of the neighbour list this force traverses (nullptr: its own)
    int fuse_ok = -1;              // cached amm_pair_can_fuse_discount(this, host) (reset by set_params)
    int dual_ok = -1;              // cached amm_pair_can_eval_dual(this, host): -1 unknown, 0 no, 1 yes (reset by set_params)
    double rnear_build = 0;        // list radius of the guest force sharing this list (front part of each row)
    int *d_flags = nullptr;        // [0] need prune [1] overflow [2] max inner row [4] need outer build [5] max outer row
    unsigned long long *d_counters = nullptr;  // [0] prunes [1] inner pairs [2] inner front pairs [3] outer pairs [4] outer builds
    unsigned long long *d_blockstats = nullptr; // per build-kernel block: (sum, max) of list lengths
    int *d_ticket = nullptr;       // last-block tickets: [0] cell assign/scan [1] list build/statistics
    long checked_epoch = -1;       // ctx->pos_epoch / position buffer of the last displacement check
    const double *checked_pos = nullptr;
    long pre_epoch = -1;           // same, for a check already made by the kernel that moved the atoms
    const double *pre_pos = nullptr;
    int lpa = 8;                   // lanes per i-atom in the traversal kernel
    int parts = 1;                 // wavefronts per cell in the list-build kernel
    double *d_epart = nullptr;
    int n_epart = 0;
    double *d_tab = nullptr;       // radial Coulomb table: pc.tab.nint x 6 doubles (pair_tab.h)
    double tab_error = 0;          // largest relative interpolation error found when the table was built
    int *d_cls = nullptr;          // per atom (original order): 1 = no Lennard-Jones site (eps = 0) -- sorted behind the others in its cell
    int *d_cell_count_lj = nullptr, *d_cell_start_lj = nullptr;   // per cell: atoms WITH a Lennard-Jones site (count, exclusive scan)
    std::vector<int> h_cls;        // host copy of d_cls: a change of the site pattern (parameter offsets on epsilon) asks for a rebuild
    bool force_rebuild = false;    // the rows' site counts and traversal order were made for another site pattern
    int sites_match = -1;          // a guest may use its list owner's site counts only if both have sites on the same atoms (-1: unknown)
    int *d_nnb_lj = nullptr;       // per row: how many of its entries are partners with a Lennard-Jones site (front | back << 16): they come first on either side
    int *d_cell_sets = nullptr;    // interaction-group forces: per cell, which of the two sets have atoms in it (bit 0 / bit 1)
    int active_cap = 0;            // rows the pair kernels' grid covers when d_active is walked
    int active_size = 0;           // slots of d_active (long rows are filed from its front, short ones from its back)
    int *d_active = nullptr;       // filtered lists: slice-relative rows that hold entries (their number: flags[8])
    float *d_member = nullptr;     // interaction-group forces: set code of each atom (0 none, 1, 2); the list keeps only (1, 2) pairs
    int *d_row_order = nullptr;    // traversal order of the slice's rows: rows with a Lennard-Jones site first (wave-uniform LJ skip)
    bool built = false;
    int64_t n_evals = 0;
    // profiling
    std::vector<hipEvent_t> ev;    // pairs (start, stop)
    size_t ev_used = 0;
};

#ifndef AMM_TICKET_INTS
#define AMM_TICKET_INTS 64      // ints per last-block ticket (device_utils.h: amm_last_block)
#endif

struct BondedSet {
    // host staging
    std::vector<int32_t> h_idx[8];
    std::vector<double> h_par[8];
    int periodic[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    bool has_near = false;
    amm_pair_desc near_desc;
    PairConsts near_pc;
    double ewald_alpha = 0, ewald_Kc = 0;
    double ljc_Kc = 138.935456;
    bool sliced = false;           // world > 1: compute only this rank's rows (group is all-reduced by the host)
    bool finalized = false;
    // device: CSR per atom of packed (atom, term) records
    int n_terms[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int *d_ref_ptr = nullptr;      // [n+1]
    int4 *d_rec_a = nullptr;       // atoms of the term
    double4 *d_rec_q = nullptr;    // parameters + kind/role/periodic code
    int4 *d_rec_l = nullptr;       // atoms of the term as slots within their connected component
    int *d_comp_ptr = nullptr, *d_comp_atoms = nullptr;   // connected components of the term graph (CSR)
    int ncomp = 0, max_comp = 0;
    // term-parallel inner loop: every component has at most G terms (G lanes per component) -> lane l evaluates term l
    bool terms_ok = false;
    // ... and, further, every component is one three-site molecule {3 c, 3 c + 1, 3 c + 2} (slot = atom order) with at most four terms,
    // all of them harmonic bonds / angles (a box of flexible three-site water): the row owner of a molecule-row pair kernel can then
    // run the inner RESPA loop of its molecules as the kernel's epilogue (cluster.hip: cepi_rows)
    bool mol3_ok = false;
    int4 *d_term_l = nullptr;                 // [ncomp*G] atoms of the term as component slots (x < 0: no term)
    double4 *d_term_q = nullptr;              // [ncomp*G] parameters + kind/periodic code
    unsigned long long *d_atom_recs = nullptr; // [n] the atom's records as (term slot | role << 3) in 5-bit fields, count in bits 60..63
    // term-parallel evaluation of large sets (force only): every term once, its per-role forces parked in d_tf, every atom
    // then adds up its records from there in record order -- the numbers and the order of k_bonded
    int n_gterms = 0;
    int4 *d_gt_a = nullptr;        // [n_gterms] atoms of the term
    double4 *d_gt_q = nullptr;     // [n_gterms] parameters + kind/periodic code
    int *d_rec_src = nullptr;      // [nref] term * 4 + role of each (atom, term) record
    double *d_tf = nullptr;        // [n_gterms][4][3]
    // mixed sets (waters + a chain: config C5) on the fused EVAL + kicks path: the SMALL components (<= 4 atoms, <= 4 terms: a water)
    // are evaluated four lanes per component from registers / LDS (no parked forces, no gather); only the atoms and terms of the big
    // components go through the term-parallel arrays above
    bool mixed_ok = false;
    int n_sc = 0;                                   // small components
    int4 *d_sc_atoms = nullptr;                     // [n_sc] their atoms (-1 padded)
    int4 *d_sc_term_l = nullptr;                    // [n_sc*4] atoms of the component's terms as slots 0..3 (x < 0: no term)
    double4 *d_sc_term_q = nullptr;                 // [n_sc*4]
    unsigned long long *d_sc_recs = nullptr;        // [n_sc*4] per (component, atom slot): the atom's records, as d_atom_recs
    int n_big = 0, n_bigterms = 0;
    int *d_big_atoms = nullptr;                     // atoms of the other components
    int *d_big_terms = nullptr;                     // global indices of their terms (into d_gt_a / d_gt_q / d_tf)
    double *d_epart = nullptr;
    int n_epart = 0;
};

struct PmeForce;      // pme.hip
struct ConstraintSet; // constraints.hip

struct ForceObj {
    int type = 0;   // 1 pair, 2 bonded, 3 PME reciprocal space
    PairForce *pair = nullptr;
    BondedSet *bonded = nullptr;
    PmeForce *pme = nullptr;
};

struct ExprDef {
    std::vector<int32_t> code;
    std::vector<double> consts, globals;
};

// native bath step of Langevin-type integrators: v <- z v + sqrt(kT (1 - z^2) / m) * gaussian  (propagators.py:727-741)
// kind 1 (Nose-Hoover-Langevin, NHL_R_Integrator: propagators.py:1362-1449 as MassiveNoseHooverLangevin emits it): the
// three per-DOF steps  v <- v exp(-h w) ; w <- z w + sqrt(kT (1 - z^2)/Q) gaussian + (m v^2 - kT)(1 - z)/(Q friction) ;
// v <- v exp(-h w)  with the thermostat velocity w of each DOF in the per-DOF buffer `slot`.
// kind 2 (stochastic-isokinetic, SIN_R_Integrator with L = 1: propagators.py:276-355, 1045-1105): v1 <- v1 exp(-h v2) ;
// isokinetic rescale of (v, v1) ; v2 <- z v2 + sqrt(kT (1 - z^2)/Q2) gaussian + (Q1 v1^2 - kT)(1 - z)/(Q2 friction) ;
// v1 <- v1 exp(-h v2) ; rescale -- thermostat velocities v1 in the context's isokinetic slot, v2 in `slot`.
struct BathDef {
    double z, kT;
    int kind = 0;                  // 0 Ornstein-Uhlenbeck, 1 Nose-Hoover-Langevin, 2 stochastic-isokinetic
    double h = 0, Q = 0, friction = 0;
    int slot = -1;
};

// Isokinetic mode of a context (SIN(R), L = 1): every AMM_OP_KICK is the isokinetic kick
//   v <- v cosh(z) + sqrt(LkT/m) sinh(z), z = coef F / sqrt(m LkT) ;  H = sqrt(LkT / (m v^2 + Q1 v1^2 / 2)) ; v <- H v ; v1 <- H v1
// with the per-DOF thermostat velocity v1 in buffer `slot` (MassiveIsokineticPropagator, propagators.py:276-355).
struct IsoDef {
    bool on = false;
    double LkT = 0, Q1 = 0;
    int slot = -1;
};

struct GroupDef {
    int slot = -1;
    std::vector<int> forces;
    int exchange = 0;              // AMM_EXCHANGE_*: how the forces of the rank's slice reach the other ranks
};

// exchange of owner-computed force slices: the pair kernel leaves the rows of this rank's slice of the
// cell-sorted order in chunk `rank` of the exchange buffer, the chunks are all-gathered, and k_unsort spreads the
// gathered rows to the group's buffer in atom order (every rank holds the same permutation).
struct PendingExchange {
    bool active = false;
    int per = 0, nf = 0;           // slots per rank; forces per chunk (2 after a dual evaluation)
    const int *perm = nullptr;
    double *force = nullptr, *gforce = nullptr;
    // kind 1: the chunks hold the STATE of the ranks' molecules -- [x: per x 3][v: per x 3] in cell-sorted order -- after an
    // evaluation whose launch integrated them (cluster.hip: cepi_rows on a rank's slice); finishing it spreads the other ranks'
    // positions and velocities to the atom-order arrays, evaluates the lists' displacement triggers for them and writes the next
    // evaluation's sorted copies of their molecules (k_state_scatter)
    int kind = 0;
    ClusterList *cl = nullptr;
    PairForce *next = nullptr;     // the force whose sorted copies the next pair evaluation reads (or none)
};

// a neighbour list whose displacement trigger the kernel that moves the atoms evaluates (saves the check launch)
#define AMM_MAX_WATCH 4
struct ListWatch {
    const double *xref = nullptr;
    double thr2 = 0;
    int *flags = nullptr;
    long *pre_epoch = nullptr;
    const double **pre_pos = nullptr;
};

// slots per rank of the cell-sorted order: whole molecules of three (the molecule rows slice by molecule)
static inline int amm_slice_per(int n, int world) {
    const int nc = (n + 2) / 3;
    return 3 * ((nc + world - 1) / world);
}

// What follows a force-only pair evaluation in the step program when it is `[KICK ...] ; n x { KICK(f0) ; MOVE ; EVAL(group of f0) ; KICK(f0) }`
// with the innermost group one bond-list set of three-site molecules (BondedSet::mol3_ok): amm_run_ops hands it to the evaluation, and a
// molecule-row kernel runs it as its epilogue -- the wavefront that has just summed a molecule's rows kicks, moves and re-evaluates that
// molecule (no other atom enters) and writes the sorted copies of the next pair evaluation (cluster.hip: cepi_rows).
struct EpiPlan {
    BondedSet *bs = nullptr;
    double *f0 = nullptr;              // the innermost group's force buffer
    int npre = 0, niter = 0;
    const double *pre_a[AMM_MAX_PRE] = {nullptr}, *pre_b[AMM_MAX_PRE] = {nullptr};
    double pre_coef[AMM_MAX_PRE] = {0};
    int pre_plus[AMM_MAX_PRE] = {0};
    double c1 = 0, d = 0, c2 = 0;
    PairForce *next = nullptr;         // the force whose sorted copies the next pair evaluation reads (nullptr: not known)
    // kind 1: per-atom rows (pair.hip, AtomEpiArgs) -- the ops behind the EVAL are plain kicks and, maybe, a move (velocity Verlet):
    // the kicks are pre_*[0 .. npre), niter = 0
    int kind = 0;
    int with_move = 0;
    double dcoef = 0;
};

struct amm_ctx {
    int n = 0;
    int opt_spec_assign = 1;           // ... which also files the moved molecules in their cells ahead of a possible rebuild (no assign launch)
    int opt_state_exchange = 1;        // multi-rank: launches that integrate their rows' molecules exchange positions and velocities, not forces
    int opt_chargeless = 1;            // per-atom rows: the chargeless instantiation of the tabulated kernel for forces whose charges are all zero
    int opt_fuse_epilogue = 1;         // molecule rows: the inner RESPA loop of a box of three-site molecules as the pair kernel's epilogue
    const EpiPlan *epi_request = nullptr;   // amm_run_ops -> amm_cluster_eval_impl (one evaluation)
    bool epi_done = false;             // ... which says here whether its launch carried the plan (it then bumped pos_epoch itself)
    long long n_epilogues = 0;         // launches that did (statistics)
    long long n_state_exchanges = 0;   // ... on a rank's slice, followed by an exchange of positions and velocities instead of forces
    // force buffers that hold THIS RANK'S rows only (written by such launches: the kicks they feed ran on this rank's molecules,
    // nobody else needs them) until their group is evaluated again in full: amm_run_ops refuses -- or, for bond-list groups,
    // re-evaluates -- before an op reads one for all atoms
    std::vector<const double *> own_only;
    long long n_copies_current = 0;    // evaluations that found their sorted copies in place (no gather launch)
    // tuning / test options (amm_set_option): never read from the environment, so that a stray variable cannot change the
    // order of summation of a production run
    int opt_cluster = 1;           // molecule rows for qualifying forces (0: per-atom rows everywhere)
    int opt_hybrid = 1;            // ... also when the three-site molecules share the box with other atoms (hybrid lists)
    double opt_rest_skin_factor = 2.0;   // Verlet buffer of a hybrid list's per-atom part relative to its molecule rows
    int opt_mixed_terms = 1;       // mixed bond-list sets: small components four lanes each on the fused EVAL + kicks path (bonded.hip)
    int opt_small_group = 1;       // interaction-group forces with a small set (a solute) without a neighbour list (group.hip)
    bool creating_rest = false;    // amm_pair_create is making the hidden child of a hybrid list
    int opt_tab = 1;               // tabulated force-only kernels (0: the analytic kernels)
    int opt_build_split = 0;       // split-stream list build (k_cbuild_split): 0 = off, -1 = for slices (world > 1) with the library's choice of blocks per cell, k > 0 = k blocks per cell
    int opt_lpa = 0, opt_parts = 0, opt_unroll = 2, opt_dual_unroll = 2, opt_tab_bs = 0, opt_tab_dual_bs = 0;
    int opt_site_tab = 1;               // molecule rows: site-site radial tables instead of Lennard-Jones arithmetic where a force has one
    CZeroRows zero_rows = CZeroRows{0, nullptr, nullptr, nullptr};      // pending request (pair.hip -> cluster.hip, one evaluation)
    int opt_positions_private = 0;      // amm_run_ops does not assume the caller moved the atoms between calls (amm_positions_changed says so)
    int opt_group_candidates = 1;       // list-free group forces on the fused inner loop: walk the atoms near the small set only while a companion list vouches for them
    int opt_row_phases = 1;             // molecule rows: the remainder of the rows after whole rounds of tasks goes out in smaller tasks (cpair_plan)
    int opt_fuse_rows = 1;              // molecule rows: host + guest force of a shared list in ONE launch when a fused kernel exists
    int opt_no_dual = 0, opt_no_defer = 0, opt_terms_from = 8192, opt_no_term_lanes = 0;
    ListWatch watched[AMM_MAX_WATCH];
    int n_watched = 0;
    int device = 0;
    hipStream_t stream = nullptr;
    Box box;
    int rank = 0, world = 1;
    std::vector<ForceObj> forces;
    double *d_x = nullptr, *d_v = nullptr;
    const double *d_mass = nullptr;
    double *slots[AMM_MAX_SLOTS] = {nullptr};
    GroupDef groups[AMM_MAX_GROUPS];
    bool profile = false;
    int profile_only = -1;         // >= 0: time only this force id (each timed launch costs two event packets)
    double *d_scratch = nullptr;   // small scratch (reductions)
    double *d_expr_part = nullptr; // block partial sums of amm_expr_eval
    double *d_fscratch = nullptr;  // [n][3] force sink of amm_pair_energy_derivative
    ConstraintSet *constraints = nullptr;   // distance constraints of the System (AMM_OP_CONSTRAIN_*)
    std::vector<ExprDef> exprs;    // registered per-DOF expressions (AMM_OP_EXPR)
    std::vector<BathDef> baths;    // registered baths (AMM_OP_BATH)
    IsoDef iso;                    // isokinetic mode: what AMM_OP_KICK means (amm_iso_define)
    unsigned long long expr_seed = 0, expr_counter = 0;
    // ping-pong partners of x, v and the group-0 force buffer for the fused inner RESPA iteration
    double *alt_x = nullptr, *alt_v = nullptr, *alt_f = nullptr;
    bool fuse_inner = true;
    // cut the force-only pair traversal into stretches with / without Lennard-Jones arithmetic (pair.hip).  A row's order of
    // summation then depends on the rows that share its wavefront: results agree to rounding, not bit for bit, between two
    // decompositions.  AMM_SITE_TRIPS=0 (read when the context is created) switches it off -- what the bit-identity tests do.
    bool site_trips = true;
    long pos_epoch = 0;            // bumped whenever the positions may have changed (see amm_pair_eval_impl)
    double skin_out = -1.0;        // outer Verlet buffer for pair forces created afterwards (<= 0: default)
    void *comm = nullptr;          // ncclComm_t of the library's own communicator (comm.hip), or none
    bool comm_failed = false;      // ... it was aborted after an asynchronous error / a wait that timed out
    double opt_comm_timeout = 0;   // > 0: seconds a wait for the stream may last while a communicator exists before it is aborted (comm.hip:
                                   // amm_comm_wait_impl); 0: no deadline -- the asynchronous error is polled, the work queued may take any time
    double *d_xchg = nullptr;      // caller-owned exchange buffer (amm_bind_exchange): world chunks of 2 x ceil(n/world) x 3 doubles
    long long xchg_doubles = 0;
    PendingExchange pending;
    long long comm_calls = 0, comm_doubles = 0;     // collectives issued / doubles per rank they carried
};

// up to four kicks v <- v + (coef (f -/+ f2)) / m that one launch applies in order (integrate.hip: k_kicks_move; bonded.hip: the
// gather of a term-parallel bond-list evaluation carries the kicks that follow it)
struct KickList {
    const double *f[4], *f2[4];
    int plus[4];
    double coef[4];
    int n;
};

// the neighbour lists whose displacement trigger a kernel that MOVES the atoms evaluates for the positions it writes (their own
// check launch is then skipped: amm_watch_moved).  Passed to kernels by value.
// a list's flag array: [0] "an atom is farther than skin / 2 from where it was at the last build" (rebuild wanted) and, raised by the
// same checks, [AMM_FLAG_FAR] "... farther than the whole skin" (who trusted the list's reference positions with a doubled margin --
// group.hip's candidate walk -- must stop); both cleared by the rebuild
#define AMM_FLAG_FAR 12
struct WatchArgs {
    int n;
    const double *xref[AMM_MAX_WATCH];
    double thr2[AMM_MAX_WATCH];
    int *flags[AMM_MAX_WATCH];
};
// collect them from the context's pair forces (also recorded in ctx->watched) / tell them the positions they watched are current
void amm_collect_watches(amm_ctx *ctx, WatchArgs &W);
void amm_watch_moved(amm_ctx *ctx);

// comm.hip
int amm_comm_unique_id_impl(const char *rccl_path, unsigned char *out);
int amm_comm_init_impl(amm_ctx *ctx, const char *rccl_path, const unsigned char *id_bytes, int rank, int world);
int amm_comm_destroy_impl(amm_ctx *ctx);
int amm_comm_poll_impl(amm_ctx *ctx);                     // asynchronous error of the communicator -> non-zero + message
int amm_comm_wait_impl(amm_ctx *ctx, const char *who);    // bounded hipStreamSynchronize while a communicator exists
int amm_comm_allreduce_impl(amm_ctx *ctx, double *d_buf, size_t count);
int amm_comm_allgather_impl(amm_ctx *ctx, double *d_buf, size_t count_per_rank);
int amm_exchange_finish_impl(amm_ctx *ctx);

// group.hip
int amm_small_group_setup(amm_ctx *ctx, PairForce *pf, const std::vector<float> &member);
// carry_terms: the launch also evaluates the terms of this (finalized, term-parallel) bond-list set into its parked-force buffer
// own_rows: the caller can take the force's rows from a buffer of the force's own (returned here; nullptr: they are in d_force
// as usual) -- then the launch may walk the candidate atoms only (group.hip: SmallArgs); rows_unused: the caller wants the energy
// alone and d_force is scratch -- candidates again
int amm_small_group_eval_impl(amm_ctx *ctx, PairForce *pf, const double *d_pos, double *d_force, int accumulate, double *d_energy,
                              BondedSet *carry_terms = nullptr, const double **own_rows = nullptr, int rows_unused = 0);
bool amm_small_group_supported(const PairForce *pf);      // k_small_group has an instantiation for the force's family
int amm_small_group_free(SmallGroup *sg);
int amm_small_group_failed(SmallGroup *sg);
int amm_small_group_stats(SmallGroup *sg, int out[2]);
// implemented in pair.hip / cells.hip / bonded.hip / integrate.hip
int amm_pair_build_consts(const amm_pair_desc &d, PairConsts &pc);
int amm_pair_eval_impl(amm_ctx *ctx, PairForce *pf, const double *d_pos, double *d_force, int accumulate,
                       double *d_energy, PairForce *guest = nullptr, double *g_force = nullptr, int g_accumulate = 0,
                       int exchange = 0);
bool amm_pair_can_eval_dual(amm_ctx *ctx, PairForce *guest, PairForce *host);
bool amm_pair_can_fuse_discount(amm_ctx *ctx, PairForce *guest, PairForce *host);
int amm_pair_free(PairForce *pf);
int amm_pair_build_table(PairForce *pf);
int amm_pair_count_within_impl(amm_ctx *ctx, PairForce *pf, const double *d_pos, double r_within, long long *count);
const char *amm_kernel_revision_impl();
int amm_bonded_eval_impl(amm_ctx *ctx, BondedSet *bs, const double *d_pos, double *d_force, int accumulate,
                         double *d_energy);
int amm_bonded_finalize_impl(amm_ctx *ctx, BondedSet *bs);
// EVAL of a term-parallel set followed by kicks (and a move): the gather launch applies them (bit-identical to the separate ops)
int amm_bonded_eval_kicks_impl(amm_ctx *ctx, BondedSet *bs, const double *d_pos, double *d_force, int accumulate, const KickList &K,
                               int with_move, double dcoef, int terms_done = 0, const double *pair_rows = nullptr);
int amm_bonded_free(BondedSet *bs);
// (with_move: the launch also evaluates the displacement triggers of the context's lists; the caller bumps pos_epoch and calls
// amm_watch_moved)
int amm_kicks_move_impl(amm_ctx *ctx, const double *const *fa, const double *const *fb, const int *plus, const double *coef, int nk,
                        int with_move, double dcoef);
int amm_inner_components_impl(amm_ctx *ctx, BondedSet *bs, double *x, double *v, double *f0, int npre, const double *const *pre_a,
                              const double *const *pre_b, const double *pre_coef, const int *pre_plus, double c1, double d,
                              double c2, int niter, const BathDef *bath = nullptr, double d2 = 0.0);
int amm_bath_impl(amm_ctx *ctx, const BathDef &bath, double *d_v, unsigned long long counter);
int amm_isokick_impl(amm_ctx *ctx, double *d_v, const double *d_f, const double *d_f2, int plus, const double *d_mass, double coef);
int amm_constraints_create_impl(amm_ctx *ctx, const int32_t *h_pairs, const double *h_dist, int n_cons, double tol,
                                ConstraintSet **out);
int amm_constraints_save_reference(amm_ctx *ctx, ConstraintSet *cs, const double *d_x);
int amm_constrain_positions(amm_ctx *ctx, ConstraintSet *cs, double *d_x);
int amm_constrain_velocities(amm_ctx *ctx, ConstraintSet *cs, const double *d_x, double *d_v);
int amm_constraints_failed(amm_ctx *ctx, ConstraintSet *cs);
int amm_constraints_free(ConstraintSet *cs);
int amm_fused_inner_impl(amm_ctx *ctx, BondedSet *bs, const double *x_in, const double *v_in, const double *f_in,
                         double *x_out, double *v_out, double *f_out, double c1, double d, double c2);
int amm_pme_create_impl(amm_ctx *ctx, double alpha, const int *K, double Kc, const double *h_q, PmeForce **out);
int amm_pme_set_charges_impl(amm_ctx *ctx, PmeForce *pm, const double *h_q);
int amm_pme_eval_impl(amm_ctx *ctx, PmeForce *pm, const double *d_pos, double *d_force, int accumulate, double *d_energy);
int amm_pme_set_sliced_impl(PmeForce *pm, int on);
int amm_pme_free(PmeForce *pm);
int amm_expr_eval_scalar_impl(amm_ctx *ctx, const int32_t *code, int n_code, const double *consts, int n_consts, double *d_scalars,
                              int n_scalars);
int amm_expr_eval_impl(amm_ctx *ctx, const int32_t *code, int n_code, const double *consts, int n_consts, const double *globals,
                       int n_globals, unsigned long long seed, unsigned long long counter, double *d_dst, double *d_sum);
int amm_kick_impl(amm_ctx *ctx, double *d_v, const double *d_f, const double *d_f2, int plus, const double *d_mass, double coef);
int amm_combine_impl(amm_ctx *ctx, double *d_dst, const double *d_a, const double *d_b, double coef);
int amm_move_impl(amm_ctx *ctx, double *d_x, const double *d_v, double coef);
int amm_copy_impl(amm_ctx *ctx, double *d_dst, const double *d_src);
int amm_mvv_impl(amm_ctx *ctx, const double *d_v, const double *d_m, double *d_out);
int amm_reduce_add(amm_ctx *ctx, const double *d_part, int n, double scale, double *d_out);
