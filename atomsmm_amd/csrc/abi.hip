// atomsmm_amd/csrc/abi.hip -- extern "C" entry points declared in include/atomsmm_hip.h.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <numeric>

#include "amm_ctx.h"
#include "cluster.h"

int amm_pair_setup_grid(amm_ctx *ctx, PairForce *pf);
static bool amm_family_allows_cluster(int family, int flags) {
    const bool tab_family = family == AMM_NEAR_NONE || family == AMM_NEAR_SHIFT || family == AMM_NEAR_FSWITCH || family == AMM_DAMPED || family == AMM_NONBONDED;
    return tab_family && !(flags & (AMM_GROUP_LJ | AMM_GROUP_Q));
}
int amm_bonded_arity(int kind);
int amm_bonded_npar(int kind);

static thread_local std::string g_error;
void amm_set_error(const std::string &msg) { g_error = msg; }

static PairForce *get_pair(amm_ctx *ctx, int id) {
    if (!ctx || id < 0 || id >= (int)ctx->forces.size() || ctx->forces[id].type != 1) {
        amm_set_error("invalid pair force id");
        return nullptr;
    }
    return ctx->forces[id].pair;
}
static BondedSet *get_bonded(amm_ctx *ctx, int id) {
    if (!ctx || id < 0 || id >= (int)ctx->forces.size() || ctx->forces[id].type != 2) {
        amm_set_error("invalid bonded force id");
        return nullptr;
    }
    return ctx->forces[id].bonded;
}

template <typename T>
static int upload(T **dst, const T *src, size_t n) {
    AMM_HIP(hipMalloc(dst, sizeof(T) * std::max<size_t>(n, 1)));
    if (n) AMM_HIP(hipMemcpy(*dst, src, sizeof(T) * n, hipMemcpyHostToDevice));
    return 0;
}

extern "C" {

int amm_abi_version(void) { return AMM_ABI_VERSION; }
const char *amm_last_error(void) { return g_error.c_str(); }

int amm_create(int32_t n_atoms, const double h_box[3], int32_t device, void *stream, amm_ctx **out) {
    if (!out || n_atoms <= 0 || !h_box) {
        amm_set_error("amm_create: bad arguments");
        return 1;
    }
    for (int k = 0; k < 3; ++k)
        if (!(h_box[k] > 0.0)) {
            amm_set_error("amm_create: box edges must be positive (orthorhombic periodic box)");
            return 1;
        }
    int ndev = 0;
    AMM_HIP(hipGetDeviceCount(&ndev));
    if (ndev <= 0) {
        amm_set_error("amm_create: no HIP device visible (the HIP path has no CPU fallback)");
        return 1;
    }
    AMM_HIP(hipSetDevice(device));
    amm_ctx *ctx = new amm_ctx();
    ctx->n = n_atoms;
    ctx->device = device;
    ctx->stream = (hipStream_t)stream;
    for (int k = 0; k < 3; ++k) {
        ctx->box.L[k] = h_box[k];
        ctx->box.invL[k] = 1.0 / h_box[k];
    }
    AMM_HIP(hipMalloc(&ctx->d_scratch, sizeof(double) * ((n_atoms + 255) / 256 + 8)));
    *out = ctx;
    return 0;
}

int amm_destroy(amm_ctx *ctx) {
    if (!ctx) return 0;
    // (with a communicator the wait is bounded and a stuck collective is aborted: comm.hip; then the stream drains)
    amm_comm_destroy_impl(ctx);
    (void)hipStreamSynchronize(ctx->stream);
    for (auto &f : ctx->forces) {
        if (f.pair) {
            amm_pair_free(f.pair);
            delete f.pair;
        }
        if (f.bonded) {
            amm_bonded_free(f.bonded);
            delete f.bonded;
        }
        if (f.pme) amm_pme_free(f.pme);
    }
    if (ctx->d_scratch) (void)hipFree(ctx->d_scratch);
    if (ctx->d_expr_part) (void)hipFree(ctx->d_expr_part);
    if (ctx->d_fscratch) (void)hipFree(ctx->d_fscratch);
    if (ctx->constraints) amm_constraints_free(ctx->constraints);
    if (ctx->alt_x) (void)hipFree(ctx->alt_x);
    if (ctx->alt_v) (void)hipFree(ctx->alt_v);
    if (ctx->alt_f) (void)hipFree(ctx->alt_f);
    delete ctx;
    return 0;
}

int amm_set_stream(amm_ctx *ctx, void *stream) {
    ctx->stream = (hipStream_t)stream;
    return 0;
}

int amm_set_slice(amm_ctx *ctx, int32_t rank, int32_t world) {
    if (world < 1 || rank < 0 || rank >= world) {
        amm_set_error("amm_set_slice: need 0 <= rank < world");
        return 1;
    }
    for (auto &f : ctx->forces)
        if (f.pair && f.pair->built) {
            amm_set_error("amm_set_slice must be called before the first force evaluation");
            return 1;
        }
    ctx->rank = rank;
    ctx->world = world;
    return 0;
}

int amm_synchronize(amm_ctx *ctx) {
    return amm_comm_wait_impl(ctx, "amm_synchronize");         // (hipStreamSynchronize when the context has no communicator)
}

int amm_check(amm_ctx *ctx) {
    if (amm_comm_wait_impl(ctx, "amm_check")) return 1;
    if (ctx->constraints && amm_constraints_failed(ctx, ctx->constraints)) {
        amm_set_error("constraint solver did not converge (SHAKE / RATTLE, 500 iterations): time step too large or bad geometry");
        return 2;
    }
    for (size_t id = 0; id < ctx->forces.size(); ++id) {
        PairForce *pf = ctx->forces[id].pair;
        if (pf && pf->cl && pf->cl->built) {
            int cf[8];
            AMM_HIP(hipMemcpy(cf, pf->cl->d_flags, sizeof(cf), hipMemcpyDeviceToHost));
            if (cf[7]) {
                amm_set_error("molecule-row list of pair force " + std::to_string(id) + ": a cell holds more molecules than its table (" +
                              std::to_string(pf->cl->capc) + ") or a molecule stretched beyond " + std::to_string(pf->cl->rext) +
                              " nm from its first atom; forces since the last rebuild may be incomplete");
                return 2;
            }
            if (cf[1]) {
                amm_set_error("molecule-row list overflow in pair force " + std::to_string(id) + ": " + std::to_string(cf[2]) +
                              " partners > capacity " + std::to_string(pf->cl->cap) + "; forces since the last rebuild are incomplete");
                return 2;
            }
        }
        if (pf && pf->small && amm_small_group_failed(pf->small) != 0) {
            amm_set_error("interaction-group pair force " + std::to_string(id) + ": a force on an atom of the small set exceeds the range of "
                          "the fixed-point sums (1e6 kJ/mol/nm per wavefront: overlapping atoms, or NaN positions)");
            return 2;
        }
        if (!pf || !pf->built) continue;
        // (a hidden child is reported under its parent's id: the caller never saw the child's)
        const std::string who = pf->hybrid_rest ? std::to_string(pf->profile_id) + " (per-atom part of its hybrid list)" : std::to_string(id);
        int flags[16];
        AMM_HIP(hipMemcpy(flags, pf->d_flags, sizeof(flags), hipMemcpyDeviceToHost));
        if (pf->d_active && flags[8] > pf->active_cap) {
            amm_set_error("interaction-group list of pair force " + who + ": " + std::to_string(flags[8]) +
                          " rows hold entries > the " + std::to_string(pf->active_cap) +
                          " the traversal covers (more than twice the first build's); forces since the last rebuild are incomplete");
            return 2;
        }
        if (flags[7]) {
            amm_set_error("cell list overflow in pair force " + who + ": a cell holds " + std::to_string(flags[6]) +
                          " atoms > capacity " + std::to_string(pf->capc) + " (local density more than doubled since the first build)");
            return 2;
        }
        if (flags[1]) {
            amm_set_error("neighbour list overflow in pair force " + who + ": " + std::to_string(flags[2]) +
                          " neighbours > capacity " + std::to_string(pf->cap) +
                          "; forces since the last rebuild are incomplete");
            return 2;
        }
    }
    return 0;
}

int amm_pair_create(amm_ctx *ctx, const amm_pair_desc *desc, const double *h_q, const double *h_sigma,
                    const double *h_eps, const int32_t *h_excl, int32_t n_excl, double skin, int32_t *force_id) {
    if (!ctx || !desc || !h_q || !h_sigma || !h_eps || !force_id) {
        amm_set_error("amm_pair_create: null argument");
        return 1;
    }
    if (desc->family < AMM_NEAR_NONE || desc->family > AMM_LJ_VIRIAL) {
        amm_set_error("amm_pair_create: unknown family");
        return 1;
    }
    if (!(desc->rc > 0.0)) {
        amm_set_error("amm_pair_create: cutoff must be positive");
        return 1;
    }
    PairForce *pf = new PairForce();
    pf->desc = *desc;
    pf->n = ctx->n;
    if (ctx->n >= (1 << 26)) {      // the traversal addresses the sorted copies with 32-bit byte offsets (32 B per slot)
        amm_set_error("amm_pair_create: more than 2^26 atoms are not supported");
        delete pf;
        return 1;
    }
    if (amm_pair_build_consts(*desc, pf->pc) || amm_pair_build_table(pf)) {
        delete pf;
        return 1;
    }
    const int n = ctx->n;
    double Lmin = std::min(ctx->box.L[0], std::min(ctx->box.L[1], ctx->box.L[2]));
    // default Verlet buffer: 0.1 nm on one GPU (C3: list build 0.58 x 320 us per step against +30 % pair work at 0.2 nm);
    // a rank's slice makes the pair kernels 2 - 5 x cheaper but the rebuild only 1.6 - 2 x (scripts/probe_pair.py --world N),
    // so the optimum moves to a larger buffer: 0.15 nm for 2 ranks, 0.2 nm beyond
    if (skin < 0) skin = ctx->world >= 4 ? 0.2 : (ctx->world >= 2 ? 0.15 : 0.1);
    skin = std::min(skin, std::max(0.0, 0.5 * Lmin - desc->rc) * 0.999);
    pf->skin = skin;
    // a force guarded by step(rc0 - r) (the discount of FarNonbondedForce, forces.py:714; group 31 of RESPASystem) vanishes
    // beyond rc0 whatever its nominal cutoff: its list needs to reach rc0 only
    const double reach = ((desc->flags & AMM_GUARD_RC0) && desc->rc0 > 0.0) ? std::min(desc->rc, desc->rc0) : desc->rc;
    pf->rlist = reach + skin;
    pf->rlist_build = pf->rlist + 2e-4;   // fp32 build: positions carry ~1e-6 nm rounding, superset is harmless
    // outer buffer: large enough that the cell-based build is rare (hydrogens consume 0.05 nm in ~2 outer steps)
    // default: single list (skin_out = skin).  Measured at C3: the prune pass costs about as much as a cell build
    // (both are bound by L1 line-access rate / instruction issue), so the dual list only pays for slow-moving systems.
    double skin_out = ctx->skin_out > 0 ? ctx->skin_out : skin;
    skin_out = std::max(skin, std::min(skin_out, std::max(0.0, 0.5 * Lmin - desc->rc) * 0.999));
    pf->skin_out = skin_out;
    pf->rlist_out_build = reach + skin_out + 2e-4;
    if (amm_pair_setup_grid(ctx, pf)) {
        delete pf;
        return 1;
    }
    // exclusions -> symmetric CSR in original atom indices
    std::vector<int> ptr(n + 1, 0);
    for (int e = 0; e < n_excl; ++e) {
        int i = h_excl[2 * e], j = h_excl[2 * e + 1];
        if (i < 0 || j < 0 || i >= n || j >= n) {
            amm_set_error("amm_pair_create: exclusion index out of range");
            delete pf;
            return 1;
        }
        if (i == j) continue;
        ptr[i + 1]++;
        ptr[j + 1]++;
    }
    for (int i = 0; i < n; ++i) ptr[i + 1] += ptr[i];
    std::vector<int> idx(ptr[n]), fill(ptr.begin(), ptr.end() - 1);
    for (int e = 0; e < n_excl; ++e) {
        int i = h_excl[2 * e], j = h_excl[2 * e + 1];
        if (i == j) continue;
        idx[fill[i]++] = j;
        idx[fill[j]++] = i;
    }
    if (upload(&pf->d_excl_ptr, ptr.data(), ptr.size()) || upload(&pf->d_excl_idx, idx.data(), idx.size())) return 1;
    if (desc->family == AMM_SOFTCORE || (desc->flags & (AMM_GROUP_LJ | AMM_GROUP_Q))) {
        pf->h_excl_ptr = ptr;
        pf->h_excl_idx = idx;
    }
    // Three-site molecules take molecule rows on the force-only hot path (cluster.h).  A box of nothing else: molecule rows only.
    // Molecules next to other atoms (ions, a solute, a chain; at least half of the atoms in molecules): a hybrid list -- molecule
    // rows for the pairs of two molecules, per-atom rows kept by a hidden child force for every pair with an atom outside them.
    // The reference makes no such distinction (forces.py:299-312 copies any particle list, every exception -> exclusion).
    std::vector<int> rest_atoms;
    if (amm_family_allows_cluster(desc->family, desc->flags) && !ctx->creating_rest) {
        std::vector<int> mol_first;
        amm_cluster_classify(n, ptr, idx, mol_first, rest_atoms);
        pf->n_mol = (int)mol_first.size();
        pf->n_rest = (int)rest_atoms.size();
        if (pf->n_mol > 0 && pf->n_rest == 0) {
            pf->cluster_ok = true;
        } else if (pf->n_mol > 0 && 2 * 3 * (long)pf->n_mol >= (long)n) {
            pf->cluster_ok = pf->hybrid = true;
            pf->h_mol_first = mol_first;
            if (upload(&pf->d_mol_first, mol_first.data(), mol_first.size()) || upload(&pf->d_rest_idx, rest_atoms.data(), rest_atoms.size())) return 1;
        }
    }
    AMM_HIP(hipMalloc(&pf->d_q, sizeof(double) * n));
    AMM_HIP(hipMalloc(&pf->d_hsig, sizeof(double) * n));
    AMM_HIP(hipMalloc(&pf->d_seps2, sizeof(double) * n));
    const int nc = pf->grid.ncell;
    AMM_HIP(hipMalloc(&pf->d_cell_of, sizeof(int) * n));
    AMM_HIP(hipMalloc(&pf->d_cell_count, sizeof(int) * (nc + 1)));
    AMM_HIP(hipMemset(pf->d_cell_count, 0, sizeof(int) * (nc + 1)));
    AMM_HIP(hipMalloc(&pf->d_cell_start, sizeof(int) * (nc + 1)));
    AMM_HIP(hipMalloc(&pf->d_cell_count_lj, sizeof(int) * (nc + 1)));
    AMM_HIP(hipMemset(pf->d_cell_count_lj, 0, sizeof(int) * (nc + 1)));
    AMM_HIP(hipMalloc(&pf->d_cell_start_lj, sizeof(int) * (nc + 1)));
    AMM_HIP(hipMalloc(&pf->d_cls, sizeof(int) * n));
    AMM_HIP(hipMalloc(&pf->d_perm, sizeof(int) * n));
    AMM_HIP(hipMalloc(&pf->d_inv_perm, sizeof(int) * n));
    AMM_HIP(hipMalloc(&pf->d_posq_s, sizeof(double4) * n));
    AMM_HIP(hipMalloc(&pf->d_lj_s, sizeof(double2) * n));
    AMM_HIP(hipMalloc(&pf->d_pos4f_s, sizeof(float4) * n));
    AMM_HIP(hipMalloc(&pf->d_xref, sizeof(double) * 3 * n));
    AMM_HIP(hipMalloc(&pf->d_xref_out, sizeof(double) * 3 * n));
    AMM_HIP(hipMalloc(&pf->d_flags, sizeof(int) * 16));
    AMM_HIP(hipMemset(pf->d_flags, 0, sizeof(int) * 16));
    AMM_HIP(hipMalloc(&pf->d_ticket, sizeof(int) * 4 * AMM_TICKET_INTS));
    AMM_HIP(hipMemset(pf->d_ticket, 0, sizeof(int) * 4 * AMM_TICKET_INTS));
    AMM_HIP(hipMalloc(&pf->d_counters, sizeof(unsigned long long) * 8));
    AMM_HIP(hipMemset(pf->d_counters, 0, sizeof(unsigned long long) * 8));
    ForceObj fo;
    fo.type = 1;
    fo.pair = pf;
    ctx->forces.push_back(fo);
    *force_id = (int)ctx->forces.size() - 1;
    pf->id = *force_id;
    if (amm_pair_set_params(ctx, *force_id, h_q, h_sigma, h_eps)) return 1;
    if (pf->hybrid) {
        // the child: same particles, parameters, exclusions and Verlet buffer; its list keeps the pairs with a rest atom (code 2)
        int32_t child_id = -1;
        ctx->creating_rest = true;
        // (twice the Verlet buffer of the molecule rows: walking the per-atom part costs a tenth of walking the molecule rows, rebuilding
        // it half as much as rebuilding them -- at config C5 its optimum lies at a larger buffer; the two lists are independent)
        const int rc = amm_pair_create(ctx, desc, h_q, h_sigma, h_eps, h_excl, n_excl, pf->skin * ctx->opt_rest_skin_factor, &child_id);
        ctx->creating_rest = false;
        if (rc) return 1;
        PairForce *child = ctx->forces[child_id].pair;
        child->hybrid_rest = true;
        child->profile_id = pf->id;
        std::vector<float> code(n, 1.0f);
        for (int i : rest_atoms) code[i] = 2.0f;
        if (upload(&child->d_member, code.data(), code.size())) return 1;
        pf->rest = child;
    }
    return 0;
}

int amm_pair_set_lambda(amm_ctx *ctx, int32_t force_id, double value) {
    PairForce *pf = get_pair(ctx, force_id);
    if (!pf) return 1;
    if (pf->desc.family != AMM_SOFTCORE) {
        amm_set_error("amm_pair_set_lambda: not a softcore pair force");
        return 1;
    }
    pf->desc.alpha = value;
    pf->d_lambda_dev = nullptr;        // (a number from the host replaces a device-resident lambda)
    return amm_pair_build_consts(pf->desc, pf->pc);
}

int amm_pair_set_lambda_dev(amm_ctx *ctx, int32_t force_id, const double *d_lambda) {
    PairForce *pf = get_pair(ctx, force_id);
    if (!pf) return 1;
    if (pf->desc.family != AMM_SOFTCORE) {
        amm_set_error("amm_pair_set_lambda_dev: not a softcore pair force");
        return 1;
    }
    if (d_lambda && !(pf->small && ctx->opt_small_group && amm_small_group_supported(pf))) {
        amm_set_error("amm_pair_set_lambda_dev: only the list-free evaluation of a softcore force with a small set reads lambda from the device");
        return 1;
    }
    pf->d_lambda_dev = d_lambda;
    return 0;
}

int amm_pair_set_scale(amm_ctx *ctx, int32_t force_id, double scale) {
    PairForce *pf = get_pair(ctx, force_id);
    if (!pf) return 1;
    pf->desc.sign = scale;
    pf->pc.sign = scale;
    // the pairings cached for the one-pass evaluations depend on the sign (host sign == 1): decide them again
    pf->dual_ok = pf->fuse_ok = -1;
    for (auto &fo : ctx->forces)
        if (fo.type == 1 && fo.pair->host == pf) fo.pair->dual_ok = fo.pair->fuse_ok = -1;
    // (a hybrid list's per-atom part is a force of its own with its own constants: same scale)
    if (pf->rest) return amm_pair_set_scale(ctx, pf->rest->id, scale);
    return 0;
}

int amm_pair_energy_derivative(amm_ctx *ctx, int32_t force_id, const double *d_pos, double *d_out) {
    PairForce *pf = get_pair(ctx, force_id);
    if (!pf || !d_pos || !d_out) return 1;
    if (pf->desc.family != AMM_SOFTCORE) {
        amm_set_error("amm_pair_energy_derivative: only the softcore family depends on a global parameter");
        return 1;
    }
    if (!ctx->d_fscratch) AMM_HIP(hipMalloc(&ctx->d_fscratch, sizeof(double) * 3 * (size_t)ctx->n));
    if (!(ctx->opt_positions_private && d_pos == ctx->d_x)) ctx->pos_epoch++;     // (the bound buffer under the caller's promise: unchanged)
    pf->pc.flags |= AMM_DERIV_LAMBDA;
    int rc = -1;
    // (a list-free group force: only the atoms near its small set when a neighbour list vouches for them -- the rows are scratch)
    if (pf->small && ctx->opt_small_group && !pf->built) rc = amm_small_group_eval_impl(ctx, pf, d_pos, ctx->d_fscratch, 0, d_out, nullptr, nullptr, 1);
    if (rc < 0) rc = amm_pair_eval_impl(ctx, pf, d_pos, ctx->d_fscratch, 0, d_out);
    pf->pc.flags &= ~AMM_DERIV_LAMBDA;
    return rc;
}

static int share_list_pf(amm_ctx *ctx, PairForce *g, PairForce *h) {
    if (g == h || h->host || g->host || g->rnear_build > 0) {
        amm_set_error("amm_pair_share_list: invalid host/guest combination");
        return 1;
    }
    if ((g->d_member || h->d_member) && !(g->hybrid_rest && h->hybrid_rest)) {
        amm_set_error("amm_pair_share_list: an interaction-group force keeps its own (filtered) neighbour list");
        return 1;
    }
    if (g->built || h->built) {
        amm_set_error("amm_pair_share_list must be called before the first force evaluation");
        return 1;
    }
    if (g->rlist_build > h->rlist_build) {
        amm_set_error("amm_pair_share_list: the guest's list radius exceeds the host's");
        return 1;
    }
    if (h->rnear_build > 0 && h->rnear_build != g->rlist_build) {
        amm_set_error("amm_pair_share_list: the host already serves a guest with a different list radius");
        return 1;
    }
    // identical exclusion sets are required: compare the CSR arrays
    const int n = ctx->n;
    std::vector<int> pa(n + 1), pb(n + 1);
    AMM_HIP(hipMemcpy(pa.data(), g->d_excl_ptr, sizeof(int) * (n + 1), hipMemcpyDeviceToHost));
    AMM_HIP(hipMemcpy(pb.data(), h->d_excl_ptr, sizeof(int) * (n + 1), hipMemcpyDeviceToHost));
    bool same = pa == pb;
    if (same && pa[n] > 0) {
        std::vector<int> ia(pa[n]), ib(pa[n]);
        AMM_HIP(hipMemcpy(ia.data(), g->d_excl_idx, sizeof(int) * pa[n], hipMemcpyDeviceToHost));
        AMM_HIP(hipMemcpy(ib.data(), h->d_excl_idx, sizeof(int) * pa[n], hipMemcpyDeviceToHost));
        for (int i = 0; i < n && same; ++i) {
            std::sort(ia.begin() + pa[i], ia.begin() + pa[i + 1]);
            std::sort(ib.begin() + pa[i], ib.begin() + pa[i + 1]);
        }
        same = ia == ib;
    }
    if (!same) {
        amm_set_error("amm_pair_share_list: exclusion lists differ");
        return 1;
    }
    // the guest's skin must be consumed no later than the host's: same displacement trigger
    g->skin = std::min(g->skin, h->skin);
    h->skin = g->skin;
    g->skin_out = h->skin_out;
    h->rlist = h->desc.rc + h->skin;
    h->rlist_build = h->rlist + 2e-4;
    if (h->skin_out < h->skin) h->skin_out = h->skin;
    h->rlist_out_build = h->desc.rc + h->skin_out + 2e-4;
    h->rnear_build = g->rlist_build;
    g->host = h;
    // both hybrid (same exclusions: the same molecules): the per-atom parts share a list the same way
    if (g->rest && h->rest) return share_list_pf(ctx, g->rest, h->rest);
    return 0;
}

int amm_pair_share_list(amm_ctx *ctx, int32_t force_id, int32_t host_id) {
    PairForce *g = get_pair(ctx, force_id), *h = get_pair(ctx, host_id);
    if (!g || !h) return 1;
    if (g->hybrid_rest || h->hybrid_rest) {
        amm_set_error("amm_pair_share_list: not a force of the caller's");
        return 1;
    }
    return share_list_pf(ctx, g, h);
}

int amm_pair_set_params(amm_ctx *ctx, int32_t force_id, const double *h_q, const double *h_sigma, const double *h_eps) {
    PairForce *pf = get_pair(ctx, force_id);
    if (!pf) return 1;
    const int n = pf->n;
    std::vector<double> hs(n), se(n);
    for (int i = 0; i < n; ++i) {
        if (h_eps[i] < 0.0) {
            amm_set_error("amm_pair_set_params: negative epsilon");
            return 1;
        }
        hs[i] = 0.5 * h_sigma[i];          // sigma = 0.5*(sigma1+sigma2)       forces.py:256
        se[i] = 2.0 * std::sqrt(h_eps[i]); // 4*epsilon = 4*sqrt(eps1*eps2)     forces.py:257
    }
    pf->all_q_zero = true;
    for (int i = 0; i < n && pf->all_q_zero; ++i) pf->all_q_zero = h_q[i] == 0.0;
    // ordered after any kernels already queued on the stream
    AMM_HIP(hipStreamSynchronize(ctx->stream));
    AMM_HIP(hipMemcpy(pf->d_q, h_q, sizeof(double) * n, hipMemcpyHostToDevice));
    AMM_HIP(hipMemcpy(pf->d_hsig, hs.data(), sizeof(double) * n, hipMemcpyHostToDevice));
    AMM_HIP(hipMemcpy(pf->d_seps2, se.data(), sizeof(double) * n, hipMemcpyHostToDevice));
    // interaction-group forces carry the set code of every atom (0 none, 1, 2) in place of a parameter: q for SOFTCORE and
    // AMM_GROUP_LJ (product 2 = a (set 1, set 2) pair), sigma for AMM_GROUP_Q (sigma/2 holds the code).  The neighbour list
    // of such a force keeps only the pairs with code_i * code_j == 2: every other entry would be evaluated to an exact zero.
    {
        const bool by_q = pf->desc.family == AMM_SOFTCORE || (pf->desc.flags & AMM_GROUP_LJ);
        const bool by_sigma = (pf->desc.flags & AMM_GROUP_Q) != 0;
        if (by_q || by_sigma) {
            std::vector<float> member(n);
            for (int i = 0; i < n; ++i) member[i] = (float)(by_q ? h_q[i] : 0.5 * h_sigma[i]);
            if (!pf->d_member) AMM_HIP(hipMalloc(&pf->d_member, sizeof(float) * n));
            AMM_HIP(hipMemcpy(pf->d_member, member.data(), sizeof(float) * n, hipMemcpyHostToDevice));
            // a small set (a solute): no neighbour list at all (group.hip); decided again whenever the codes change
            if (!pf->built && amm_small_group_setup(ctx, pf, member)) return 1;
        }
    }
    // one Lennard-Jones site class?  (water: the oxygens) -- the molecule-row kernels then need no per-atom LJ records
    {
        // (hybrid lists: the molecule rows hold the molecules' atoms only -- the sites that matter are theirs)
        std::vector<char> in_mol;
        if (pf->hybrid) {
            in_mol.assign(n, 0);
            for (int i0 : pf->h_mol_first) in_mol[i0] = in_mol[i0 + 1] = in_mol[i0 + 2] = 1;
        }
        auto skip = [&](int i) { return h_eps[i] == 0.0 || (pf->hybrid && !in_mol[i]); };
        pf->one_site_class = true;
        bool seen = false;
        for (int i = 0; i < n; ++i) {
            if (skip(i)) continue;
            if (!seen) {
                seen = true;
                pf->site_hsig = hs[i];
                pf->site_seps2 = se[i];
            } else if (hs[i] != pf->site_hsig || se[i] != pf->site_seps2) {
                pf->one_site_class = false;
                break;
            }
        }
        if (!seen) pf->site_hsig = pf->site_seps2 = 0.0;
        // ... and one charge?  Then the pairs of two sites have ONE radial force function and a table of their own (pair_tab.h)
        pf->site_one_charge = pf->one_site_class && seen;
        pf->site_q = 0.0;
        bool first = true;
        for (int i = 0; i < n && pf->site_one_charge; ++i) {
            if (skip(i)) continue;
            if (first) {
                first = false;
                pf->site_q = h_q[i];
            } else if (h_q[i] != pf->site_q) {
                pf->site_one_charge = false;
            }
        }
        pf->site_atoms = 0;
        if (pf->hybrid) {
            for (int i0 : pf->h_mol_first)
                for (int a = 0; a < 3; ++a)
                    if (h_eps[i0 + a] != 0.0) pf->site_atoms |= 1 << a;
        } else {
            for (int i = 0; i < n; ++i)
                if (h_eps[i] != 0.0) pf->site_atoms |= 1 << (i % 3);
        }
        const double now[3] = {pf->site_hsig, pf->site_seps2, (pf->one_site_class && pf->site_one_charge) ? pf->site_q : 0.0};
        if (pf->cluster_ok && (now[0] != pf->ss_built_for[0] || now[1] != pf->ss_built_for[1] || now[2] != pf->ss_built_for[2])) {
            AMM_HIP(hipStreamSynchronize(ctx->stream));       // (kernels in flight read the tables about to be replaced)
            if (amm_pair_build_table(pf)) return 1;
            pf->dual_ok = -1;
            pf->fuse_ok = -1;
        }
    }
    // class of each atom for the traversal order: 1 = no Lennard-Jones site (its rows skip the LJ arithmetic)
    std::vector<int> cls(n);
    for (int i = 0; i < n; ++i) cls[i] = h_eps[i] == 0.0 ? 1 : 0;
    AMM_HIP(hipMemcpy(pf->d_cls, cls.data(), sizeof(int) * n, hipMemcpyHostToDevice));
    if (cls != pf->h_cls) {
        // another set of atoms has a site (an epsilon offset crossed zero): the list's order within the cells and the rows' site
        // counts were made for the old one -- rebuild at the next evaluation; guests re-check that their sites are the owner's
        if (!pf->h_cls.empty() && pf->built) pf->force_rebuild = true;
        if (!pf->h_cls.empty() && pf->cl) pf->force_rebuild_c = true;
        pf->h_cls = cls;
        pf->sites_match = -1;
        for (auto &fo : ctx->forces)
            if (fo.type == 1 && fo.pair->host == pf) fo.pair->sites_match = -1;
    }
    // dual evaluation needs bitwise equal parameters on guest and host: re-check after any change
    pf->dual_ok = pf->fuse_ok = -1;
    for (auto &fo : ctx->forces)
        if (fo.type == 1 && fo.pair->host == pf) fo.pair->dual_ok = fo.pair->fuse_ok = -1;
    if (pf->rest) return amm_pair_set_params(ctx, pf->rest->id, h_q, h_sigma, h_eps);
    return 0;
}

int amm_bonded_create(amm_ctx *ctx, int32_t *force_id) {
    BondedSet *bs = new BondedSet();
    std::memset(&bs->near_pc, 0, sizeof(bs->near_pc));
    ForceObj fo;
    fo.type = 2;
    fo.bonded = bs;
    ctx->forces.push_back(fo);
    *force_id = (int)ctx->forces.size() - 1;
    return 0;
}

int amm_bonded_add_terms(amm_ctx *ctx, int32_t force_id, int32_t kind, const int32_t *h_idx, const double *h_params,
                         int32_t n_terms, int32_t periodic, const amm_pair_desc *desc) {
    BondedSet *bs = get_bonded(ctx, force_id);
    if (!bs) return 1;
    if (bs->finalized) {
        amm_set_error("amm_bonded_add_terms after finalize");
        return 1;
    }
    if (kind < 0 || kind > 7) {
        amm_set_error("amm_bonded_add_terms: unknown kind");
        return 1;
    }
    if (!bs->h_idx[kind].empty() && bs->periodic[kind] != periodic) {
        amm_set_error("amm_bonded_add_terms: mixed periodic flags within one kind");
        return 1;
    }
    const int ar = amm_bonded_arity(kind), np = amm_bonded_npar(kind);
    if (kind == AMM_BOND_NEAR) {
        if (!desc) {
            amm_set_error("AMM_BOND_NEAR needs a pair descriptor");
            return 1;
        }
        if (bs->has_near && std::memcmp(&bs->near_desc, desc, sizeof(*desc)) != 0) {
            amm_set_error("one bonded set supports a single near-exception descriptor");
            return 1;
        }
        bs->has_near = true;
        bs->near_desc = *desc;
        if (amm_pair_build_consts(*desc, bs->near_pc)) return 1;
    }
    double scale = 1.0;
    if (kind == AMM_BOND_EWALD_EXCL) {
        if (!desc) {
            amm_set_error("AMM_BOND_EWALD_EXCL needs a descriptor carrying alpha and Kc");
            return 1;
        }
        bs->ewald_alpha = desc->alpha;
        bs->ewald_Kc = desc->Kc;
        scale = desc->Kc;
    }
    if (kind == AMM_BOND_LJC && desc) bs->ljc_Kc = desc->Kc;
    bs->periodic[kind] = periodic;
    // Terms whose energy is identically zero are not stored: the exceptions of a water model (chargeprod = 0,
    // epsilon = 0: SURVEY.md 8a-6, `tests/test_systems.py:146` expects their energy to be 0.0) would otherwise be
    // evaluated in every inner RESPA iteration -- 3 of the 6 terms of a flexible water.
    for (int t = 0; t < n_terms; ++t) {
        const double *pr = h_params + (size_t)t * np;
        bool zero = false;
        if (kind == AMM_BOND_LJC || kind == AMM_BOND_NEAR) zero = pr[0] == 0.0 && pr[2] == 0.0;
        else if (kind == AMM_BOND_EWALD_EXCL) zero = pr[0] == 0.0;
        if (zero) continue;
        bs->h_idx[kind].insert(bs->h_idx[kind].end(), h_idx + (size_t)t * ar, h_idx + (size_t)(t + 1) * ar);
        for (int k = 0; k < np; ++k) bs->h_par[kind].push_back(pr[k] * scale);
    }
    return 0;
}

int amm_bonded_finalize(amm_ctx *ctx, int32_t force_id) {
    BondedSet *bs = get_bonded(ctx, force_id);
    if (!bs) return 1;
    return amm_bonded_finalize_impl(ctx, bs);
}

int amm_bonded_set_sliced(amm_ctx *ctx, int32_t force_id, int32_t on) {
    BondedSet *bs = get_bonded(ctx, force_id);
    if (!bs) return 1;
    bs->sliced = on != 0;
    return 0;
}

// A bond-list set that the host has replaced (parameter offsets rebuild the exception terms, groups are merged anew): its
// device arrays are freed and the id stays retired -- ids are positions in the force table and are never reused.
int amm_bonded_release(amm_ctx *ctx, int32_t force_id) {
    BondedSet *bs = get_bonded(ctx, force_id);
    if (!bs) return 1;
    AMM_HIP(hipStreamSynchronize(ctx->stream));            // nothing in flight may still read it
    for (int g = 0; g < AMM_MAX_GROUPS; ++g) {
        std::vector<int> &m = ctx->groups[g].forces;
        m.erase(std::remove(m.begin(), m.end(), (int)force_id), m.end());
    }
    amm_bonded_free(bs);
    delete bs;
    ctx->forces[force_id].bonded = nullptr;
    ctx->forces[force_id].type = 0;
    return 0;
}

static int force_eval_dispatch(amm_ctx *ctx, int32_t force_id, const double *d_pos, double *d_force, int32_t accumulate,
                               double *d_energy) {
    ForceObj &f = ctx->forces[force_id];
    if (f.type == 0) {
        amm_set_error("evaluation of a released force id");
        return 1;
    }
    if (f.type == 1) return amm_pair_eval_impl(ctx, f.pair, d_pos, d_force, accumulate, d_energy);
    if (f.type == 3) return amm_pme_eval_impl(ctx, f.pme, d_pos, d_force, accumulate, d_energy);
    return amm_bonded_eval_impl(ctx, f.bonded, d_pos, d_force, accumulate, d_energy);
}

int amm_pme_create(amm_ctx *ctx, double alpha, int32_t nx, int32_t ny, int32_t nz, double Kc, const double *h_q,
                   int32_t *force_id) {
    if (!ctx || !h_q || !force_id) {
        amm_set_error("amm_pme_create: bad arguments");
        return 1;
    }
    AMM_HIP(hipSetDevice(ctx->device));
    const int K[3] = {nx, ny, nz};
    PmeForce *pm = nullptr;
    if (amm_pme_create_impl(ctx, alpha, K, Kc, h_q, &pm)) return 1;
    ForceObj fo;
    fo.type = 3;
    fo.pme = pm;
    ctx->forces.push_back(fo);
    *force_id = (int32_t)ctx->forces.size() - 1;
    return 0;
}

static PmeForce *get_pme(amm_ctx *ctx, int id) {
    if (!ctx || id < 0 || id >= (int)ctx->forces.size() || ctx->forces[id].type != 3) {
        amm_set_error("not a PME force id");
        return nullptr;
    }
    return ctx->forces[id].pme;
}

int amm_pme_set_charges(amm_ctx *ctx, int32_t force_id, const double *h_q) {
    PmeForce *pm = get_pme(ctx, force_id);
    if (!pm || !h_q) return 1;
    return amm_pme_set_charges_impl(ctx, pm, h_q);
}

int amm_pme_set_sliced(amm_ctx *ctx, int32_t force_id, int32_t on) {
    PmeForce *pm = get_pme(ctx, force_id);
    if (!pm) return 1;
    return amm_pme_set_sliced_impl(pm, on);
}

int amm_force_eval(amm_ctx *ctx, int32_t force_id, const double *d_pos, double *d_force, int32_t accumulate,
                   double *d_energy) {
    if (!ctx || force_id < 0 || force_id >= (int)ctx->forces.size() || !d_pos || !d_force) {
        amm_set_error("amm_force_eval: bad arguments");
        return 1;
    }
    if (!(ctx->opt_positions_private && d_pos == ctx->d_x)) ctx->pos_epoch++;     // a caller's positions may have changed in any way since the last call
    return force_eval_dispatch(ctx, force_id, d_pos, d_force, accumulate, d_energy);
}

int amm_positions_changed(amm_ctx *ctx) {
    if (!ctx) return 1;
    ctx->pos_epoch++;
    return 0;
}

int amm_kick(amm_ctx *ctx, double *d_v, const double *d_f, const double *d_f2, int32_t plus, const double *d_mass, double coef) {
    return amm_kick_impl(ctx, d_v, d_f, d_f2, plus, d_mass, coef);
}
int amm_move(amm_ctx *ctx, double *d_x, const double *d_v, double coef) {
    ctx->pos_epoch++;
    return amm_move_impl(ctx, d_x, d_v, coef);
}
int amm_copy(amm_ctx *ctx, double *d_dst, const double *d_src) {
    ctx->pos_epoch++;
    return amm_copy_impl(ctx, d_dst, d_src);
}
int amm_mvv(amm_ctx *ctx, const double *d_v, const double *d_m, double *d_out) { return amm_mvv_impl(ctx, d_v, d_m, d_out); }

int amm_expr_eval(amm_ctx *ctx, const int32_t *code, int32_t n_code, const double *consts, int32_t n_consts,
                  const double *globals, int32_t n_globals, uint64_t seed, uint64_t counter, double *d_dst, double *d_sum) {
    if (!ctx || !code || (n_consts > 0 && !consts) || (n_globals > 0 && !globals) || (!d_dst && !d_sum)) {
        amm_set_error("amm_expr_eval: bad arguments");
        return 1;
    }
    if (d_dst) ctx->pos_epoch++;          // (the destination may be the position buffer, or alias it)
    return amm_expr_eval_impl(ctx, code, n_code, consts, n_consts, globals, n_globals, seed, counter, d_dst, d_sum);
}

int amm_expr_eval_scalar(amm_ctx *ctx, const int32_t *code, int32_t n_code, const double *consts, int32_t n_consts, double *d_scalars,
                         int32_t n_scalars) {
    if (!ctx || !code || (n_consts > 0 && !consts) || !d_scalars) {
        amm_set_error("amm_expr_eval_scalar: bad arguments");
        return 1;
    }
    return amm_expr_eval_scalar_impl(ctx, code, n_code, consts, n_consts, d_scalars, n_scalars);
}

int amm_constraints_create(amm_ctx *ctx, const int32_t *h_pairs, const double *h_dist, int32_t n_constraints, double tolerance) {
    if (!ctx || (n_constraints > 0 && (!h_pairs || !h_dist)) || n_constraints < 0) {
        amm_set_error("amm_constraints_create: bad arguments");
        return 1;
    }
    if (ctx->constraints) {
        amm_constraints_free(ctx->constraints);
        ctx->constraints = nullptr;
    }
    AMM_HIP(hipSetDevice(ctx->device));
    return amm_constraints_create_impl(ctx, h_pairs, h_dist, n_constraints, tolerance, &ctx->constraints);
}

int amm_expr_define(amm_ctx *ctx, const int32_t *code, int32_t n_code, const double *consts, int32_t n_consts,
                    const double *globals, int32_t n_globals, int32_t *expr_id) {
    if (!ctx || !code || n_code < 1 || !expr_id || (n_consts > 0 && !consts) || (n_globals > 0 && !globals)) {
        amm_set_error("amm_expr_define: bad arguments");
        return 1;
    }
    ExprDef e;
    e.code.assign(code, code + n_code);
    if (n_consts > 0) e.consts.assign(consts, consts + n_consts);
    if (n_globals > 0) e.globals.assign(globals, globals + n_globals);
    ctx->exprs.push_back(e);
    *expr_id = (int32_t)ctx->exprs.size() - 1;
    return 0;
}

int amm_bath_define(amm_ctx *ctx, double z, double kT, int32_t *bath_id) {
    if (!ctx || !bath_id || !(z >= 0.0 && z <= 1.0) || !(kT >= 0.0)) {
        amm_set_error("amm_bath_define: need 0 <= z <= 1 and kT >= 0");
        return 1;
    }
    BathDef b;
    b.z = z;
    b.kT = kT;
    ctx->baths.push_back(b);
    *bath_id = (int32_t)ctx->baths.size() - 1;
    return 0;
}

int amm_bath_define_nhl(amm_ctx *ctx, double h, double z, double kT, double Q, double friction, int32_t slot, int32_t *bath_id) {
    if (!ctx || !bath_id || !(z >= 0.0 && z <= 1.0) || !(kT >= 0.0) || !(Q > 0.0) || !(friction > 0.0) || slot < 0 || slot >= AMM_SLOT_X) {
        amm_set_error("amm_bath_define_nhl: need 0 <= z <= 1, kT >= 0, Q > 0, friction > 0 and a per-DOF buffer slot");
        return 1;
    }
    BathDef b;
    b.z = z;
    b.kT = kT;
    b.kind = 1;
    b.h = h;
    b.Q = Q;
    b.friction = friction;
    b.slot = slot;
    ctx->baths.push_back(b);
    *bath_id = (int32_t)ctx->baths.size() - 1;
    return 0;
}

int amm_bath_define_sin(amm_ctx *ctx, double h, double z, double kT, double Q2, double friction, int32_t slot_v2, int32_t *bath_id) {
    if (!ctx || !bath_id || !(z >= 0.0 && z <= 1.0) || !(kT >= 0.0) || !(Q2 > 0.0) || !(friction > 0.0) || slot_v2 < 0 || slot_v2 >= AMM_SLOT_X) {
        amm_set_error("amm_bath_define_sin: need 0 <= z <= 1, kT >= 0, Q2 > 0, friction > 0 and a per-DOF buffer slot");
        return 1;
    }
    BathDef b;
    b.z = z;
    b.kT = kT;
    b.kind = 2;
    b.h = h;
    b.Q = Q2;
    b.friction = friction;
    b.slot = slot_v2;
    ctx->baths.push_back(b);
    *bath_id = (int32_t)ctx->baths.size() - 1;
    return 0;
}

int amm_iso_define(amm_ctx *ctx, int32_t on, double LkT, double Q1, int32_t slot_v1) {
    if (!ctx || (on && (!(LkT > 0.0) || !(Q1 > 0.0) || slot_v1 < 0 || slot_v1 >= AMM_SLOT_X))) {
        amm_set_error("amm_iso_define: need LkT > 0, Q1 > 0 and a per-DOF buffer slot");
        return 1;
    }
    ctx->iso.on = on != 0;
    ctx->iso.LkT = LkT;
    ctx->iso.Q1 = Q1;
    ctx->iso.slot = slot_v1;
    return 0;
}

int amm_comm_unique_id(const char *rccl_path, uint8_t id[AMM_COMM_ID_BYTES]) {
    if (!id) {
        amm_set_error("amm_comm_unique_id: null output");
        return 1;
    }
    return amm_comm_unique_id_impl(rccl_path, id);
}
int amm_comm_init(amm_ctx *ctx, const char *rccl_path, const uint8_t id[AMM_COMM_ID_BYTES], int32_t rank, int32_t world) {
    if (!ctx || !id || world < 1 || rank < 0 || rank >= world) {
        amm_set_error("amm_comm_init: bad arguments");
        return 1;
    }
    return amm_comm_init_impl(ctx, rccl_path, id, rank, world);
}
int amm_comm_destroy(amm_ctx *ctx) {
    if (!ctx) return 1;
    return amm_comm_destroy_impl(ctx);
}
int amm_comm_stats(amm_ctx *ctx, int64_t out[2]) {
    if (!ctx || !out) return 1;
    out[0] = ctx->comm_calls;
    out[1] = ctx->comm_doubles;
    return 0;
}
int amm_comm_allreduce(amm_ctx *ctx, double *d_buf, int64_t count) {
    if (!ctx || !d_buf || count < 0) {
        amm_set_error("amm_comm_allreduce: bad arguments");
        return 1;
    }
    return amm_comm_allreduce_impl(ctx, d_buf, (size_t)count);
}

int amm_expr_seed(amm_ctx *ctx, uint64_t seed) {
    if (!ctx) {
        amm_set_error("amm_expr_seed: null context");
        return 1;
    }
    ctx->expr_seed = seed;
    ctx->expr_counter = 0;
    return 0;
}

int amm_bind_state(amm_ctx *ctx, double *d_x, double *d_v, const double *d_mass) {
    if (!ctx || !d_x || !d_v || !d_mass) {
        amm_set_error("amm_bind_state: null argument");
        return 1;
    }
    ctx->d_x = d_x;
    ctx->d_v = d_v;
    ctx->d_mass = d_mass;
    ctx->slots[AMM_SLOT_X] = d_x;
    ctx->slots[AMM_SLOT_V] = d_v;
    return 0;
}
int amm_bind_buffer(amm_ctx *ctx, int32_t slot, double *d_buf) {
    if (slot < 0 || slot >= AMM_MAX_SLOTS) {
        amm_set_error("amm_bind_buffer: slot out of range");
        return 1;
    }
    ctx->slots[slot] = d_buf;
    return 0;
}
int amm_group_define(amm_ctx *ctx, int32_t group, int32_t slot, const int32_t *force_ids, int32_t n_forces) {
    if (!ctx || (n_forces > 0 && !force_ids) || n_forces < 0) {
        amm_set_error("amm_group_define: null argument");
        return 1;
    }
    if (group < 0 || group >= AMM_MAX_GROUPS || slot < 0 || slot >= AMM_MAX_SLOTS) {
        amm_set_error("amm_group_define: group/slot out of range");
        return 1;
    }
    for (int32_t k = 0; k < n_forces; ++k)
        if (force_ids[k] < 0 || force_ids[k] >= (int32_t)ctx->forces.size()) {
            amm_set_error("amm_group_define: unknown force id");
            return 1;
        }
    ctx->groups[group].slot = slot;
    ctx->groups[group].forces.assign(force_ids, force_ids + n_forces);
    return 0;
}

// without a communicator of its own the library cannot complete an exchanged EVAL inside a program: such an EVAL must
// be the LAST op of its amm_run_ops call; the host then gathers the chunks and calls amm_exchange_finish
static int exchange_left_to_host(amm_ctx *ctx, bool last_op) {
    if (!ctx->pending.active || last_op) return 0;
    amm_set_error("amm_run_ops: without a communicator (amm_comm_init) an exchanged EVAL must be the last op of the call");
    return 1;
}

int amm_group_set_exchange(amm_ctx *ctx, int32_t group, int32_t mode) {
    if (!ctx || group < 0 || group >= AMM_MAX_GROUPS || (mode != AMM_EXCHANGE_REDUCE && mode != AMM_EXCHANGE_GATHER)) {
        amm_set_error("amm_group_set_exchange: bad group or mode");
        return 1;
    }
    ctx->groups[group].exchange = mode;
    return 0;
}
int amm_bind_exchange(amm_ctx *ctx, double *d_buf, int64_t n_doubles) {
    if (!ctx || (n_doubles > 0 && !d_buf) || n_doubles < 0) {
        amm_set_error("amm_bind_exchange: bad arguments");
        return 1;
    }
    ctx->d_xchg = d_buf;
    ctx->xchg_doubles = n_doubles;
    return 0;
}
int amm_exchange_pending(amm_ctx *ctx, int32_t *nf) {
    if (!ctx || !nf) return 1;
    *nf = ctx->pending.active ? ctx->pending.nf : 0;
    return 0;
}
int amm_exchange_finish(amm_ctx *ctx) {
    if (!ctx) return 1;
    return amm_exchange_finish_impl(ctx);
}

int amm_run_ops(amm_ctx *ctx, const amm_op *ops, int32_t n_ops, int32_t repeat) { return amm_run_ops_from(ctx, ops, n_ops, repeat, nullptr); }

// cursor != nullptr: resumable.  Starts at op *cursor of the unrolled program (repetition * n_ops + index) and runs to its end --
// or to the first exchanged evaluation whose exchange is the HOST's to make (no communicator of the library's own): then it returns 0
// with *cursor at the op to go on from and the exchange pending (the host all-gathers the chunks, calls amm_exchange_finish and
// calls again).  *cursor == repeat * n_ops on return: the program is through.
int amm_run_ops_from(amm_ctx *ctx, const amm_op *ops, int32_t n_ops, int32_t repeat, int64_t *cursor) {
    if (!ctx->d_x || !ctx->d_v || !ctx->d_mass) {
        amm_set_error("amm_run_ops: state not bound (amm_bind_state)");
        return 1;
    }
    const long total_ops = (long)repeat * n_ops;
    if (cursor && (*cursor < 0 || *cursor > total_ops)) {
        amm_set_error("amm_run_ops_from: cursor out of range");
        return 1;
    }
    // (*cursor == total_ops: nothing left to run -- the call winds the program up: force buffers that hold this rank's rows only)
    if (ctx->pending.active) {
        amm_set_error("amm_run_ops: an exchanged evaluation still waits for amm_exchange_finish");
        return 1;
    }
    // user-visible buffers; the fused inner iteration ping-pongs between them and library-owned partners
    double *const user_x = ctx->d_x, *const user_v = ctx->d_v;
    if (!ctx->opt_positions_private) ctx->pos_epoch++;                 // the caller may have written the bound position buffer
    int f0_slot = -1;
    double *user_f0 = nullptr;
    bool swapped = false;
    // Every exit -- also the early `return 1` of a failed launch, an unbound buffer or a failed collective -- must leave the
    // context bound to the CALLER's buffers: the fused inner iteration ping-pongs d_x / d_v / the group-0 slot onto the
    // library's alt_* buffers.  On the error path the state held in the alt buffers is copied back on a best-effort basis.
    struct RestoreBindings {
        amm_ctx *ctx;
        double *ux, *uv;
        double *&uf0;
        int &slot;
        bool &swapped;
        bool done = false;
        void restore(bool copy_back) {
            if (done) return;
            done = true;
            if (slot < 0) return;
            if (swapped && copy_back) {
                const size_t bytes = sizeof(double) * 3 * (size_t)ctx->n;
                (void)hipMemcpyAsync(ux, ctx->d_x, bytes, hipMemcpyDeviceToDevice, ctx->stream);
                (void)hipMemcpyAsync(uv, ctx->d_v, bytes, hipMemcpyDeviceToDevice, ctx->stream);
                (void)hipMemcpyAsync(uf0, ctx->slots[slot], bytes, hipMemcpyDeviceToDevice, ctx->stream);
            }
            ctx->d_x = ux;
            ctx->d_v = uv;
            ctx->slots[slot] = uf0;
            ctx->slots[AMM_SLOT_X] = ux;
            ctx->slots[AMM_SLOT_V] = uv;
        }
        ~RestoreBindings() { restore(true); }
    } bindings{ctx, user_x, user_v, user_f0, f0_slot, swapped};
    // kicks that close one repetition of the program ride on the first inner-loop launch of the next (as further
    // "preceding kicks"): same order, same arithmetic, two launches less per outer step
    std::vector<amm_op> deferred;
    auto flush_deferred = [&]() -> int {
        for (const amm_op &ko : deferred) {
            double *fa = (ko.a >= 0 && ko.a < AMM_MAX_SLOTS) ? ctx->slots[ko.a] : nullptr;
            double *fb = (ko.b >= 0 && ko.b < AMM_MAX_SLOTS) ? ctx->slots[ko.b] : nullptr;
            if (!fa || (ko.b >= 0 && !fb)) {
                amm_set_error("amm_run_ops: KICK buffer not bound");
                return 1;
            }
            if (!ctx->own_only.empty()) {
                const auto has = [&](const double *b) { return b && std::find(ctx->own_only.begin(), ctx->own_only.end(), b) != ctx->own_only.end(); };
                if (has(fa) || has(fb)) {
                    amm_set_error("amm_run_ops: a deferred kick reads a force buffer that holds this rank's rows only (state exchange)");
                    return 1;
                }
            }
            if (amm_kick_impl(ctx, ctx->d_v, fa, fb, ko.c, ctx->d_mass, ko.coef)) return 1;
        }
        deferred.clear();
        return 0;
    };
    const bool no_defer = ctx->opt_no_defer != 0;      // tuning option (A/B)
    // ---- epilogue plans (cluster.hip: cepi_rows) ----
    // The ops that follow a force-only pair evaluation at op index `after` -- [KICK ...] ; n x { KICK(f0) ; MOVE ; EVAL(g0) ; KICK(f0) }
    // with g0 = one bond-list set of three-site molecules -- as a plan the evaluation's launch can carry.  When the program ENDS
    // with the kicks (the closing half kicks of an outer step) and another repetition follows, the plan goes on with the kicks and
    // the inner loop that open that repetition (`wraps`; what the deferred kicks do for the stand-alone inner-loop launch).
    // q_resume: the first op not covered (in the next repetition when wraps).
    auto slot_of = [&](int a) -> double * { return (a >= 0 && a < AMM_MAX_SLOTS) ? ctx->slots[a] : nullptr; };
    // Buffers that hold this rank's rows only (state exchange, cluster.hip).  complete(buf): before an op reads `buf` for ALL atoms --
    // a bond-list group's buffer is evaluated again (every rank can: positions are whole after every exchange; the same numbers the
    // owners hold), anything else is an error: a pair group must be evaluated again before its forces are read.
    auto forget_own_only = [&](const double *buf) {
        auto it = std::find(ctx->own_only.begin(), ctx->own_only.end(), buf);
        if (it != ctx->own_only.end()) ctx->own_only.erase(it);
    };
    auto complete = [&](const double *buf) -> int {
        if (!buf || ctx->own_only.empty() || std::find(ctx->own_only.begin(), ctx->own_only.end(), buf) == ctx->own_only.end()) return 0;
        for (int gi = 0; gi < AMM_MAX_GROUPS; ++gi) {
            GroupDef &g = ctx->groups[gi];
            if (g.slot < 0 || ctx->slots[g.slot] != buf || g.forces.empty()) continue;
            bool bonded_only = true;
            for (int fid : g.forces) bonded_only = bonded_only && ctx->forces[fid].type == 2 && !ctx->forces[fid].bonded->sliced;
            if (!bonded_only) break;
            bool first = true;
            for (int fid : g.forces) {
                if (amm_bonded_eval_impl(ctx, ctx->forces[fid].bonded, ctx->d_x, ctx->slots[g.slot], first ? 0 : 1, nullptr)) return 1;
                first = false;
            }
            forget_own_only(buf);
            return 0;
        }
        amm_set_error("amm_run_ops: an op reads a force buffer that holds this rank's rows only (state exchange) before its group was evaluated again");
        return 1;
    };
    auto plan_epilogue = [&](int after, int rep, EpiPlan &P, int &q_resume, bool &wraps) -> bool {
        if (!ctx->fuse_inner || !ctx->opt_fuse_epilogue || ctx->iso.on || swapped || f0_slot >= 0) return false;
        std::vector<amm_op> kicks;
        int j = after;
        wraps = false;
        while (true) {
            while (j < n_ops && ops[j].op == AMM_OP_KICK && (int)kicks.size() <= AMM_MAX_PRE) kicks.push_back(ops[j++]);
            if (j == n_ops && !wraps && rep + 1 < repeat && !no_defer && !kicks.empty() && ops[0].op == AMM_OP_KICK) {
                wraps = true;
                j = 0;
                continue;
            }
            break;
        }
        // the last kick of the run opens the first inner iteration
        if (kicks.empty() || j < 1 || j + 2 >= n_ops) return false;
        const int start = j - 1;
        const amm_op &k1 = ops[start];
        if (!(k1.op == AMM_OP_KICK && k1.b < 0 && ops[start + 1].op == AMM_OP_MOVE && ops[start + 2].op == AMM_OP_EVAL &&
              ops[start + 3 < n_ops ? start + 3 : start].op == AMM_OP_KICK && start + 3 < n_ops)) return false;
        const int g0 = ops[start + 2].a;
        if (g0 < 0 || g0 >= AMM_MAX_GROUPS) return false;
        GroupDef &g = ctx->groups[g0];
        if (!(g.slot == k1.a && g.forces.size() == 1 && ctx->forces[g.forces[0]].type == 2 && !g.exchange)) return false;
        BondedSet *bs = ctx->forces[g.forces[0]].bonded;
        if (!bs->mol3_ok || bs->sliced) return false;
        auto is_iter = [&](int q) {
            return q + 3 < n_ops && ops[q].op == AMM_OP_KICK && ops[q].b < 0 && ops[q].a == k1.a && ops[q].coef == k1.coef &&
                   ops[q + 1].op == AMM_OP_MOVE && ops[q + 1].coef == ops[start + 1].coef && ops[q + 2].op == AMM_OP_EVAL &&
                   ops[q + 2].a == g0 && ops[q + 3].op == AMM_OP_KICK && ops[q + 3].b < 0 && ops[q + 3].a == k1.a &&
                   ops[q + 3].coef == ops[start + 3].coef;
        };
        int niter = 0, q = start;
        while (is_iter(q)) { ++niter; q += 4; }
        const int npre = (int)kicks.size() - 1;
        if (niter < 1 || npre > AMM_MAX_PRE) return false;
        P = EpiPlan();
        P.bs = bs;
        P.f0 = ctx->slots[g.slot];
        P.npre = npre;
        P.niter = niter;
        for (int p = 0; p < npre; ++p) {
            P.pre_a[p] = slot_of(kicks[p].a);
            P.pre_b[p] = kicks[p].b >= 0 ? slot_of(kicks[p].b) : nullptr;
            if (!P.pre_a[p] || (kicks[p].b >= 0 && !P.pre_b[p])) return false;
            P.pre_coef[p] = kicks[p].coef;
            P.pre_plus[p] = kicks[p].c;
        }
        P.c1 = k1.coef;
        P.d = ops[start + 1].coef;
        P.c2 = ops[start + 3].coef;
        if (!P.f0) return false;
        // the force whose sorted copies the next pair evaluation reads: the next EVAL in program order (the list owner when it is
        // one of a pair that is evaluated in one pass)
        P.next = nullptr;
        for (int t = q, seen = 0; seen < n_ops; ++seen, ++t) {
            if (t >= n_ops) {
                if (rep + (wraps ? 2 : 1) >= repeat) break;
                t = 0;
            }
            if (ops[t].op != AMM_OP_EVAL) continue;
            const int ga = ops[t].a;
            if (ga < 0 || ga >= AMM_MAX_GROUPS || ctx->groups[ga].forces.empty() || ctx->forces[ctx->groups[ga].forces[0]].type != 1) break;
            PairForce *pa = ctx->forces[ctx->groups[ga].forces[0]].pair;
            P.next = pa;
            if (t + 1 < n_ops && ops[t + 1].op == AMM_OP_EVAL && ops[t + 1].a >= 0 && ops[t + 1].a < AMM_MAX_GROUPS &&
                !ctx->groups[ops[t + 1].a].forces.empty() && ctx->forces[ctx->groups[ops[t + 1].a].forces[0]].type == 1) {
                PairForce *pb = ctx->forces[ctx->groups[ops[t + 1].a].forces[0]].pair;
                if (pa->host == pb) P.next = pb;
                else if (pb->host == pa) P.next = pa;
            }
            break;
        }
        q_resume = q;
        return true;
    };
    // ... and for per-atom rows (pair.hip: AtomEpiArgs): `[KICK ...] [; MOVE]` behind the EVAL of a group that is one pair force -- a
    // velocity-Verlet step's closing half kick and, across the end of the repetition, the opening half kick + move of the next
    auto plan_atoms = [&](int after, int rep, EpiPlan &P, int &q_resume, bool &wraps) -> bool {
        if (!ctx->fuse_inner || !ctx->opt_fuse_epilogue || ctx->iso.on || ctx->world != 1 || swapped || f0_slot >= 0) return false;
        std::vector<amm_op> kicks;
        int j = after;
        wraps = false;
        while (true) {
            while (j < n_ops && ops[j].op == AMM_OP_KICK && (int)kicks.size() < 4) kicks.push_back(ops[j++]);
            if (j == n_ops && !wraps && rep + 1 < repeat && !no_defer && !kicks.empty() && ops[0].op == AMM_OP_KICK) {
                wraps = true;
                j = 0;
                continue;
            }
            break;
        }
        if (kicks.empty() || (j < n_ops && ops[j].op == AMM_OP_KICK)) return false;        // (a fifth kick: left to the plain path)
        const bool moves = j < n_ops && ops[j].op == AMM_OP_MOVE;
        if (!moves && !wraps && j == n_ops) return false;                                  // (closing kicks of the call's last step: as before)
        P = EpiPlan();
        P.kind = 1;
        P.npre = (int)kicks.size();
        for (int p = 0; p < P.npre; ++p) {
            P.pre_a[p] = slot_of(kicks[p].a);
            P.pre_b[p] = kicks[p].b >= 0 ? slot_of(kicks[p].b) : nullptr;
            if (!P.pre_a[p] || (kicks[p].b >= 0 && !P.pre_b[p])) return false;
            P.pre_coef[p] = kicks[p].coef;
            P.pre_plus[p] = kicks[p].c;
        }
        P.with_move = moves ? 1 : 0;
        P.dcoef = moves ? ops[j].coef : 0.0;
        q_resume = moves ? j + 1 : j;
        // the next pair evaluation's force: the next EVAL in program order
        P.next = nullptr;
        for (int t = q_resume, seen = 0; seen < n_ops; ++seen, ++t) {
            if (t >= n_ops) {
                if (rep + (wraps ? 2 : 1) >= repeat) break;
                t = 0;
            }
            if (ops[t].op != AMM_OP_EVAL) continue;
            const int ga = ops[t].a;
            if (ga >= 0 && ga < AMM_MAX_GROUPS && ctx->groups[ga].forces.size() == 1 && ctx->forces[ctx->groups[ga].forces[0]].type == 1)
                P.next = ctx->forces[ctx->groups[ga].forces[0]].pair;
            break;
        }
        return q_resume < n_ops || !wraps;        // (a wrapped plan that swallowed the whole next repetition: not a step program)
    };
    // leave to the host what only it can do: an exchange pending without a communicator of the library's own.  0: go on, 1: error,
    // 2: *cursor is set -- wind up and return
    auto leave_to_host = [&](long next_pos) -> int {
        if (!ctx->pending.active) return 0;
        if (cursor) {
            *cursor = next_pos;
            return 2;
        }
        if (next_pos == total_ops) return 0;       // (the plain entry point: the caller finishes the exchange of the program's last op)
        amm_set_error("amm_run_ops: without a communicator (amm_comm_init) an exchanged EVAL must be the last op of the call (or use amm_run_ops_from)");
        return 1;
    };
    bool yielded = false;
    int k_start = cursor ? (int)(*cursor % n_ops) : 0;
    for (int rep = cursor ? (int)(*cursor / n_ops) : 0; rep < repeat && !yielded; ++rep) {
        const int k_first = k_start;
        k_start = 0;
        for (int k = k_first; k < n_ops && !yielded; ++k) {
            const amm_op &op = ops[k];
            if (!deferred.empty() && !(k == 0 && op.op == AMM_OP_KICK) && flush_deferred()) return 1;
            // trailing block of the program = only KICKs and COPYs, and the program opens with KICKs: defer the kicks
            if (ctx->fuse_inner && !no_defer && !swapped && f0_slot < 0 && rep + 1 < repeat && k > 0 && op.op == AMM_OP_KICK &&
                deferred.empty() && ops[0].op == AMM_OP_KICK) {
                bool safe = true;
                int nk = 0;
                for (int j = k; j < n_ops && safe; ++j) {
                    if (ops[j].op == AMM_OP_KICK) ++nk;
                    else if (ops[j].op == AMM_OP_COPY) {
                        // the copy runs now, the kicks before it later: it must not feed or clobber what they read
                        if (ops[j].a >= AMM_SLOT_X || ops[j].b >= AMM_SLOT_X) safe = false;
                        for (int i = k; i < j; ++i)
                            if (ops[i].op == AMM_OP_KICK && (ops[i].a == ops[j].a || ops[i].b == ops[j].a)) safe = false;
                    } else safe = false;
                }
                if (safe && nk <= 3) {
                    for (int j = k; j < n_ops; ++j) {
                        if (ops[j].op == AMM_OP_KICK) deferred.push_back(ops[j]);
                        else {
                            double *dst = (ops[j].a >= 0 && ops[j].a < AMM_MAX_SLOTS) ? ctx->slots[ops[j].a] : nullptr;
                            double *src = (ops[j].b >= 0 && ops[j].b < AMM_MAX_SLOTS) ? ctx->slots[ops[j].b] : nullptr;
                            if (!dst || !src) {
                                amm_set_error("amm_run_ops: COPY buffer not bound");
                                return 1;
                            }
                            if (amm_copy_impl(ctx, dst, src)) return 1;
                        }
                    }
                    break;          // next repetition
                }
            }
            // component-parallel inner loop: [preceding KICKs] + n x {KICK(c1, fg) ; MOVE(d) ; EVAL(g) ; KICK(c2, fg)} in one launch
            if (ctx->fuse_inner && !swapped && op.op == AMM_OP_KICK) {
                int p = k, npre = 0;
                while (p < n_ops && ops[p].op == AMM_OP_KICK && npre < 4) { ++p; ++npre; }
                // the last KICK of the run is the first op of the inner pattern
                int start = p - 1;
                npre -= 1;
                // iteration = KICK(c1, fg) ; MOVE(d) ; [BATH(b) ; MOVE(d2) ;] EVAL(g) ; KICK(c2, fg)
                const bool bathed = start + 5 < n_ops && ops[start + 2].op == AMM_OP_BATH && ops[start + 3].op == AMM_OP_MOVE;
                const int stride = bathed ? 6 : 4, eo = bathed ? 4 : 2;       // ops per iteration / offset of the EVAL
                auto is_iter = [&](int q, const amm_op &first) {
                    if (!(q + stride - 1 < n_ops && ops[q].op == AMM_OP_KICK && ops[q].b < 0 && ops[q + 1].op == AMM_OP_MOVE &&
                          ops[q + eo].op == AMM_OP_EVAL && ops[q + eo + 1].op == AMM_OP_KICK && ops[q + eo + 1].b < 0 &&
                          ops[q + eo + 1].a == ops[q].a && ops[q].a == first.a && ops[q].coef == first.coef &&
                          ops[q + 1].coef == ops[start + 1].coef && ops[q + eo + 1].coef == ops[start + eo + 1].coef &&
                          ops[q + eo].a == ops[start + eo].a))
                        return false;
                    if (bathed)
                        return ops[q + 2].op == AMM_OP_BATH && ops[q + 2].a == ops[start + 2].a && ops[q + 2].a >= 0 &&
                               ops[q + 2].a < (int)ctx->baths.size() && ops[q + 3].op == AMM_OP_MOVE &&
                               ops[q + 3].coef == ops[start + 3].coef;
                    return true;
                };
                const int ndef = (int)deferred.size();
                if (npre <= 3 && start >= k && is_iter(start, ops[start]) && ops[start + eo].a >= 0 && ops[start + eo].a < AMM_MAX_GROUPS) {
                    GroupDef &g = ctx->groups[ops[start + eo].a];
                    BondedSet *bs = (g.slot == ops[start].a && g.forces.size() == 1 && ctx->forces[g.forces[0]].type == 2)
                                        ? ctx->forces[g.forces[0]].bonded : nullptr;
                    if (bs && bs->max_comp <= 8 && !(bs->sliced && ctx->world > 1)) {
                        int niter = 0, q = start;
                        while (is_iter(q, ops[start])) { ++niter; q += stride; }
                        const double *pa[AMM_MAX_PRE] = {nullptr}, *pb[AMM_MAX_PRE] = {nullptr};
                        double pc[AMM_MAX_PRE] = {0};
                        int pp[AMM_MAX_PRE] = {0};
                        bool ok = true;
                        for (int j = 0; j < ndef + npre; ++j) {
                            const amm_op &ko = j < ndef ? deferred[j] : ops[k + j - ndef];
                            pa[j] = (ko.a >= 0 && ko.a < AMM_MAX_SLOTS) ? ctx->slots[ko.a] : nullptr;
                            pb[j] = (ko.b >= 0 && ko.b < AMM_MAX_SLOTS) ? ctx->slots[ko.b] : nullptr;
                            pc[j] = ko.coef;
                            pp[j] = ko.c;
                            if (!pa[j] || (ko.b >= 0 && !pb[j])) ok = false;
                        }
                        double *f0 = ctx->slots[g.slot];
                        if (ok && f0) {
                            if (complete(f0)) return 1;
                            for (int j = 0; j < ndef + npre; ++j)
                                if (complete(pa[j]) || complete(pb[j])) return 1;
                            deferred.clear();
                            if (amm_inner_components_impl(ctx, bs, ctx->d_x, ctx->d_v, f0, ndef + npre, pa, pb, pc, pp, ops[start].coef,
                                                          ops[start + 1].coef, ops[start + eo + 1].coef, niter,
                                                          bathed ? &ctx->baths[ops[start + 2].a] : nullptr,
                                                          bathed ? ops[start + 3].coef : 0.0)) return 1;
                            ctx->pos_epoch++;
                            amm_watch_moved(ctx);
                            k = q - 1;
                            continue;
                        }
                    }
                }
                if (!deferred.empty()) {
                    // no component launch to ride on: the deferred kicks can still lead the launch of the run of plain kicks (+ move)
                    // that opens this repetition (the block further down) -- a velocity-Verlet step is then KICK + KICK + MOVE in one
                    int run = 0;
                    for (int j = k; j < n_ops && ops[j].op == AMM_OP_KICK; ++j) ++run;
                    if (ctx->iso.on || (int)deferred.size() + run > 4) {
                        if (flush_deferred()) return 1;
                    }
                }
            }
            // fused inner RESPA iteration: KICK(c1, fg) ; MOVE(d) ; EVAL(g) ; KICK(c2, fg) with g = one bond-list set
            // (kicks deferred from the previous repetition must not be overtaken by this block's move: deferral requires
            // f0_slot < 0, i.e. that this block never matched -- flushed here all the same, so that the order does not rest on that)
            if (!deferred.empty() && ctx->fuse_inner && !ctx->iso.on && op.op == AMM_OP_KICK && op.b < 0 && k + 3 < n_ops &&
                ops[k + 1].op == AMM_OP_MOVE && ops[k + 2].op == AMM_OP_EVAL && flush_deferred()) return 1;
            if (ctx->fuse_inner && !ctx->iso.on && op.op == AMM_OP_KICK && op.b < 0 && k + 3 < n_ops && ops[k + 1].op == AMM_OP_MOVE &&
                ops[k + 2].op == AMM_OP_EVAL && ops[k + 3].op == AMM_OP_KICK && ops[k + 3].b < 0 && ops[k + 3].a == op.a &&
                ops[k + 2].a >= 0 && ops[k + 2].a < AMM_MAX_GROUPS) {
                GroupDef &g = ctx->groups[ops[k + 2].a];
                if (g.slot == op.a && g.forces.size() == 1 && ctx->forces[g.forces[0]].type == 2 &&
                    !(ctx->forces[g.forces[0]].bonded->sliced && ctx->world > 1) && (f0_slot < 0 || f0_slot == g.slot)) {
                    BondedSet *bs = ctx->forces[g.forces[0]].bonded;
                    if (!ctx->alt_x) {
                        const size_t bytes = sizeof(double) * 3 * (size_t)ctx->n;
                        AMM_HIP(hipMalloc(&ctx->alt_x, bytes));
                        AMM_HIP(hipMalloc(&ctx->alt_v, bytes));
                        AMM_HIP(hipMalloc(&ctx->alt_f, bytes));
                    }
                    if (f0_slot < 0) {
                        f0_slot = g.slot;
                        user_f0 = ctx->slots[f0_slot];
                    }
                    double *xi = ctx->d_x, *vi = ctx->d_v, *fi = ctx->slots[f0_slot];
                    double *xo = swapped ? user_x : ctx->alt_x, *vo = swapped ? user_v : ctx->alt_v,
                           *fo = swapped ? user_f0 : ctx->alt_f;
                    if (amm_fused_inner_impl(ctx, bs, xi, vi, fi, xo, vo, fo, op.coef, ops[k + 1].coef, ops[k + 3].coef)) return 1;
                    ctx->pos_epoch++;
                    ctx->d_x = xo;
                    ctx->d_v = vo;
                    ctx->slots[f0_slot] = fo;
                    ctx->slots[AMM_SLOT_X] = xo;
                    ctx->slots[AMM_SLOT_V] = vo;
                    swapped = !swapped;
                    k += 3;
                    continue;
                }
            }
            // EVAL(ga) ; EVAL(gb) of a guest pair force and the owner of its list, same positions: one pass for both
            const bool no_dual = ctx->opt_no_dual != 0;     // tuning option
            if (ctx->fuse_inner && !no_dual && op.op == AMM_OP_EVAL && k + 1 < n_ops && ops[k + 1].op == AMM_OP_EVAL && op.a >= 0 &&
                op.a < AMM_MAX_GROUPS && ops[k + 1].a >= 0 && ops[k + 1].a < AMM_MAX_GROUPS && op.a != ops[k + 1].a) {
                GroupDef &g1 = ctx->groups[op.a], &g2 = ctx->groups[ops[k + 1].a];
                // further members of the two groups (bond-list terms, reciprocal space of a PME outer force) are added after
                // the shared pass; they must not be pair forces themselves
                auto tail_ok = [&](const GroupDef &g) {
                    for (size_t j = 1; j < g.forces.size(); ++j)
                        if (ctx->forces[g.forces[j]].type == 1) return false;
                    return true;
                };
                if (!g1.forces.empty() && !g2.forces.empty() && tail_ok(g1) && tail_ok(g2) && g1.slot >= 0 && g2.slot >= 0 &&
                    ctx->slots[g1.slot] && ctx->slots[g2.slot] && ctx->forces[g1.forces[0]].type == 1 &&
                    ctx->forces[g2.forces[0]].type == 1) {
                    PairForce *pa = ctx->forces[g1.forces[0]].pair, *pb = ctx->forces[g2.forces[0]].pair;
                    PairForce *guest = pa->host == pb ? pa : (pb->host == pa ? pb : nullptr);
                    if (guest) {
                        PairForce *host = guest->host;
                        double *fg = ctx->slots[guest == pa ? g1.slot : g2.slot], *fh = ctx->slots[guest == pa ? g2.slot : g1.slot];
                        if (g1.exchange == g2.exchange && amm_pair_can_eval_dual(ctx, guest, host)) {
                            // the kicks and the inner loop that follow as the launch's epilogue, when both groups are the pair forces alone
                            EpiPlan plan;
                            int q_resume = 0;
                            bool wraps = false;
                            // (several ranks: only groups whose exchange is the all-gather of slices -- the launch then integrates
                            // this rank's molecules and the ranks exchange positions and velocities: cluster.hip, state exchange)
                            const bool planned = (ctx->world == 1 ? !g1.exchange : g1.exchange == AMM_EXCHANGE_GATHER) &&
                                                 g1.forces.size() == 1 && g2.forces.size() == 1 && plan_epilogue(k + 2, rep, plan, q_resume, wraps);
                            ctx->epi_request = planned ? &plan : nullptr;
                            ctx->epi_done = false;
                            const int rc_dual = amm_pair_eval_impl(ctx, host, ctx->d_x, fh, 0, nullptr, guest, fg, 0, g1.exchange);
                            ctx->epi_request = nullptr;
                            if (rc_dual) return 1;
                            if (ctx->epi_done) {
                                ctx->epi_done = false;
                                const int lv = leave_to_host(wraps ? (long)(rep + 1) * n_ops + q_resume : (long)rep * n_ops + q_resume);
                                if (lv == 1) return 1;
                                if (lv == 2) {
                                    yielded = true;
                                    break;
                                }
                                if (wraps) {
                                    k_start = q_resume;
                                    break;
                                }
                                k = q_resume - 1;
                                continue;
                            }
                            forget_own_only(fh);           // (both buffers are written in full: by the kernel, or by the exchange's unsort)
                            forget_own_only(fg);
                            {
                                const int lv = leave_to_host((long)rep * n_ops + k + 2);
                                if (lv == 1) return 1;
                                if (lv == 2) {
                                    // (further members of the two groups are added after the exchange: only pair-only groups get here --
                                    // an exchanged group holds exactly one pair force)
                                    yielded = true;
                                    break;
                                }
                            }
                            for (const GroupDef *g : {&g1, &g2})
                                for (size_t j = 1; j < g->forces.size(); ++j)
                                    if (force_eval_dispatch(ctx, g->forces[j], ctx->d_x, ctx->slots[g->slot], 1, nullptr)) return 1;
                            k += 1;
                            continue;
                        }
                    }
                }
            }
            // EVAL(g) ; KICK ... [; MOVE] with g = a term-parallel bond-list set [+ an interaction-group pair force with a small set,
            // which group.hip evaluates without a list and which writes EVERY row]: the pair force goes first, the terms are
            // evaluated, and the launch that gathers their forces also applies the kicks and the move that follow -- an inner RESPA
            // iteration of a system that is not pure water (config C5: chain + solute + waters) is then 3 launches, not 8.
            if (ctx->fuse_inner && !ctx->iso.on && op.op == AMM_OP_EVAL && op.a >= 0 && op.a < AMM_MAX_GROUPS && ctx->groups[op.a].slot >= 0 &&
                !ctx->groups[op.a].exchange && k + 1 < n_ops && ops[k + 1].op == AMM_OP_KICK) {
                GroupDef &g = ctx->groups[op.a];
                BondedSet *bs = nullptr;
                PairForce *ps = nullptr;
                bool plain = g.forces.size() >= 1 && g.forces.size() <= 2;
                for (int fid : g.forces) {
                    ForceObj &fo = ctx->forces[fid];
                    if (fo.type == 2 && !bs) bs = fo.bonded;
                    else if (fo.type == 1 && !ps && fo.pair->small && ctx->opt_small_group && !fo.pair->built && amm_small_group_supported(fo.pair)) ps = fo.pair;
                    else plain = false;
                }
                double *buf = ctx->slots[g.slot];
                if (plain && bs && buf && bs->n_gterms > 0 && !(bs->sliced && ctx->world > 1) && ctx->world == 1) {
                    KickList K;
                    K.n = 0;
                    int j = k + 1;
                    bool bound = true;
                    while (j < n_ops && K.n < 4 && ops[j].op == AMM_OP_KICK) {
                        const amm_op &kick = ops[j];
                        const double *a_ = (kick.a >= 0 && kick.a < AMM_MAX_SLOTS) ? ctx->slots[kick.a] : nullptr;
                        const double *b_ = (kick.b >= 0 && kick.b < AMM_MAX_SLOTS) ? ctx->slots[kick.b] : nullptr;
                        if (!a_ || (kick.b >= 0 && !b_)) {
                            bound = false;
                            break;
                        }
                        K.f[K.n] = a_;
                        K.f2[K.n] = b_;
                        K.plus[K.n] = kick.c;
                        K.coef[K.n] = kick.coef;
                        ++K.n;
                        ++j;
                    }
                    for (int q = K.n; q < 4; ++q) {
                        K.f[q] = K.f2[q] = nullptr;
                        K.plus[q] = 0;
                        K.coef[q] = 0.0;
                    }
                    const bool more_kicks = j < n_ops && ops[j].op == AMM_OP_KICK;      // a fifth kick: left to the next launch
                    const bool moves = !more_kicks && j < n_ops && ops[j].op == AMM_OP_MOVE;
                    if (bound && K.n >= 1) {
                        // (the pair force's launch evaluates the bond-list terms too: group.hip, TermsWork)
                        const double *pair_rows = nullptr;         // (the pair force's rows: in `buf`, or in a buffer of its own)
                        const bool own = bs->mixed_ok && ctx->opt_mixed_terms;       // (k_mixed_eval_kicks takes them from anywhere)
                        if (ps && amm_small_group_eval_impl(ctx, ps, ctx->d_x, buf, 0, nullptr, bs, own ? &pair_rows : nullptr) != 0) return 1;
                        if (amm_bonded_eval_kicks_impl(ctx, bs, ctx->d_x, buf, ps ? 1 : 0, K, moves ? 1 : 0, moves ? ops[j].coef : 0.0, ps ? 1 : 0, pair_rows)) return 1;
                        if (moves) {
                            ctx->pos_epoch++;
                            amm_watch_moved(ctx);
                        }
                        k = j - (moves ? 0 : 1);
                        continue;
                    }
                }
            }
            // a run of plain kicks, then (maybe) a move: one launch (same arithmetic per degree of freedom, same order)
            if (ctx->fuse_inner && !ctx->iso.on && op.op == AMM_OP_KICK) {
                const double *fa[4], *fb[4];
                int plus[4], nk = 0;
                double coef[4];
                // (kicks deferred from the end of the previous repetition come first: the order they were written in)
                const int ndef = (int)deferred.size();
                bool def_ok = ndef <= 4;
                for (int q = 0; q < ndef && def_ok; ++q) {
                    const amm_op &kick = deferred[q];
                    const double *a_ = (kick.a >= 0 && kick.a < AMM_MAX_SLOTS) ? ctx->slots[kick.a] : nullptr;
                    const double *b_ = (kick.b >= 0 && kick.b < AMM_MAX_SLOTS) ? ctx->slots[kick.b] : nullptr;
                    if (!a_ || (kick.b >= 0 && !b_)) {
                        def_ok = false;
                        break;
                    }
                    fa[nk] = a_;
                    fb[nk] = b_;
                    plus[nk] = kick.c;
                    coef[nk] = kick.coef;
                    ++nk;
                }
                if (ndef > 0 && !def_ok) {
                    if (flush_deferred()) return 1;         // (reports the unbound buffer)
                    nk = 0;
                }
                const int nlead = nk;
                int j = k;
                while (j < n_ops && nk < 4 && ops[j].op == AMM_OP_KICK) {
                    const amm_op &kick = ops[j];
                    const double *a_ = (kick.a >= 0 && kick.a < AMM_MAX_SLOTS) ? ctx->slots[kick.a] : nullptr;
                    const double *b_ = (kick.b >= 0 && kick.b < AMM_MAX_SLOTS) ? ctx->slots[kick.b] : nullptr;
                    if (!a_ || (kick.b >= 0 && !b_)) break;                 // left to the plain path, which reports it
                    fa[nk] = a_;
                    fb[nk] = b_;
                    plus[nk] = kick.c;
                    coef[nk] = kick.coef;
                    ++nk;
                    ++j;
                }
                const bool moves = j < n_ops && ops[j].op == AMM_OP_MOVE;
                const bool whole_run = !(j < n_ops && ops[j].op == AMM_OP_KICK);       // (a fifth kick: the run goes on)
                if (nk - nlead == j - k && (nlead == 0 || whole_run) && (nk >= 2 || (nk == 1 && moves))) {
                    for (int q = 0; q < nk; ++q)
                        if (complete(fa[q]) || complete(fb[q])) return 1;
                    if (amm_kicks_move_impl(ctx, fa, fb, plus, coef, nk, moves ? 1 : 0, moves ? ops[j].coef : 0.0)) return 1;
                    deferred.clear();
                    if (moves) {
                        ctx->pos_epoch++;
                        amm_watch_moved(ctx);
                    }
                    k = j - (moves ? 0 : 1);
                    continue;
                }
                if (!deferred.empty() && flush_deferred()) return 1;       // (not taken along: before anything else, in their order)
            }
            switch (op.op) {
            case AMM_OP_EVAL: {
                if (op.a < 0 || op.a >= AMM_MAX_GROUPS || ctx->groups[op.a].slot < 0) {
                    amm_set_error("amm_run_ops: EVAL of an undefined group");
                    return 1;
                }
                GroupDef &g = ctx->groups[op.a];
                double *buf = ctx->slots[g.slot];
                if (!buf) {
                    amm_set_error("amm_run_ops: group buffer not bound");
                    return 1;
                }
                if (g.forces.empty()) AMM_HIP(hipMemsetAsync(buf, 0, sizeof(double) * 3 * (size_t)ctx->n, ctx->stream));
                if ((ctx->world == 1 ? !g.exchange : g.exchange == AMM_EXCHANGE_GATHER) && g.forces.size() == 1 &&
                    ctx->forces[g.forces[0]].type == 1) {
                    // one pair force: the kicks and the inner loop that follow can ride on its launch (molecule rows: cepi_rows; several
                    // ranks: followed by an exchange of positions and velocities instead of forces)
                    EpiPlan plan;
                    int q_resume = 0;
                    bool wraps = false;
                    if (plan_epilogue(k + 1, rep, plan, q_resume, wraps) ||
                        (ctx->forces[g.forces[0]].pair->all_q_zero && !ctx->forces[g.forces[0]].pair->cluster_ok &&
                         plan_atoms(k + 1, rep, plan, q_resume, wraps))) {
                        ctx->epi_request = &plan;
                        ctx->epi_done = false;
                        const int rc_one = amm_pair_eval_impl(ctx, ctx->forces[g.forces[0]].pair, ctx->d_x, buf, 0, nullptr, nullptr, nullptr, 0, g.exchange);
                        ctx->epi_request = nullptr;
                        if (rc_one) return 1;
                        if (ctx->epi_done) {
                            ctx->epi_done = false;
                            const int lv = leave_to_host(wraps ? (long)(rep + 1) * n_ops + q_resume : (long)rep * n_ops + q_resume);
                            if (lv == 1) return 1;
                            if (lv == 2) yielded = true;
                            else if (wraps) {
                                k_start = q_resume;
                                k = n_ops;          // (leaves the loop over this repetition's ops)
                            } else {
                                k = q_resume - 1;
                            }
                        } else {
                            forget_own_only(buf);
                            const int lv = leave_to_host((long)rep * n_ops + k + 1);
                            if (lv == 1) return 1;
                            if (lv == 2) yielded = true;
                        }
                        break;
                    }
                }
                forget_own_only(buf);           // (every path below writes the group's buffer in full)
                if (g.exchange) {
                    if (g.forces.size() != 1 || ctx->forces[g.forces[0]].type != 1) {
                        amm_set_error("amm_run_ops: an exchanged group must hold exactly one pair force");
                        return 1;
                    }
                    if (amm_pair_eval_impl(ctx, ctx->forces[g.forces[0]].pair, ctx->d_x, buf, 0, nullptr, nullptr, nullptr, 0, 1)) return 1;
                    const int lv = leave_to_host((long)rep * n_ops + k + 1);
                    if (lv == 1) return 1;
                    if (lv == 2) yielded = true;
                    break;
                }
                // FarNonbondedForce (forces.py:710-724) = total + discount, two forces of one group: when the discount
                // (guarded near force, sign -1) shares the total's neighbour list, both are evaluated in ONE traversal that
                // accumulates into the same buffer (the reference, and OpenMM, run two passes)
                std::vector<char> done(g.forces.size(), 0);
                bool first = true;
                for (size_t j = 0; j < g.forces.size(); ++j) {
                    if (done[j]) continue;
                    ForceObj &fj = ctx->forces[g.forces[j]];
                    if (fj.type == 1) {
                        size_t partner = g.forces.size();
                        for (size_t i = 0; i < g.forces.size() && partner == g.forces.size(); ++i)
                            if (i != j && !done[i] && ctx->forces[g.forces[i]].type == 1 &&
                                amm_pair_can_fuse_discount(ctx, ctx->forces[g.forces[i]].pair, fj.pair)) partner = i;
                        if (partner < g.forces.size()) {
                            if (amm_pair_eval_impl(ctx, fj.pair, ctx->d_x, buf, first ? 0 : 1, nullptr, ctx->forces[g.forces[partner]].pair, buf, 1, 0)) return 1;
                            done[j] = done[partner] = 1;
                            first = false;
                            continue;
                        }
                        // the discount itself comes later in the list: let its host pick it up
                        bool is_discount = false;
                        for (size_t i = 0; i < g.forces.size(); ++i)
                            if (i != j && !done[i] && ctx->forces[g.forces[i]].type == 1 && amm_pair_can_fuse_discount(ctx, fj.pair, ctx->forces[g.forces[i]].pair)) is_discount = true;
                        if (is_discount) continue;
                    }
                    if (force_eval_dispatch(ctx, g.forces[j], ctx->d_x, buf, first ? 0 : 1, nullptr)) return 1;
                    done[j] = 1;
                    first = false;
                }
                for (size_t j = 0; j < g.forces.size(); ++j)      // (a discount whose host was consumed by another pairing)
                    if (!done[j]) {
                        if (force_eval_dispatch(ctx, g.forces[j], ctx->d_x, buf, first ? 0 : 1, nullptr)) return 1;
                        first = false;
                    }
            } break;
            case AMM_OP_KICK: {
                double *fa = (op.a >= 0 && op.a < AMM_MAX_SLOTS) ? ctx->slots[op.a] : nullptr;
                double *fb = (op.b >= 0 && op.b < AMM_MAX_SLOTS) ? ctx->slots[op.b] : nullptr;
                if (!fa || (op.b >= 0 && !fb)) {
                    amm_set_error("amm_run_ops: KICK buffer not bound");
                    return 1;
                }
                if (complete(fa) || complete(fb)) return 1;
                if (amm_kick_impl(ctx, ctx->d_v, fa, fb, op.c, ctx->d_mass, op.coef)) return 1;
            } break;
            case AMM_OP_MOVE:
                if (amm_move_impl(ctx, ctx->d_x, ctx->d_v, op.coef)) return 1;
                ctx->pos_epoch++;
                break;
            case AMM_OP_COPY: {
                double *dst = (op.a >= 0 && op.a < AMM_MAX_SLOTS) ? ctx->slots[op.a] : nullptr;
                double *src = (op.b >= 0 && op.b < AMM_MAX_SLOTS) ? ctx->slots[op.b] : nullptr;
                if (!dst || !src) {
                    amm_set_error("amm_run_ops: COPY buffer not bound");
                    return 1;
                }
                if (complete(src)) return 1;
                if (amm_copy_impl(ctx, dst, src)) return 1;
                if (dst == ctx->d_x) ctx->pos_epoch++;
            } break;
            case AMM_OP_COMBINE: {
                double *dst = (op.a >= 0 && op.a < AMM_MAX_SLOTS) ? ctx->slots[op.a] : nullptr;
                double *sa = (op.b >= 0 && op.b < AMM_MAX_SLOTS) ? ctx->slots[op.b] : nullptr;
                double *sb = (op.c >= 0 && op.c < AMM_MAX_SLOTS) ? ctx->slots[op.c] : nullptr;
                if (!dst || !sa || !sb) {
                    amm_set_error("amm_run_ops: COMBINE buffer not bound");
                    return 1;
                }
                if (complete(sa) || complete(sb)) return 1;
                if (amm_combine_impl(ctx, dst, sa, sb, op.coef)) return 1;
                if (dst == ctx->d_x) ctx->pos_epoch++;
            } break;
            case AMM_OP_EXPR: {
                double *dst = (op.b >= 0 && op.b < AMM_MAX_SLOTS) ? ctx->slots[op.b] : nullptr;
                if (op.a < 0 || op.a >= (int)ctx->exprs.size() || !dst) {
                    amm_set_error("amm_run_ops: EXPR with an unknown expression or an unbound destination");
                    return 1;
                }
                const ExprDef &e = ctx->exprs[op.a];
                // the high bit of the counter keeps these streams apart from those of direct amm_expr_eval calls
                const unsigned long long counter = (1ull << 63) | ++ctx->expr_counter;
                if (amm_expr_eval_impl(ctx, e.code.data(), (int)e.code.size(), e.consts.data(), (int)e.consts.size(),
                                       e.globals.data(), (int)e.globals.size(), ctx->expr_seed, counter, dst, nullptr)) return 1;
            } break;
            case AMM_OP_SAVE_REF:
            case AMM_OP_CONSTRAIN_X:
            case AMM_OP_CONSTRAIN_V: {
                if (!ctx->constraints) break;       // no constraints in the System: identity (OpenMM does the same)
                if (op.op == AMM_OP_SAVE_REF) {
                    if (amm_constraints_save_reference(ctx, ctx->constraints, ctx->d_x)) return 1;
                } else if (op.op == AMM_OP_CONSTRAIN_X) {
                    if (amm_constrain_positions(ctx, ctx->constraints, ctx->d_x)) return 1;
                    ctx->pos_epoch++;
                } else if (amm_constrain_velocities(ctx, ctx->constraints, ctx->d_x, ctx->d_v)) return 1;
            } break;
            case AMM_OP_BATH: {
                if (op.a < 0 || op.a >= (int)ctx->baths.size()) {
                    amm_set_error("amm_run_ops: BATH with an unknown bath id");
                    return 1;
                }
                if (amm_bath_impl(ctx, ctx->baths[op.a], ctx->d_v, (1ull << 63) | ++ctx->expr_counter)) return 1;
            } break;
            case AMM_OP_ALLREDUCE: {
                auto bound = [&](const amm_op &o) { return o.a >= 0 && o.a < AMM_MAX_SLOTS && ctx->slots[o.a]; };
                if (!bound(op)) {
                    amm_set_error("amm_run_ops: ALLREDUCE of an unbound buffer");
                    return 1;
                }
                // consecutive all-reduces of buffers that are neighbours in memory (the near and the outer force after a
                // dual evaluation) travel as ONE message: the exchange is latency bound at 2.4 MB
                const size_t n3 = 3 * (size_t)ctx->n;
                double *lo = ctx->slots[op.a];
                size_t count = n3;
                while (k + 1 < n_ops && ops[k + 1].op == AMM_OP_ALLREDUCE && bound(ops[k + 1])) {
                    double *nb = ctx->slots[ops[k + 1].a];
                    if (nb == lo + count) count += n3;
                    else if (nb + n3 == lo) { lo = nb; count += n3; }
                    else break;
                    ++k;
                }
                if (amm_comm_allreduce_impl(ctx, lo, count)) return 1;
            } break;
            default: amm_set_error("amm_run_ops: unknown op"); return 1;
            }
        }
    }
    if (flush_deferred()) return 1;
    if (cursor && !yielded) *cursor = total_ops;
    if (!yielded && !ctx->own_only.empty()) {
        // the program is through: the caller may read any force buffer now (the engine serves cached forces): bond-list groups are
        // evaluated again in full, a pair group left with this rank's rows only is an error (a RESPA program ends on a whole evaluation)
        const std::vector<const double *> left = ctx->own_only;
        for (const double *b : left)
            if (complete(b)) return 1;
    }
    if (swapped) {   // odd number of fused iterations: bring the state back into the caller's buffers
        const size_t bytes = sizeof(double) * 3 * (size_t)ctx->n;
        AMM_HIP(hipMemcpyAsync(user_x, ctx->d_x, bytes, hipMemcpyDeviceToDevice, ctx->stream));
        AMM_HIP(hipMemcpyAsync(user_v, ctx->d_v, bytes, hipMemcpyDeviceToDevice, ctx->stream));
        AMM_HIP(hipMemcpyAsync(user_f0, ctx->slots[f0_slot], bytes, hipMemcpyDeviceToDevice, ctx->stream));
    }
    bindings.restore(false);        // the copies above were checked; only the pointers are left to rebind
    return 0;
}

int amm_pair_get_stats(amm_ctx *ctx, int32_t force_id, amm_pair_stats *out) {
    PairForce *pf = get_pair(ctx, force_id);
    if (!pf || !out) return 1;
    std::memset(out, 0, sizeof(*out));
    AMM_HIP(hipStreamSynchronize(ctx->stream));
    out->n_evals = pf->n_evals;
    PairForce *L = pf->host ? pf->host : pf;
    out->capacity = L->cap;
    out->lanes_per_atom = L->lpa;
    out->n_cells = L->grid.ncell;
    out->rlist = pf->rlist;
    out->n_slice_atoms = L->s_end - L->s_begin;
    out->shares_list = pf->host ? 1 : 0;
    out->list_kind = L->last_kind;
    out->tab_error = pf->tab_error;
    out->has_table = (pf->pc.tab.nint > 0 && pf->d_tab) ? 1 : 0;
    out->rode_along = pf->last_fused;
    out->chargeless = pf->last_chargeless;
    out->build_split = (L->cl && L->last_kind != 0) ? L->cl->split_parts : 0;
    out->has_site_table = (ctx->opt_site_tab && pf->d_tab_ss && pf->pc.tab.ss_first >= 0) ? 1 : 0;
    out->site_tab_error = pf->ss_error;
    out->n_rest_atoms = L->hybrid ? L->n_rest : 0;
    if (pf->small) {
        int cs[2];
        if (amm_small_group_stats(pf->small, cs)) return 1;
        out->n_candidates = cs[0];
        out->n_candidate_walks = cs[1];
    }
    if (L->last_kind >= 1 && L->cl && L->cl->built) {
        ClusterList *cl = L->cl;
        int flags[8];
        unsigned long long cnt[8];
        AMM_HIP(hipMemcpy(flags, cl->d_flags, sizeof(flags), hipMemcpyDeviceToHost));
        AMM_HIP(hipMemcpy(cnt, cl->d_counters, sizeof(cnt), hipMemcpyDeviceToHost));
        out->capacity = cl->cap;
        out->lanes_per_atom = cl->lpa;
        out->n_cells = cl->grid.ncell;
        out->n_slice_atoms = 3 * (int64_t)(cl->c_end - cl->c_begin);
        out->n_builds = (int64_t)cnt[0];
        out->n_list_pairs = 9 * (int64_t)(pf->host ? cnt[2] : cnt[1]);      // atom pairs evaluated: nine per molecule-pair entry
        out->max_neighbors = flags[2];
        out->rlist_outer = L->desc.rc + L->skin;
        if (L->last_kind == 2 && L->rest && L->rest->built) {          // + the entries of the per-atom part
            AMM_HIP(hipMemcpy(cnt, L->rest->d_counters, sizeof(cnt), hipMemcpyDeviceToHost));
            out->n_list_pairs += (int64_t)(pf->host ? cnt[2] : cnt[1]);
        }
        return 0;
    }
    if (L->built) {
        int flags[8];
        unsigned long long cnt[8];
        AMM_HIP(hipMemcpy(flags, L->d_flags, sizeof(flags), hipMemcpyDeviceToHost));
        AMM_HIP(hipMemcpy(cnt, L->d_counters, sizeof(cnt), hipMemcpyDeviceToHost));
        out->n_builds = (int64_t)cnt[0];
        out->n_outer_builds = (int64_t)cnt[4];
        out->n_outer_pairs = (int64_t)cnt[3];
        out->rlist_outer = L->desc.rc + L->skin_out;
        out->n_list_pairs = (int64_t)(pf->host ? cnt[2] : cnt[1]);   // a guest walks the front parts only
        out->max_neighbors = flags[2];
    }
    return 0;
}

int amm_pair_row_padding(amm_ctx *ctx, int32_t force_id, int64_t out[2]) {
    PairForce *pf = get_pair(ctx, force_id);
    if (!pf || !out) {
        amm_set_error("amm_pair_row_padding: null argument or not a pair force");
        return 1;
    }
    PairForce *L = pf->host ? pf->host : pf;
    out[0] = out[1] = 0;
    if (!(L->last_kind >= 1 && L->cl && L->cl->built)) return 0;       // per-atom rows: not reported
    long long v[2];
    if (amm_cluster_row_padding_impl(ctx, pf, v)) return 1;
    out[0] = v[0];
    out[1] = v[1];
    return 0;
}

int amm_pair_count_within(amm_ctx *ctx, int32_t force_id, const double *d_pos, double r_within, int64_t *count) {
    PairForce *pf = get_pair(ctx, force_id);
    if (!pf || !d_pos || !count) {
        amm_set_error("amm_pair_count_within: null argument or not a pair force");
        return 1;
    }
    long long c = 0;
    PairForce *Lw = pf->host ? pf->host : pf;
    if (Lw->last_kind >= 1 && Lw->cl && Lw->cl->built) {
        if (amm_cluster_count_within_impl(ctx, pf, d_pos, r_within, &c)) return 1;
        if (Lw->last_kind == 2 && pf->rest && Lw->rest && Lw->rest->built) {
            long long c2 = 0;
            if (amm_pair_count_within_impl(ctx, pf->rest, d_pos, r_within, &c2)) return 1;
            c += c2;
        }
    } else if (amm_pair_count_within_impl(ctx, pf, d_pos, r_within, &c)) return 1;
    *count = (int64_t)c;
    return 0;
}

const char *amm_kernel_revision(void) { return amm_kernel_revision_impl(); }

int amm_set_outer_skin(amm_ctx *ctx, double skin_out) {
    ctx->skin_out = skin_out;
    return 0;
}

int amm_set_option(amm_ctx *ctx, const char *name, double value) {
    if (!ctx || !name) {
        amm_set_error("amm_set_option: null argument");
        return 1;
    }
    const std::string k(name);
    const int v = (int)value;
    if (k == "cluster") ctx->opt_cluster = v;
    else if (k == "hybrid") ctx->opt_hybrid = v;
    else if (k == "small_group") ctx->opt_small_group = v;
    else if (k == "mixed_terms") ctx->opt_mixed_terms = v;
    else if (k == "rest_skin_factor") ctx->opt_rest_skin_factor = value;
    else if (k == "tab") ctx->opt_tab = v;
    else if (k == "site_trips") ctx->site_trips = v != 0;
    else if (k == "lanes_per_row") ctx->opt_lpa = v;
    else if (k == "build_parts") ctx->opt_parts = v;
    else if (k == "build_split") ctx->opt_build_split = v;
    else if (k == "unroll") ctx->opt_unroll = v;
    else if (k == "dual_unroll") ctx->opt_dual_unroll = v;
    else if (k == "tab_block") ctx->opt_tab_bs = v;
    else if (k == "tab_dual_block") ctx->opt_tab_dual_bs = v;
    else if (k == "no_dual") ctx->opt_no_dual = v;
    else if (k == "fuse_rows") ctx->opt_fuse_rows = v;
    else if (k == "row_phases") ctx->opt_row_phases = v;
    else if (k == "group_candidates") ctx->opt_group_candidates = v;
    else if (k == "positions_private") ctx->opt_positions_private = v;
    else if (k == "site_tab") ctx->opt_site_tab = v;
    else if (k == "no_defer") ctx->opt_no_defer = v;
    else if (k == "terms_from") ctx->opt_terms_from = v;
    else if (k == "no_term_lanes") ctx->opt_no_term_lanes = v;
    else if (k == "fuse_epilogue") ctx->opt_fuse_epilogue = v;
    else if (k == "comm_timeout") ctx->opt_comm_timeout = value;
    else if (k == "spec_assign") ctx->opt_spec_assign = v;
    else if (k == "chargeless") ctx->opt_chargeless = v;
    else if (k == "state_exchange") ctx->opt_state_exchange = v;
    else {
        amm_set_error("amm_set_option: unknown option '" + k + "'");
        return 1;
    }
    return 0;
}

int amm_run_stats(amm_ctx *ctx, int64_t out[4]) {
    if (!ctx || !out) return 1;
    out[0] = ctx->n_epilogues;
    out[1] = ctx->n_copies_current;
    out[2] = ctx->n_state_exchanges;
    out[3] = 0;
    return 0;
}

int amm_exchange_per(amm_ctx *ctx, int32_t *per) {
    if (!ctx || !per) return 1;
    *per = amm_slice_per(ctx->n, ctx->world);
    return 0;
}

int amm_set_fuse_inner(amm_ctx *ctx, int32_t on) {
    ctx->fuse_inner = on != 0;
    return 0;
}

int amm_profile_enable(amm_ctx *ctx, int32_t on) {
    ctx->profile = on != 0;
    ctx->profile_only = on < 0 ? -on - 1 : -1;
    return 0;
}

int amm_profile_read(amm_ctx *ctx, int32_t force_id, int64_t *n_launches, double *total_ms) {
    PairForce *pf = get_pair(ctx, force_id);
    if (!pf) return 1;
    AMM_HIP(hipStreamSynchronize(ctx->stream));
    double tot = 0.0;
    for (size_t k = 0; k + 1 < pf->ev_used; k += 2) {
        float ms = 0.f;
        AMM_HIP(hipEventElapsedTime(&ms, pf->ev[k], pf->ev[k + 1]));
        tot += ms;
    }
    if (n_launches) *n_launches = (int64_t)(pf->ev_used / 2);
    pf->ev_used = 0;
    // hybrid lists: an evaluation is two launches -- the molecule rows (timed above) and the per-atom part kept by the hidden child,
    // whose time belongs to the same evaluations (the launch count stays the parent's)
    if (pf->rest) {
        PairForce *rc = pf->rest;
        for (size_t k = 0; k + 1 < rc->ev_used; k += 2) {
            float ms = 0.f;
            AMM_HIP(hipEventElapsedTime(&ms, rc->ev[k], rc->ev[k + 1]));
            tot += ms;
        }
        rc->ev_used = 0;
    }
    if (total_ms) *total_ms = tot;
    return 0;
}

}  // extern "C"
