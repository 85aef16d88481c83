// atomsmm_amd/csrc/cluster.h -- molecule-row ("cluster") neighbour lists of the force-only pair traversal (gfx950).
//
// The three-site molecules of a pair force -- three consecutive atoms whose three intramolecular pairs are their only exclusions (a
// water: the reference's exceptions -> exclusions, forces.py:310-312) -- keep ONE neighbour row per MOLECULE on the force-only hot
// path instead of one per atom: an entry names a partner molecule, and the traversal evaluates the nine atom pairs of the two
// molecules from registers (this is the "i-tile" of BASELINE.json's north_star, three atoms tall).  Against per-atom rows:
//   * a ninth of the row entries (list build writes, index stream, row bookkeeping per pair);
//   * a third of the j-record gathers per pair, one periodic image per molecule pair instead of nine;
//   * the Lennard-Jones arithmetic exactly where two sites meet (O-O: one pair in nine), with no ordering of the rows by
//     site class -- so a row's order of summation depends on the row alone again (bit-identical between decompositions);
//   * exclusions need no look-up at all: a molecule is not its own neighbour.
// Price: a molecule pair is listed as soon as ANY of its nine atom pairs is within the list radius, so somewhat fewer of the
// evaluated pairs lie inside the cutoff (measured in DESIGN.md).  A box of nothing but such molecules walks these rows only; when
// the molecules share the box with other atoms (ions, a solute, a chain) the list is HYBRID: these rows for the pairs of two
// molecules, per-atom rows (pair.hip, kept by the force's hidden child) for every pair with another atom.  Energy evaluations,
// guarded / grouped / softcore / virial forces and particle lists with fewer than half of their atoms in molecules keep the per-atom
// rows of pair.hip (built only when such an evaluation asks).
#pragma once
#include "amm_ctx.h"

struct ClusterList {
    int nc = 0;                    // molecules (clusters of 3 atoms: first[m], first[m] + 1, first[m] + 2)
    const int *d_first = nullptr;  // first atom of every molecule (the force's); nullptr: 3 m (every atom of the force is in a molecule)
    const int *d_rest = nullptr;   // hybrid list: the atoms outside the molecules (their pairs are the child's, per-atom rows) ...
    int nrest = 0;                 // ... and their number
    bool built = false;
    CellGrid grid;
    int capc = 0;                  // members per cell in the cell tables
    int parts = 1;                 // wavefronts per cell in the build
    int *d_slice_cells = nullptr;  // [2]: the cells that hold the first / the last row of this rank's slice (k_csort_gather)
    int split_parts = 0;           // > 0: blocks per cell of the split-stream build (slices; k_cbuild<.., SPLIT>)
    double rext = 0;               // bound on the distance of a molecule's atoms from its first atom (cells, interior margin)
    double rlist_build = 0, rnear_build = 0, skin = 0;
    int *d_cell_count = nullptr, *d_cell_start = nullptr, *d_cell_members = nullptr;
    int *d_cperm = nullptr;        // sorted cluster -> molecule
    int *d_aperm = nullptr;        // sorted atom slot (3 c + a) -> atom (3 m + a): what k_unsort and the stores index with
    float4 *d_pos4f = nullptr;     // [3 nc] fp32 positions at the last build, molecules kept whole; .w: atom 0 extent, atom 1 site bits
    double *d_xref = nullptr;      // [n][3] positions at the last build (displacement trigger)
    int c_begin = 0, c_end = 0;    // this rank's slice of the sorted clusters
    int cap = 0;                   // entries per row
    int lpa = 8;                   // lanes per row in the traversal
    int *d_nl = nullptr, *d_nnb = nullptr, *d_nnb_near = nullptr;
    int *d_flags = nullptr;        // [0] rebuild wanted [1] row overflow [2] longest row [6] fullest cell [7] cell table / extent overflow
    unsigned long long *d_counters = nullptr;    // [0] builds [1] entries [2] front entries [7] scratch (count_within)
    unsigned long long *d_blockstats = nullptr;
    int *d_ticket = nullptr;
    long checked_epoch = -1, pre_epoch = -1;
    const double *checked_pos = nullptr, *pre_pos = nullptr;
    // sorted fp64 copies written ahead of time by the launch that moved the atoms (cluster.hip: cepi_rows): for which force, and for
    // which positions (epoch / buffer) -- the next evaluation of that force then skips the gather
    PairForce *sorted_for = nullptr;
    long sorted_epoch = -1;
    const double *sorted_pos = nullptr;
    // cells assigned ahead of time by the launch that moved the atoms (cluster.hip: cepi_rows files every molecule it has moved in
    // its new cell; should the triggers ask for a rebuild, the kernel's last block turns the counts into cell_start and the next
    // evaluation's chain starts at the sort): two count arrays in turn (a launch clears the one the NEXT launch counts in)
    int *d_spec_count[2] = {nullptr, nullptr};
    int *d_spec_ticket = nullptr;
    int spec_parity = 0;
    long assigned_epoch = -1;
    const double *assigned_pos = nullptr;
    bool per_pair_image = false;   // small box: the periodic image is chosen per atom pair, not per molecule pair
};

// three-site molecules of a force and the atoms outside them (host; exclusion CSR in original indices)
void amm_cluster_classify(int n, const std::vector<int> &excl_ptr, const std::vector<int> &excl_idx, std::vector<int> &mol_first,
                          std::vector<int> &rest);
// the force-only evaluation of `pf` (and of `guest` on the same pass) over molecule rows; same contract as amm_pair_eval_impl
int amm_cluster_eval_impl(amm_ctx *ctx, PairForce *pf, const double *d_pos, double *d_force, int accumulate, PairForce *guest,
                          double *g_force, int g_accumulate, int exchange);
int amm_cluster_count_within_impl(amm_ctx *ctx, PairForce *pf, const double *d_pos, double r_within, long long *count);
int amm_cluster_row_padding_impl(amm_ctx *ctx, PairForce *pf, long long out[2]);
int amm_cluster_state_finish_impl(amm_ctx *ctx);       // second half of a state exchange: k_state_scatter
int amm_cluster_free(ClusterList *cl);
