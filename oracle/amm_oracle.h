/*
 * oracle/amm_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU (plain C, fp64) restatement of the arithmetic that AtomsMM's RESPA-split nonbonded path
 * denotes.  AtomsMM itself contains no arithmetic: it hands energy-expression strings to OpenMM
 * (third-party, un-vendored, absent from the build container; PDB fixtures were written by
 * OpenMM 7.2.2 -- SURVEY.md section 8c).  This file restates those expressions, each function
 * citing the reference file:line it follows, plus the OpenMM pair-loop semantics that SURVEY.md
 * Appendix B confirmed numerically against the reference's own known-answer literals.
 *
 * Pinning: tests/test_oracle_golden.py checks this oracle against the reference's golden
 * energies G1-G7 (tests/golden/goldens.json, literals cited from the reference test files).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 * The product path (atomsmm_amd/) never imports, links or calls it.
 */
#ifndef AMM_ORACLE_H
#define AMM_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

/* pair-energy families (same numbering as include/atomsmm_hip.h) */
enum {
    AMMO_NEAR_NONE = 0,      /* S(u) * V_LJC                          forces.py:541-543 */
    AMMO_NEAR_SHIFT = 1,     /* S(u) * (V_LJC(r) - V_LJC(rc0))        forces.py:544-548 */
    AMMO_NEAR_FSWITCH = 2,   /* force-switched LJC                    forces.py:549-563 */
    AMMO_DAMPED = 3,         /* SW * (LJ + erfc(alpha r) Kc qq / r)   forces.py:448-455 */
    AMMO_NONBONDED = 4,      /* OpenMM NonbondedForce direct space    forces.py:134-190, 723 */
    AMMO_SOFTCORE = 5,       /* 4 lambda eps (1-x)/x^2, x = (r/sigma)^6 + (1-lambda)/2, between the two sets of an
                                interaction group (systems.py:266-272); lambda in `alpha`; the `charge` array carries
                                the set of each atom (1, 2, 0 = none): a pair counts iff the codes multiply to 2 */
    AMMO_LJ_VIRIAL = 6       /* -r dV_LJ/dr = 24 eps (2 (sigma/r)^12 - (sigma/r)^6) as an "energy" (ComputingSystem,
                                systems.py:894), with the imported built-in switch */
};

/* flags */
enum {
    AMMO_GUARD_RC0 = 1,      /* multiply by step(rc0 - r)             forces.py:661, 714 */
    AMMO_COULOMB_EWALD = 2,  /* NONBONDED: Kc qq erfc(alpha r)/r (Ewald/PME direct space) */
    AMMO_COULOMB_RF = 4,     /* NONBONDED: reaction field (CutoffPeriodic); parity unpinned */
    AMMO_SWITCH = 8,         /* NONBONDED: built-in switch rswitch->rc on the LJ term */
    AMMO_NO_SHIFT = 16,      /* NEAR_FSWITCH without the constant -V*(rc0)   systems.py:823-846 */
    AMMO_GROUP_LJ = 32,      /* LJ-only interaction group: charges = set codes, pair iff product == 2, no Coulomb */
    AMMO_GROUP_Q = 64        /* Coulomb-only interaction group: sigma_i = twice the set code, pair iff the mixed sigma is 3, no LJ (systems.py:848-856) */
};

typedef struct {
    int family;
    int flags;
    int degree;      /* DAMPED: exponent d of u = (r^d - rs^d)/(rc^d - rs^d); 1 == OpenMM built-in switch */
    int pad_;
    double sign;     /* +1, or -1 for `subtract=True` / discount / group-31 copies */
    double rc;       /* actual cutoff: pairs with r >= rc are skipped */
    double rswitch;  /* DAMPED / NONBONDED switch start */
    double rc0, rs0; /* near-force inner cutoff / switch start */
    double alpha;    /* damping / Ewald splitting parameter, 1/nm */
    double Kc;       /* 138.935456 (forces.py:407) */
    double krf, crf; /* reaction-field constants (AMMO_COULOMB_RF) */
} ammo_pair_desc;

/* Energy (kJ/mol) and/or forces (kJ/mol/nm, ACCUMULATED into f[n*3]) of one pair force.
 * pos[n*3] AoS, orthorhombic box[3], minimum image; excl_ptr[n+1]/excl_idx = symmetric CSR of
 * excluded partners.  use_cells != 0 -> OpenMP cell-list traversal (large n), else O(n^2) i<j loop.
 * Returns number of in-cutoff pairs (i<j) or <0 on error. */
long ammo_pair_eval(const ammo_pair_desc *d, int n, const double *pos, const double *box,
                    const double *q, const double *sigma, const double *eps,
                    const int *excl_ptr, const int *excl_idx,
                    double *energy, double *f, int use_cells);

/* One pair: energy and -dE/dr / r ("force over r") for given mixed parameters. */
void ammo_pair_kernel(const ammo_pair_desc *d, double r2, double qq, double sig, double eps,
                      double *e, double *f_over_r);

/* Ewald exclusion correction: - Kc q_i q_j erf(alpha r)/r over all listed pairs (min image, no cutoff). */
void ammo_ewald_exclusion(int npairs, const int *pairs, const double *pos, const double *box,
                          const double *q, double alpha, double Kc, double *energy, double *f);

/* Ewald reciprocal-space sum (exact, |k| index <= kmax per axis) + self term; energy and forces. */
void ammo_ewald_reciprocal(int n, const double *pos, const double *box, const double *q,
                           double alpha, double Kc, int kmax, double *energy, double *f);

/* Long-range dispersion correction with switching (energy only). */
double ammo_dispersion_correction(int n, const double *sigma, const double *eps, const double *box,
                                  double rc, double rswitch, int use_switch);

/* Bond-list terms.  periodic != 0 -> minimum image. */
void ammo_ljc_bonds(int nb, const int *ij, const double *qq, const double *sig, const double *eps,
                    double Kc, const double *pos, const double *box, int periodic,
                    double *energy, double *f);                       /* forces.py:406 */
void ammo_near_bonds(const ammo_pair_desc *d, int nb, const int *ij, const double *qq, const double *sig,
                     const double *eps, const double *pos, const double *box, int periodic,
                     double *energy, double *f);                      /* forces.py:673-680 */
void ammo_harmonic_bonds(int nb, const int *ij, const double *r0, const double *k,
                         const double *pos, const double *box, int periodic, double *energy, double *f);
void ammo_harmonic_angles(int na, const int *ijk, const double *t0, const double *k,
                          const double *pos, const double *box, int periodic, double *energy, double *f);
void ammo_periodic_torsions(int nt, const int *ijkl, const int *per, const double *phase, const double *k,
                            const double *pos, const double *box, int periodic, double *energy, double *f);

/* Step-program primitives (propagators.py:249, 271). */
void ammo_kick(int n, double *v, const double *f, const double *fsub, const double *m, double coef);
void ammo_move(int n, double *x, const double *v, double coef);
double ammo_mvv(int n, const double *v, const double *m);

int ammo_num_threads(void);

#ifdef __cplusplus
}
#endif
#endif
