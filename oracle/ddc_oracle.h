/*
 * ddc_oracle.h -- CPU restatement of ddcMD's Martini MD inner loop.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is product code: only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * or call it, and there only as the checker / the timed CPU baseline.
 *
 * PARITY UNPINNED: the reference ships no golden outputs and no runnable test
 * for this path, and cannot be compiled here (its util/, recbis/, cub/
 * submodules are empty; every hot-path .c includes three_algebra.h/object.h/
 * units.h from them).  The exception: primes.c builds by itself (oracle/_ref,
 * oracle/Makefile) and pins orc_next_prime / orc_lcg64_default's primes
 * (tests/test_ref_pinned.py).  This file restates the reference algorithm function by
 * function (each block cites /root/reference/src file:line) and is
 * cross-validated by an independent O(N^2) evaluation, finite differences
 * (forcetest.c method), sum(F)=0 and NVE drift in tests/.
 *
 * All quantities are in ddcMD internal units (bohr, fs, Rydberg, e; kB=1;
 * ddcMD.c:71-73).  Plain C99, FP64, single thread -- like the reference's
 * per-rank Martini path (no OpenMP in bioMartini.c).
 */
#ifndef DDC_ORACLE_H
#define DDC_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Flat parameter block.  Arrays are owned by the caller and must outlive the
 * calls.  Layout mirrors what martini_parms()/martiniLJ_parms() hold
 * (bioMartini.c:868-950,1210-1353) without the pointer graphs. */
typedef struct orc_params
{
   /* BOX (orthorhombic, centred on the origin: box.c:56-67) */
   double hxx, hyy, hzz;
   int pbc;                    /* bitmask x=1 y=2 z=4 */
   /* NEIGHBOR */
   double deltaR;              /* skin (neighbor.c:49-54) */
   /* MARTINI potential */
   double rmax;                /* "cutoff" key: LJ and Coulomb cutoff */
   double krf, crf;            /* bioMartini.c:1234-1245 */
   double keR;                 /* ke/epsilon_r (bioMartini.c:1033) */
   int nlj;                    /* number of LJ atom types (mmff->nAtomType) */
   const double *sigma;        /* [nlj*nlj] */
   const double *eps;          /* [nlj*nlj] */
   const double *shift;        /* [nlj*nlj] CGLennardJones_setShift */
   /* SPECIES tables */
   int nspecies;
   const double *mass;         /* [nspecies] */
   const double *charge;       /* [nspecies]  (ddcenergy.c:210 q[i]=species charge) */
   const int *ljtype;          /* [nspecies]  getCGLJindexbySpecie */
   const int *moltype;         /* [nspecies]  speciesIndexToMoleculeIndex */
   const int *resitype;        /* [nspecies]  findResiConnNew(species name) */
   /* MOLECULE types: exclusion (bpair) lists, reOrgPairs bioMartini.c:1392 */
   int nmoltype;
   const int *mol_nspecies;    /* [nmoltype] moleculeType->nSpecies */
   const int *bpair_off;       /* [nmoltype+1] */
   const int *bpairI;          /* atmgrp codes (group<<8|atom) */
   const int *bpairJ;
   /* RESIDUE types: bonded terms, offsets are indices inside the residue */
   int nresi;
   const int *resi_natoms;     /* [nresi] atomListSize */
   const int *bond_off;        /* [nresi+1] */
   const int *bondI, *bondJ;
   const double *bond_kb, *bond_b0;
   const int *angle_off;       /* [nresi+1] */
   const int *angleI, *angleJ, *angleK, *angle_func; /* func 1,2,10 */
   const double *angle_k, *angle_t0;
   const int *tors_off;        /* [nresi+1] */
   const int *torsI, *torsJ, *torsK, *torsL, *tors_func, *tors_n; /* func 1 proper, 2 improper */
   const double *tors_k, *tors_delta;
   int excludePotentialTerm;   /* bitmask bioCharmmParms.h:25-28 */
   /* RESTRAINT potential (restraint.c:259-361): harmonic position restraints by gid */
   int nrest;
   const uint64_t *rest_gid;   /* [nrest] */
   const int *rest_fc;         /* [3*nrest] fcx fcy fcz */
   const double *rest_r0;      /* [3*nrest] x0 y0 z0 as fractions of the box */
   const double *rest_kb;      /* [nrest] */
   int rest_origin;            /* 0: box centred on the origin (x0*L - L/2) */
   /* distance constraints of nglfconstraint (CONSLISTPARMS/CONSPARMS bioMMFF.c:64-104; groups built by
    * genConstraint bioMartini.c:300-445): pairs of residue type r are [cons_off[r], cons_off[r+1]),
    * atom offsets inside the residue, cons_grp = constraint list of the pair.  NULL cons_off: none. */
   const int *cons_off, *consI, *consJ, *cons_grp;
   const double *cons_r0;
   int baro_isotropic;         /* 1: changeVolumeGPUisotropic (molecularPressureGPU.cu:204-239) instead of changeVolume's semi-isotropic form */
} orc_params;

/* energies returned by orc_forces: */
enum { ORC_E_LJ = 0, ORC_E_ELE, ORC_E_BOND, ORC_E_ANGLE, ORC_E_TORS, ORC_E_IMPR, ORC_E_TOTAL, ORC_E_RESTRAINT, ORC_NE };

typedef struct orc_nbr orc_nbr;   /* half neighbour list (opaque) */

/* pairlist1 (pairlist.c:205-314) + reOrgPairs (bioMartini.c:1392-1485).
 * Builds the half list {(i,j): gid_i<gid_j, |r_ij|_minimage^2 < (rmax+deltaR)^2}
 * split into kept (list 0) and pruned/excluded (list 1) pairs. */
orc_nbr *orc_nbr_build(const orc_params *p, int n, const double *rx, const double *ry, const double *rz,
                       const uint64_t *gid, const int *species);
void orc_nbr_free(orc_nbr *nb);
void orc_branch_census(long out[8], int reset);                   /* dihedral series / wrap branches taken (process-wide; tests) */
long orc_nbr_build_count(void);                                   /* list builds so far (process-wide; tests) */
/* neighborCheck (neighbor.c:117-208), constant box: 1 = the list must be rebuilt (updateRate == 0 decks) */
int orc_neighbor_check(const orc_params *p, const orc_nbr *nb, int n, const double *rx, const double *ry, const double *rz);
long orc_nbr_npairs(const orc_nbr *nb, int which);             /* which=0 kept, 1 excluded */
void orc_nbr_csr(const orc_nbr *nb, int which, const int **start, const int **j);

/* ddcenergy (ddcenergy.c:160-238) for one rank: zero f; martiniNonBond +
 * martiniIntraMoleReaction + charmmConvalent.  e[ORC_NE], virial[6]=xx,yy,zz,xy,xz,yz */
void orc_forces(const orc_params *p, const orc_nbr *nb, int n,
                const double *rx, const double *ry, const double *rz,
                const uint64_t *gid, const int *species,
                double *fx, double *fy, double *fz, double *e, double *virial);

/* individual pieces (accumulate into f, e, virial like the reference) */
void orc_nonbond(const orc_params *p, const orc_nbr *nb, int n,
                 const double *rx, const double *ry, const double *rz, const int *species,
                 double *fx, double *fy, double *fz, double *vLJ, double *vEle, double *virial);
void orc_intramol(const orc_params *p, const orc_nbr *nb, int n,
                  const double *rx, const double *ry, const double *rz, const int *species,
                  double *fx, double *fy, double *fz, double *vEle, double *virial);
void orc_bonded(const orc_params *p, int n,
                const double *rx, const double *ry, const double *rz,
                const uint64_t *gid, const int *species,
                double *fx, double *fy, double *fz, double *e4 /* bond,angle,tors,impr */, double *virial);

/* Independent check: O(N^2) over all index pairs with rint-based minimum
 * image; no cell grid, no list, no gid ordering. */
void orc_brute_force(const orc_params *p, int n,
                     const double *rx, const double *ry, const double *rz,
                     const uint64_t *gid, const int *species,
                     double *fx, double *fy, double *fz, double *vLJ, double *vEle, double *virial,
                     long *npair_in_cut);

/* kinetic_terms (energy.c:104-147): per-group (by_species 0) / per-species copies and the thermal flux, 12 doubles per class */
void orc_kinetic_detail(const orc_params *p, int n, const double *vx, const double *vy, const double *vz,
                        const int *species, const int *group, int by_species, int nclass, double *out);
/* kinetic_terms (energy.c:48-163): rk and tion[6] */
void orc_kinetic(const orc_params *p, int n, const double *vx, const double *vy, const double *vz,
                 const int *species, double *rk, double *tion);

/* eval_energyInfo arithmetic (energyInfo.c:75-116): out[0]=temperature,
 * out[1]=pion (pressure), out[2..7]=sion, out[8]=energy total */
void orc_energyinfo(const orc_params *p, double natoms, int nConstraints, double eion, double rk,
                    const double *virial, const double *tion, double *out);

/* LCG64_PARM (lcg64.h:8-12): the random state a particle carries */
typedef struct orc_lcg64_parm { unsigned long long state; unsigned int multID, prime; } orc_lcg64_parm;
typedef struct orc_primes { unsigned long long blockSize, taskId, nTasks, upperBound, prime, iBlock; } orc_primes;
double orc_lcg64(orc_lcg64_parm *q);                       /* lcg64.c:127-135 */
void orc_gasdev3d(orc_lcg64_parm *q, double g[3]);         /* random.c:135-160 */
int orc_is_prime1(unsigned long long N);                   /* primes.c:120-155 */
void orc_prime_init(orc_primes *g, unsigned blockSize, unsigned taskId, unsigned nTasks);      /* primes.c:25-31 */
unsigned long long orc_next_prime(orc_primes *g);          /* primes.c:35-63 */
void orc_lcg64_default(int n, const uint64_t *label, unsigned taskId, unsigned nTasks, orc_lcg64_parm *out);   /* collection.c:95-109, lcg64.c:98-110 */

/* thermostat description per group (group.c:48-90) */
typedef struct orc_group
{
   int type;         /* 0 FREE (free.c), 1 BERENDSEN (berendsen.c), 2 LANGEVIN (langevin.c: Teq as the caller sets it step by step) */
   double Teq, tau;  /* berendsen, langevin (Teq in energy units, kB = 1) */
   int interval;
   /* dynamic state (berendsen.c:12-20) */
   double lambda, Tsum; int nT, doScaling;
   double temperature; /* g->energyInfo.temperature, refreshed by orc_group_temperature */
   unsigned long long seed;   /* langevin: seed of the counter-based normal stream (see orc_gauss3) */
   orc_lcg64_parm *lcg;       /* langevin: not NULL = the reference's per-particle LCG64 streams, [n] in particle order (advanced in place) */
   double vcm[3];             /* langevin: the velocity the friction relaxes towards (langevin.c:106,167; p->vcm) */
} orc_group;

/* nglf (nglf.c:67-112): one velocity-Verlet step.  Rebuilds the list when
 * (loop % updateRate)==0 after the drift (ddcUpdateAll.c:64-71).  loop/time are
 * advanced in place.  *pnb may be replaced. */
void orc_nglf_step(const orc_params *p, orc_nbr **pnb, int updateRate, double dt,
                   long *loop, double *time, int n,
                   double *rx, double *ry, double *rz, double *vx, double *vy, double *vz,
                   double *fx, double *fy, double *fz,
                   const uint64_t *gid, const int *species, const int *group,
                   int ngroup, orc_group *groups,
                   double *e, double *virial, double *rk, double *tion);

/* eval_energyInfo group branch (energyInfo.c:118-141): per-group temperature */
void orc_group_temperature(const orc_params *p, int n, const double *vx, const double *vy, const double *vz,
                           const int *species, const int *group, int ngroup, orc_group *groups);

void orc_back_in_box(const orc_params *p, int n, double *rx, double *ry, double *rz);

/* nglfconstraint's barostat for a system of single-bead molecules (nglfconstraint.c:527-536,
 * changeVolume :64-84, molecularPressure molecularPressure.c:57-67, adjustPosn :44-56): from the virial
 * of the last force evaluation, scales the box (p->hxx.. in place) and the positions.  Call before
 * orc_nglf_step: together they are one nglfconstraint step without constraints. */
void orc_barostat(orc_params *p, int n, double *rx, double *ry, double *rz, const double virial[6],
                  double T, double P0, double beta, double tau, double dt);

/* velocityConstraintOld + resMoveConsOld (nglfconstraint.c:180-264, 438-455): Gauss-Seidel sweeps over the
 * pairs of every constraint group until max |rvab dt| < 1e-12 (at most 500 sweeps).  location 0 = FRONT
 * (frontFunc :122-131: the drifted pair r + dt v has the constrained length), 1 = BACK (backFunc :133-137:
 * r.v = 0).  Velocities change in place.  Returns the largest sweep count used. */
int orc_velocity_constraint(const orc_params *p, int n, double dt, int location,
                            const double *rx, const double *ry, const double *rz, double *vx, double *vy, double *vz,
                            const uint64_t *gid, const int *species);

/* The barostat for molecules of several beads: molecularVirial (molecularPressure.c:23-56) takes
 * sum_i (r_i - R_molecule) f_i off the diagonal of the virial (R = centre of mass, images resolved
 * relative to one atom of the molecule), molecularPressure adds N_molecules kB T; then changeVolume.
 * f = the forces of the last evaluation.  Molecules are runs of equal gid & molMask. */
void orc_barostat_mol(orc_params *p, int n, double *rx, double *ry, double *rz,
                      const double *fx, const double *fy, const double *fz, const uint64_t *gid, const int *species,
                      const double virial[6], double T, double P0, double beta, double tau, double dt, double pmol[3]);

#ifdef __cplusplus
}
#endif
#endif
