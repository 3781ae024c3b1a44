/*
 * deck.h -- load a ddcMD input deck (object.data + restart + parmfile + atoms
 * file) into one flat, plain-C description of the Martini hot path.
 *
 * Replaces, for this path only, what simulate_init / system_init /
 * martini_parms / mmff_init / collection_read build as pointer graphs
 * (simulate.c:104-297, system.c:79-214, bioMartini.c:1210-1353, bioMMFF.c,
 * collection_read.c:86-200).  Every key and default follows those call sites.
 */
#ifndef DDCMI_DECK_H
#define DDCMI_DECK_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

enum { DDCMI_GROUP_FREE = 0, DDCMI_GROUP_BERENDSEN = 1, DDCMI_GROUP_LANGEVIN = 2, DDCMI_GROUP_OTHER = 3 };

typedef struct ddcmi_setup
{
   /* SIMULATE (simulate.c:141-169,239-244) */
   int64_t loop, maxloop, deltaloop;
   double time, dt;
   int printrate, snapshotrate, checkpointrate;
   /* BOX (box.c:56-67) */
   double h[9];
   int pbc;
   /* NEIGHBOR (neighbor.c:49-54), DDC (ddc.c:49-107) */
   double deltaR;
   int updateRate;
   int lx, ly, lz;
   /* POTENTIAL type=MARTINI (bioMartini.c:1210-1245, :876-879) */
   double rmax, rcoulomb, epsilon_r, epsilon_rf, krf, crf, keR;
   int excludePotentialTerm;
   int potentialShift;
   /* LJ atom types (mmff->nAtomType) */
   int nlj;
   double *sigma, *eps, *shift;     /* [nlj*nlj]; unset pairs are NaN */
   /* SPECIES in ddcMD index order (molecule order, species.c:30) */
   int nspecies;
   char **species_name;
   double *mass, *charge;
   int *ljtype;       /* getCGLJindexbySpecie (bioMartini.c:952-987) */
   int *moltype;      /* speciesIndexToMoleculeIndex (molecule.c:56) */
   int *resitype;     /* index of the RESIPARMS the species belongs to */
   int *atomoffset;   /* position of the atom inside its residue's atomList */
   /* MOLECULE types + reOrgPairs exclusion lists (bioMartini.c:135-282,1416-1423) */
   int nmoltype;
   int *mol_nspecies, *bpair_off, *bpairI, *bpairJ;
   /* RESIDUE types: bonded terms (genMartiniConn bioMartini.c:571-838) */
   int nresi;
   int *resi_natoms;
   int *bond_off, *bondI, *bondJ;
   double *bond_kb, *bond_b0;
   int *angle_off, *angleI, *angleJ, *angleK, *angle_func;
   double *angle_k, *angle_t0;
   int *tors_off, *torsI, *torsJ, *torsK, *torsL, *tors_func, *tors_n;
   double *tors_k, *tors_delta;
   /* GROUPs (group.c:48-90, berendsen.c:91-113) */
   int ngroup;
   char **group_name;
   int *group_type;
   double *group_Teq, *group_tau;
   int *group_interval;
   /* atoms (collection_read.c:86-200), internal units */
   int natoms;
   double *rx, *ry, *rz, *vx, *vy, *vz;
   uint64_t *gid;
   int *species, *group;
   int nConstraints;
   /* INTEGRATOR / ACCELERATOR */
   char *integrator_type;
   int has_accelerator;
   char *accelerator_type;
   /* PRINTINFO unit strings (printinfo.c:51-64) */
   char *u_pressure, *u_volume, *u_temperature, *u_energy, *u_time, *u_length;
   /* RANDOM seed (random.c:44-60): seeds the Langevin noise */
   uint64_t rng_seed;
   /* POTENTIAL type=RESTRAINT (restraint.c:28-49): restraints by gid, r0 as box fractions */
   int nrest, rest_origin;
   int printMolecularPressure;        /* PRINTINFO printMolecularPressure (printinfo.c:56) */
   int nresicons;                     /* constraint pairs in the residues of the molecules in use (nglfconstraint needs 0 here) */
   /* INTEGRATOR type=NGLFCONSTRAINT (nglfconstraint.c:86-95): T, P0, beta, tauBarostat; beta = 0: no barostat */
   double npt_T, npt_P0, npt_beta, npt_tau;
   uint64_t *rest_gid;
   int *rest_fc;
   double *rest_r0, *rest_kb;
   /* RESIDUE types: distance constraints (CONSLISTPARMS/CONSPARMS, bioMMFF.c:64-104): pairs of residue r are
    * [cons_off[r], cons_off[r+1]); cons_grp = the constraint list (group) of the pair inside its residue */
   int *cons_off, *consI, *consJ, *cons_grp;
   double *cons_r0;
   /* INTEGRATOR type=NGLFGPULANGEVIN (nglfGPU.cu:422-507): isotropic barostat, every bead under group 0's Langevin thermostat */
   int npt_isotropic;
   /* PRINTINFO printStress / printHmatrix (printinfo.c:52-53): stress.data (stress tensor + thermal flux) and hmatrix.data */
   int printStress, printHmatrix;
   char *u_energyflux;                /* PRINTINFO ENERGYFLUX (printinfo.c:35-36), default ueV/Ang^2/fs */
   /* RANDOM type=LCG64 (random.c:44-71, lcg64.c): the particles' own streams, LCG64_PARM {state, multID, prime} per atom in file
    * order -- read from the random field of the atoms records (collection_read.c:160-166) or, when a record carries none,
    * lcg64_default's values for every atom (collection.c:95-109).  random_lcg64 = 0: no such object, the arrays are NULL */
   char *random_name;
   int random_lcg64, lcg_from_file;
   uint64_t *lcg_state;
   uint32_t *lcg_multID, *lcg_prime;
   double *group_vcm;      /* [3 ngroup] LANGEVIN groups: `vcm` (langevin.c:167), internal units; zero otherwise */
} ddcmi_setup;

/* lcg64_default over n particles in order (lcg64.c:98-110, primes.c:35-63 with prime_init(30000, task, ntasks), ddcMD.c:70) */
void ddcmi_lcg64_default(int n, const uint64_t *label, unsigned task, unsigned ntasks, uint64_t *state, uint32_t *multID, uint32_t *prime);

/* Load a deck.  object_file is required; restart_file may be NULL (then
 * "restart" next to object_file is tried, as run_ddcMD_CPU.sh does).  Relative
 * file names inside the deck (parmfile, atoms files) resolve against the
 * directory of object_file.  Returns NULL and fills err on failure. */
ddcmi_setup *ddcmi_deck_load(const char *object_file, const char *restart_file, char *err, int errlen);
/* Same, but additional object text (e.g. overriding the integrator/group
 * objects as SURVEY 8c prescribes) is compiled after the files. */
ddcmi_setup *ddcmi_deck_load_with(const char *object_file, const char *restart_file, const char *extra_objects, char *err, int errlen);
void ddcmi_setup_free(ddcmi_setup *s);
int ddcmi_setup_sizeof(void);

/* CRC-32 of the record checksums (crc32.c:46-84 of the reference: standard IEEE CRC-32) */
uint32_t ddcmi_crc32(const unsigned char *p, size_t n);

#ifdef __cplusplus
}
#endif
#endif
