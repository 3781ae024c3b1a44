/*
 * plugin.h -- host-side mirror of the ddcMD plugin surface for the Martini path.
 *
 * Same type names, member names and call signatures as the reference for the
 * members this path touches, so the glue reads like ddcMD's own:
 *   POTENTIAL   potential.h:42-58      eval_potential(sys, parms, e)
 *   INTEGRATOR  integrator.h:5-17      eval_integrator(ddc, simulate, parms)
 *   ACCELERATOR accelerator.h:11-31
 *   STATE       state.h:7-27           SoA particle store
 *   ETYPE       energyInfo.h           rk, eion, virial, tion, sion, pion, temperature
 *   SYSTEM / SIMULATE / DDC            the members nglf.c / ddcenergy.c / masters.c use
 * The device work behind them is libddcmi's C-ABI (include/ddcmi.h).  There is no
 * CPU force path in this library: decks without an ACCELERATOR object run on HIP
 * device 0 and say so.
 */
#ifndef DDCMI_PLUGIN_H
#define DDCMI_PLUGIN_H
#include <stdint.h>
#include <stdio.h>
#include "ddcmi.h"
#include "deck.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef uint64_t gid_type;
typedef struct { double x, y, z; } THREE_VECTOR;
typedef struct { double xx, yy, zz, xy, xz, yz; } THREE_SMATRIX;

typedef struct etype_st
{
   double rk, eion, pion, temperature, number, mass, eBath;
   THREE_SMATRIX virial, tion, sion;
   double energy;                       /* sys->energy = eion + rk (energyInfo.c:116) */
   THREE_VECTOR thermal_flux;           /* J = sum (K + U) v - S v / 2 (energy.c:112-114) */
} ETYPE;

typedef struct species_st { char *name; int index; double mass, charge; ETYPE energyInfo; /* the per-species copy kinetic_terms keeps (energy.c:136-147) */ } SPECIES;
enum GROUP_CLASS { FREE, BERENDSEN, LANGEVIN_GROUP, OTHER_GROUP };
typedef struct group_st { char *name; int index; int itype; double Teq, tau; int interval; ETYPE energyInfo;
                          double Tsum, lambda; int nT, doScaling;      /* BERENDSEN_PARMS' running state (berendsen.c:30-62), host integrator */
} GROUP;

typedef struct state_st
{
   double *rx, *ry, *rz;
   double *vx, *vy, *vz;
   double *fx, *fy, *fz;
   double *q;
   gid_type *label;
   SPECIES **species;
   GROUP **group;
   int nlocal, nion;
} STATE;

typedef struct box_st { double h0[9]; int pbc; double volume; } BOX_STRUCT;

/* The three plugin structs below have EVERY member of the reference's declarations, in the reference's order
 * (potential.h:42-58, integrator.h:5-17, accelerator.h:22-31), so that the glue of INTEGRATION.md compiles
 * against ddcMD's own headers unchanged and objects can be handed across; tests/abi/layout_check.c compares
 * offsets and sizes with a transcription of the reference's declarations.  The enums keep the reference's
 * enumerators and values (new ones are appended). */
enum ACCELERATOR_CLASS { GPU_CUDA, GPU_HIP };                /* accelerator.h:11; GPU_HIP: type=HIP */
typedef struct accelerator_st
{
   char *name;
   char *objclass;
   char *value;
   char *type;                          /* model */
   void *parent;
   enum ACCELERATOR_CLASS itype;
   void *parms;                         /* ddcmi_ctx* (GPUCUDAPARMS* in the reference's CUDA build) */
} ACCELERATOR;

/* neighbor.h:42-50 */
enum RCUT_ENUMS { RCUT_NONE = 0, RCUT_LOCAL = 1, RCUT_REMOTE = 2, RCUT_ALL = 3 };
enum NEIGHBORTABLETYPE { NEIGHBORTABLE_NONE = 0, NEIGHBORTABLE_SKINNY = 1, NEIGHBORTABLE_FAT = 2, NEIGHBORTABLE_GPU = 4 };
typedef struct rcut_str
{
   double value;
   enum RCUT_ENUMS mode;
   int type;
} RCUT_TYPE;

/* potential.h:7-31 */
enum POTENTIAL_CLASS { NO_POTENTIAL = -1, ZEROPOTENTIAL, MGPT, EAM, EAM1PASS, EAM2PASS, EAM_OPT, EAM_ONEPASS, PAIR, CHARMM, MARTINI, RESTRAINT, EWALD,
                       PLASMA, ORDERSH, ONEBODY, REFLECT, PAIRENERGY, MEAM, MIRRORSYM, LOCALYUKAWA, HYCOP, FMM, GPU_PAIR };
enum POT_COMM_MODE { POT_ONESIDED, POT_TWOSIDED };
typedef struct potential_st
{
   char *name;                          /* potential name */
   char *objclass;
   char *value;
   char *type;                          /* model */
   void *parent;
   enum POTENTIAL_CLASS itype;          /* integer label for type */
   void (*eval_potential)(void *sys, void *parms, void *e);
   void (*write_dynamics)(void *potential, FILE *file);
   RCUT_TYPE *(*getCutoffs)(void *sys, void *parms, int *nCutoffs);
   enum NEIGHBORTABLETYPE neighborTableType;
   int call_fsumX;
   int use_gpu_list;
   enum POT_COMM_MODE commMode;
   void *parms;                         /* MARTINIHIP_PARMS* */
} POTENTIAL;
/* what POTENTIAL.parms points at for type=MARTINI on this path (CHARMMPOT_PARMS in the reference) */
typedef struct martinihip_parms_st
{
   ddcmi_ctx *ctx;
   double rmax;                         /* "cutoff" key (bioMartini.c:876-879) */
   RCUT_TYPE rcut[2];                   /* charmmCutoff's answer (bioCharmm.c:386-414) */
   void *simulate;                      /* SIMULATE*: is the integrator on the device (martiniGPU1, bioMartini.cu:146-171)? */
} MARTINIHIP_PARMS;

/* integrator.h:4 */
enum INTEGRATOR_CLASS { NGLF, NGLFNEW, NGLFNK, NGLFRATTLE, NGLFCONSTRAINT, PNGLF, NGLFTEST, NVEGLF, NVEGLF_SIMPLE, NVTGLF, NPTGLF, STATIC, NEXTFILE, HYCOPINTEGRATOR };
typedef struct integrator_st
{
   char *name;
   char *objclass;
   char *value;
   char *type;
   void *parent;
   enum INTEGRATOR_CLASS itype;
   int uses_gpu;
   void (*eval_integrator)(void *ddc, void *simulate, void *parms);
   void (*writedynamic)(struct integrator_st *integrator, FILE *file);
   void *parms;
} INTEGRATOR;

typedef struct ddc_st { int updateRate; int lx, ly, lz; int update; double rcut; } DDC;

typedef struct system_st
{
   char *name;
   int npotential, ngroup, nspecies;
   POTENTIAL **potential;
   GROUP **group;
   SPECIES **species;
   STATE *state;                        /* sys->collection->state in the reference */
   BOX_STRUCT *box;
   ETYPE energyInfo;
   unsigned nlocal, nion;
   gid_type nglobal;
   int64_t loop;
   double time, energy;
   int nConstraints;
   double deltaR;                       /* sys->neighbor->deltaR */
} SYSTEM;

typedef struct simulate_st
{
   char *name;
   SYSTEM *system;
   INTEGRATOR *integrator;
   ACCELERATOR *accelerator;
   DDC *ddc;
   int64_t loop, maxloop;
   double time, dt;
   int printrate, snapshotrate, checkpointrate;
   char snapshotdir[512];
   ddcmi_setup *setup;                  /* the parsed deck */
   FILE *datafile;
   FILE *stressfile, *hmatfile;         /* stress.data / hmatrix.data next to the data file (printinfo.c:248-249) */
} SIMULATE;

/* accelerator.c:10-56 */
ACCELERATOR *accelerator_init(void *parent, const char *name, const char *type);
ACCELERATOR *accelerator_getAccelerator(ACCELERATOR *a);
/* potential.c:85-300 (type=MARTINI only) and bioMartini.c:1210-1353 */
POTENTIAL *potential_init(void *parent, const char *name, const char *type);
void martiniHIP(SYSTEM *sys, void *parms, ETYPE *e);
RCUT_TYPE *martiniCutoff(SYSTEM *sys, void *parms, int *n);      /* charmmCutoff, bioCharmm.c:386-414 */
/* integrator.c:37-167 (NGLF, NGLFGPU, NGLFHIP) and nglf.c:67-112 */
INTEGRATOR *integrator_init(void *parent, const char *name, const char *type);
void nglfHIP(DDC *ddc, SIMULATE *simulate, void *parms);
/* nglf (nglf.c:67-112) on the HOST over STATE, potential on the accelerator: INTEGRATOR.uses_gpu = 0, martiniHIP copies
 * positions up and forces back every step.  Selected for type NGLF / NVTGLF when DDCMI_CPU_INTEGRATOR=1 */
void nglf(DDC *ddc, SIMULATE *simulate, void *parms);
/* ddcenergy.c:160-238, energy.c:48-163, energyInfo.c:75-148 */
int ddcenergy(DDC *ddc, SYSTEM *sys, int e_eval_flag);
void kinetic_terms(SYSTEM *sys, int flag);
void eval_energyInfo(SYSTEM *sys);
/* more than one rank (RANK / WORLD_SIZE / LOCAL_RANK in the environment): the process grid simulate_init uses */
void plugin_plan_grid(int world, int lx, int ly, int lz, const double h[9], int grid[3]);
/* simulate.c:104-297, masters.c:369-559 (MD loop), printinfo.c:125-232 (data file) */
SIMULATE *simulate_init(const char *object_file, const char *restart_file, const char *extra_objects, char *err, int errlen);
int simulateMaster(SIMULATE *simulate, const char *datafile_path);
void simulate_free(SIMULATE *simulate);
void printinfo(SIMULATE *simulate, ETYPE *energyInfo, int header);
/* copy device state back into STATE (sendForceVelocityToHost + sendPosnToHost) */
int sendHostState(SYSTEM *sys);
/* writeRestart (io.c:58-113) + collection_writeBLOCK (collection_write.c:57-186): copies the state
 * back, writes <dir>/atoms#000000 (FIXRECORDASCII, CRC32 column) and <dir>/restart; dir = NULL
 * names it snapshot.<loop> (CreateSnapshotdir, io.c:115-142); restartLink: ./restart -> it */
int writeRestart(SIMULATE *simulate, const char *dir, int restartLink);

#ifdef __cplusplus
}
#endif
#endif
