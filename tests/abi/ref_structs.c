/* Transcription of the three plugin struct declarations of the reference, member for member in the reference's order,
 * with the enums and helper types they need -- potential.h:7-58, integrator.h:4-17, accelerator.h:11,22-31,
 * neighbor.h:42-50 -- for ONE purpose: tests/test_abi.py compares offsetof/sizeof of these with the structs of
 * ddcmd_amd/csrc/host/plugin.h, so that glue compiled against ddcMD's own headers and objects created by either side
 * are interchangeable.  Test data, compiled only by the test. */
#include <stddef.h>
#include <stdio.h>

enum RCUT_ENUMS { RCUT_NONE=0, RCUT_LOCAL=1, RCUT_REMOTE=2, RCUT_ALL=3 } ;
enum NEIGHBORTABLETYPE { NEIGHBORTABLE_NONE=0, NEIGHBORTABLE_SKINNY=1, NEIGHBORTABLE_FAT=2,  NEIGHBORTABLE_GPU=4};
typedef struct rcut_str
{
   double value;
   enum RCUT_ENUMS  mode;
   int type;
} RCUT_TYPE;

enum POTENTIAL_CLASS { NO_POTENTIAL=-1, ZEROPOTENTIAL, MGPT, EAM, EAM1PASS, EAM2PASS, EAM_OPT, EAM_ONEPASS, PAIR, CHARMM, MARTINI, RESTRAINT, EWALD, PLASMA,
                       ORDERSH, ONEBODY, REFLECT, PAIRENERGY, MEAM, MIRRORSYM, LOCALYUKAWA, HYCOP, FMM, GPU_PAIR };
enum POT_COMM_MODE {POT_ONESIDED, POT_TWOSIDED};
typedef struct potential_st
{
   char *name;
   char *objclass;
   char *value;
   char *type;
   void *parent;
   enum POTENTIAL_CLASS itype;
   void (*eval_potential) (void *sys, void *parms, void *e);
   void (*write_dynamics) (void *potential,FILE *file);
   RCUT_TYPE* (*getCutoffs) (void* sys, void* parms, int* nCutoffs);
   enum NEIGHBORTABLETYPE neighborTableType;
   int call_fsumX;
   int use_gpu_list;
   enum POT_COMM_MODE commMode;
   void *parms;
} POTENTIAL;

enum INTEGRATOR_CLASS { NGLF, NGLFNEW, NGLFNK,  NGLFRATTLE, NGLFCONSTRAINT, PNGLF, NGLFTEST, NVEGLF, NVEGLF_SIMPLE, NVTGLF, NPTGLF, STATIC, NEXTFILE, HYCOPINTEGRATOR };
typedef struct integrator_st
{
   char *name;
   char *objclass;
   char *value;
   char *type;
   void  *parent;
   enum INTEGRATOR_CLASS itype;
   int uses_gpu;
   void (*eval_integrator) (void *, void *, void *parm);
   void (*writedynamic) (struct integrator_st *integrator, FILE *file);
   void *parms;
} INTEGRATOR;

enum ACCELERATOR_CLASS { GPU_CUDA };
typedef struct accelerator_st
{
    char *name;
    char *objclass;
    char *value;
    char *type;
    void *parent;
    enum ACCELERATOR_CLASS itype;
    void *parms;
} ACCELERATOR;

#include "layout_table.inc"
