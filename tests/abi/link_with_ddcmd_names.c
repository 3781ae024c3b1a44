/* A stand-in for ddcMD as libddcmi.so meets it at link time: this program DEFINES, with the reference's signatures, the
 * functions whose names ddcMD owns -- nglf (nglf.c:67), ddcenergy (ddcenergy.c:160), kinetic_terms (energy.c:48),
 * eval_energyInfo (energyInfo.c:75), integrator_init (integrator.c:37), accelerator_init / accelerator_getAccelerator
 * (accelerator.c:21,52), potential_init, printinfo, writeRestart (io.c:58), simulate_init, object_get / object_getv /
 * object_compilefile and units_convert (simutil) -- and links with -lddcmi alone (the library used to export those
 * very names).  It must link without duplicate definitions, its own definitions must be the ones it reaches, and the library
 * must never call into them.  With a device it then drives the C-ABI: forces + three NGLF steps of the water box whose arrays
 * the test wrote to argv[1]; tests/test_gpu_abi_link.py compares the printed numbers with the oracle. */
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include "ddcmi.h"

static int stub_calls = 0;
typedef struct { int dummy; } DDC, SIMULATE, SYSTEM, NGLF_PARMS, OBJECT, POTENTIAL, INTEGRATOR, ACCELERATOR, ETYPE;
void nglf(DDC *ddc, SIMULATE *simulate, NGLF_PARMS *p) { (void)ddc; (void)simulate; (void)p; stub_calls++; }
int ddcenergy(DDC *ddc, SYSTEM *sys, int e_eval_flag) { (void)ddc; (void)sys; (void)e_eval_flag; stub_calls++; return -77; }
void kinetic_terms(SYSTEM *sys, int flag) { (void)sys; (void)flag; stub_calls++; }
void eval_energyInfo(SYSTEM *sys) { (void)sys; stub_calls++; }
INTEGRATOR *integrator_init(void *parent, char *name) { (void)parent; (void)name; stub_calls++; return NULL; }
ACCELERATOR *accelerator_init(void *parent, char *name) { (void)parent; (void)name; stub_calls++; return NULL; }
ACCELERATOR *accelerator_getAccelerator(ACCELERATOR *a) { stub_calls++; return a; }
POTENTIAL *potential_init(void *parent, char *name) { (void)parent; (void)name; stub_calls++; return NULL; }
void printinfo(SIMULATE *simulate, ETYPE *e) { (void)simulate; (void)e; stub_calls++; }
void writeRestart(SIMULATE *simulate, int restartLink) { (void)simulate; (void)restartLink; stub_calls++; }
SIMULATE *simulate_init(void *parent, char *name, int comm) { (void)parent; (void)name; (void)comm; stub_calls++; return NULL; }
int object_get(OBJECT *object, char *name, void *ptr, int type, int length, char *dvalue, ...) { (void)object; (void)name; (void)ptr; (void)type; (void)length; (void)dvalue; stub_calls++; return -4242; }
int object_getv(OBJECT *object, char *name, void **ptr, int type, int ignore) { (void)object; (void)name; (void)ptr; (void)type; (void)ignore; stub_calls++; return -4243; }
void object_compilefile(const char *filename) { (void)filename; stub_calls++; }
double units_convert(double value, char *from, char *to) { (void)from; (void)to; stub_calls++; return 4242.0 * value; }

static void *rd(FILE *f, size_t n) { void *p = malloc(n ? n : 1); if (fread(p, 1, n, f) != n) { fprintf(stderr, "short read\n"); exit(3); } return p; }
#define CHK(call) do { int rc_ = (call); if (rc_) { fprintf(stderr, "%s failed (%d): %s\n", #call, rc_, ddcmi_last_error(ctx)); return 4; } } while (0)

int main(int argc, char **argv)
{
   /* our own definitions are the ones this program reaches (no interposition by the library's export table) */
   if (units_convert(1.0, "a", "b") != 4242.0 || object_get(NULL, "x", NULL, 0, 0, NULL) != -4242 || ddcenergy(NULL, NULL, 0) != -77) { printf("interposed\n"); return 2; }
   const int mine = stub_calls;
   ddcmi_ctx *ctx = NULL;
   int rc = ddcmi_create(&ctx, 0);
   if (rc == DDCMI_ENODEVICE) { printf("nodevice stubs_called_by_library %d version %s\n", stub_calls - mine, ddcmi_version()); return 0; }
   if (rc || argc < 2) { fprintf(stderr, "ddcmi_create: %d\n", rc); return 4; }
   FILE *f = fopen(argv[1], "rb");
   if (!f) return 3;
   int *hd = (int *)rd(f, 4 * sizeof(int));
   const int n = hd[0], nsp = hd[1], nlj = hd[2], pbc = hd[3];
   double *sc = (double *)rd(f, 16 * sizeof(double));      /* h[9], rmax, keR, krf, crf, deltaR, dt, (pad) */
   double *mass = (double *)rd(f, nsp * 8), *charge = (double *)rd(f, nsp * 8);
   int *ljtype = (int *)rd(f, nsp * 4), *moltype = (int *)rd(f, nsp * 4);
   double *sigma = (double *)rd(f, nlj * nlj * 8), *eps = (double *)rd(f, nlj * nlj * 8), *shift = (double *)rd(f, nlj * nlj * 8);
   double *r[3], *v[3];
   for (int k = 0; k < 3; k++) r[k] = (double *)rd(f, (size_t)n * 8);
   for (int k = 0; k < 3; k++) v[k] = (double *)rd(f, (size_t)n * 8);
   uint64_t *gid = (uint64_t *)rd(f, (size_t)n * 8);
   int *species = (int *)rd(f, (size_t)n * 4), *group = (int *)rd(f, (size_t)n * 4);
   fclose(f);
   CHK(ddcmi_set_box(ctx, sc, pbc));
   CHK(ddcmi_set_species(ctx, nsp, mass, charge, ljtype, moltype));
   CHK(ddcmi_set_nonbonded(ctx, nlj, sigma, eps, shift, sc[9], sc[10], sc[11], sc[12]));
   CHK(ddcmi_set_molecules(ctx, 0, NULL, NULL, NULL, NULL));
   CHK(ddcmi_set_neighbor(ctx, sc[13], 20));
   int gt = DDCMI_FREE, gi = 1; double gz = 0.0;
   CHK(ddcmi_set_groups(ctx, 1, &gt, &gz, &gz, &gi));
   CHK(ddcmi_upload_state(ctx, n, r[0], r[1], r[2], v[0], v[1], v[2], gid, species, group));
   double e[DDCMI_NE], vir[6], rk, tion[6];
   CHK(ddcmi_eval_forces(ctx, e, vir));
   printf("E0 %.17g %.17g\nV0 %.17g %.17g %.17g %.17g %.17g %.17g\n", e[DDCMI_E_LJ], e[DDCMI_E_TOTAL], vir[0], vir[1], vir[2], vir[3], vir[4], vir[5]);
   double *fo[3];
   for (int k = 0; k < 3; k++) fo[k] = (double *)malloc((size_t)n * 8);
   CHK(ddcmi_download_state(ctx, DDCMI_FORCE, NULL, NULL, NULL, NULL, NULL, NULL, fo[0], fo[1], fo[2]));
   double fs[3] = {0, 0, 0}, fmax = 0;
   for (int i = 0; i < n; i++) for (int k = 0; k < 3; k++) { fs[k] += fo[k][i]; if (fo[k][i] > fmax) fmax = fo[k][i]; }
   printf("F0 %.17g %.17g %.17g %.17g %.17g\n", fo[0][0], fo[1][n / 2], fo[2][n - 1], fmax, fs[0] + fs[1] + fs[2]);
   CHK(ddcmi_step_nglf(ctx, sc[14], 3));
   CHK(ddcmi_get_energies(ctx, e, vir, &rk, tion));
   printf("E3 %.17g %.17g\n", e[DDCMI_E_TOTAL], rk);
   ddcmi_destroy(ctx);
   printf("stubs_called_by_library %d\n", stub_calls - mine);
   return 0;
}
