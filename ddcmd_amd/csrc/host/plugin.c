/* plugin.c -- see plugin.h: ddcMD's plugin surface for the Martini path, in C, on
 * top of the C-ABI.  Reference lines are cited per function. */
#include "plugin.h"
#include <errno.h>
#include "object.h"
#include "units.h"
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <inttypes.h>
#include <sys/stat.h>
#include <unistd.h>
#include <time.h>

static ACCELERATOR *the_accelerator = NULL;

static void die(const char *where, const char *msg)
{
   /* error_action(msg, ERROR_IN(where, ABORT)) in the reference: print and exit */
   fprintf(stderr, "%s: %s\n", where, msg);
   exit(1);
}

/* ------------------------------------------------------------------------- */
/* More than one rank: what ddcMD takes from MPI and the DDC object (ddc.c: lx ly lz, COMM_LOCAL rank/size).
 * The ranks of a launch are started by any launcher that hands each process RANK / WORLD_SIZE / LOCAL_RANK
 * (python -m torch.distributed.run, or a shell loop); they meet over libddcmi's TCP rendezvous and exchange
 * halos over RCCL (one GPU per rank) or, with DDCMI_TRANSPORT=host, over the rendezvous' streams (ranks that
 * share a GPU).  Rank 0 owns stdout, the data file and the restart files; STATE on rank 0 is the gathered
 * global state at print / checkpoint steps, exactly what it is on one rank. */
typedef struct { int rank, world, local_rank, grid[3], host_transport; ddcmi_rdzv *rdzv; } PARENV;
static PARENV par = {0, 1, 0, {1, 1, 1}, 0, NULL};
static int env_int(const char *name, int dflt) { const char *v = getenv(name); return (v && *v) ? atoi(v) : dflt; }
/* the process grid: the deck's ddc DDC { lx ly lz } when it multiplies to the number of ranks; otherwise WORLD_SIZE factored
 * over the axes, smallest prime factors first, each onto the axis whose bricks are widest at that point (2 -> 2x1x1,
 * 4 -> 2x2x1, 8 -> 2x2x2 for a cubic box: bench.py's grids) */
void plugin_plan_grid(int world, int lx, int ly, int lz, const double h[9], int grid[3])
{
   grid[0] = grid[1] = grid[2] = 1;
   if (world <= 1) return;
   if (lx > 0 && ly > 0 && lz > 0 && (long)lx * ly * lz == world) { grid[0] = lx; grid[1] = ly; grid[2] = lz; return; }
   const double L[3] = {h[0], h[4], h[8]};
   for (int f = 2, w = world; w > 1;)
   {
      if (w % f) { f++; continue; }
      int a = 0;
      for (int b = 1; b < 3; b++) if (L[b] / grid[b] > L[a] / grid[a] * (1.0 + 1e-12)) a = b;
      grid[a] *= f; w /= f;
   }
}
static void parallel_init(const ddcmi_setup *s)
{
   par.rank = env_int("RANK", 0); par.world = env_int("WORLD_SIZE", 1); par.local_rank = env_int("LOCAL_RANK", par.rank);
   par.grid[0] = par.grid[1] = par.grid[2] = 1;
   if (par.world <= 1) { par.world = 1; par.rank = 0; return; }
   if (par.rank < 0 || par.rank >= par.world) die("parallel_init", "RANK outside 0..WORLD_SIZE-1");
   setenv("HSA_ENABLE_IPC_MODE_LEGACY", "0", 0);      /* RCCL between processes: dmabuf IPC (read when the HIP runtime initialises, below) */
   plugin_plan_grid(par.world, s->lx, s->ly, s->lz, s->h, par.grid);
   const char *addr = getenv("MASTER_ADDR");
   if (!addr || !*addr) addr = "127.0.0.1";
   char pf[1024];
   const char *file = getenv("DDCMI_RDZV_FILE");
   if (!file || !*file)
   {
      /* one name per launch: the workers of a launch share the launcher as parent (the same key bench.py uses) */
      const char *tmp = getenv("TMPDIR"), *mp = getenv("MASTER_PORT"), *rid = getenv("TORCHELASTIC_RUN_ID");
      snprintf(pf, sizeof(pf), "%s/ddcmi_rdzv_%s_%s_%s_%d", (tmp && *tmp) ? tmp : "/tmp", addr, (mp && *mp) ? mp : "0", (rid && *rid) ? rid : "none", (int)getppid());
      for (char *c = pf + strlen((tmp && *tmp) ? tmp : "/tmp") + 1; *c; c++) if (*c == '/') *c = '_';
      file = pf;
   }
   if (ddcmi_rdzv_create(&par.rdzv, par.rank, par.world, addr, env_int("DDCMI_RDZV_PORT", 0), file, 300.0) != DDCMI_OK)
      die("parallel_init", ddcmi_rdzv_last_error(NULL));
   const char *tr = getenv("DDCMI_TRANSPORT");
   par.host_transport = tr && strcmp(tr, "host") == 0;
}
/* the first real exchange of a launch is a checked one (ddcmi_comm_preflight): a fabric that does not carry one of the brick's links, a
 * rank on the wrong device or a stale id ends the launch here, on every rank, with the stage / peer / direction in the message --
 * not as a hang inside the first step's ddcUpdate (ddcUpdate.c:56-85) */
static void parallel_preflight(ddcmi_ctx *ctx)
{
   int64_t rep[16];
   if (ddcmi_comm_preflight(ctx, (double)env_int("DDCMI_PREFLIGHT_TIMEOUT", 60), rep) != DDCMI_OK) die("ddcmi_comm_preflight", ddcmi_last_error(ctx));
   if (par.rank == 0 && getenv("DDCMI_VERBOSE"))
      fprintf(stderr, "ddcmi_md: communicator preflight ok: %d peers, %d directions x %d bytes, all-reduce, all-gather in %.1f ms\n", (int)rep[0], (int)rep[1], (int)rep[2], rep[7] * 1e-3);
}
static void parallel_comm_init(ddcmi_ctx *ctx)
{
   if (par.world <= 1) return;
   if (par.host_transport)
   {
      if (ddcmi_comm_init_host(ctx, par.rdzv, par.grid[0], par.grid[1], par.grid[2]) != DDCMI_OK) die("ddcmi_comm_init_host", ddcmi_last_error(ctx));
      parallel_preflight(ctx);
      return;
   }
   char id[128];
   memset(id, 0, sizeof(id));
   if (par.rank == 0 && ddcmi_comm_unique_id(id) != DDCMI_OK) die("ddcmi_comm_unique_id", ddcmi_last_error(ctx));
   if (ddcmi_rdzv_bcast(par.rdzv, id, sizeof(id), 0) != DDCMI_OK) die("parallel_comm_init", ddcmi_rdzv_last_error(par.rdzv));      /* MPI_Bcast of the RCCL id */
   if (ddcmi_comm_init(ctx, par.rank, par.world, id, par.grid[0], par.grid[1], par.grid[2]) != DDCMI_OK) die("ddcmi_comm_init", ddcmi_last_error(ctx));
   parallel_preflight(ctx);
}
/* owner of a bead in the px x py x pz brick decomposition of a box centred on the origin (domain.c:191-208 for a cubic
 * lattice of domain centres) */
static int brick_of(const ddcmi_setup *s, double x, double y, double z)
{
   const double r[3] = {x, y, z}, L[3] = {s->h[0], s->h[4], s->h[8]};
   int out = 0, mult = 1;
   for (int a = 0; a < 3; a++)
   {
      const int P = par.grid[a];
      double w = r[a] - L[a] * rint(r[a] / L[a]);
      int b = (int)floor((w + 0.5 * L[a]) / (L[a] / P));
      if (b < 0) b = 0;
      if (b > P - 1) b = P - 1;
      out += mult * b; mult *= P;
   }
   return out;
}


/* accelerator_init, accelerator.c:10-56 */
ACCELERATOR *accelerator_init(void *parent, const char *name, const char *type)
{
   (void)parent;
   ACCELERATOR *a = calloc(1, sizeof(ACCELERATOR));
   a->name = strdup(name ? name : "accelerator");
   a->type = strdup(type ? type : "HIP");
   if (strcmp(a->type, "CUDA") == 0) a->itype = GPU_CUDA;          /* existing decks say CUDA: accepted, runs on HIP */
   else if (strcmp(a->type, "HIP") == 0) a->itype = GPU_HIP;
   else die("accelerator_init", "ACCELERATOR type must be CUDA or HIP");
   ddcmi_ctx *ctx = NULL;
   /* one GPU per rank (LOCAL_RANK) unless the ranks are told to share one (tests on a one-GPU box) */
   int dev = (par.world > 1 && !env_int("DDCMI_SINGLE_DEVICE", 0)) ? par.local_rank : 0;
   {
      /* a launcher that shows every rank exactly ONE device: ordinal 0 there.  Fewer devices than ranks otherwise is an error, not a
       * reason to put two ranks on one GPU silently (a bogus scaling number in the making) */
      const int ndev = ddcmi_device_count();
      if (ndev == 1) dev = 0;
      else if (ndev > 1 && dev >= ndev && !getenv("DDCMI_DEVICE")) die("accelerator_init", "LOCAL_RANK is beyond the visible devices: start one rank per GPU (or set DDCMI_SINGLE_DEVICE=1 / DDCMI_DEVICE for tests that share one)");
   }
   const char *env = getenv("DDCMI_DEVICE");
   if (env) dev = atoi(env);
   if (ddcmi_create(&ctx, dev) != DDCMI_OK) die("accelerator_init", ddcmi_last_error(NULL));
   a->parms = ctx;
   the_accelerator = a;
   return a;
}
ACCELERATOR *accelerator_getAccelerator(ACCELERATOR *a) { return a ? a : the_accelerator; }

/* ------------------------------------------------------------------------- */
/* bonded term lists over particle indices: charmmResidues (bioCharmmCovalent.c:48-93)
 * sorts by gid and cuts residue runs; term atoms are offsets inside the run */
typedef struct { uint64_t gid; int id; } gid_order;
static int cmp_gid(const void *a, const void *b)
{
   uint64_t ga = ((const gid_order *)a)->gid, gb = ((const gid_order *)b)->gid;
   return (ga > gb) - (ga < gb);
}
static int martiniBondHIPParms(ddcmi_ctx *ctx, const ddcmi_setup *s)
{
   const uint64_t molResMask = DDCMI_GID_MOLRESMASK;
   int n = s->natoms;
   int nb_tot = s->bond_off[s->nresi], na_tot = s->angle_off[s->nresi], nt_tot = s->tors_off[s->nresi];
   if (nb_tot + na_tot + nt_tot == 0)
      return ddcmi_set_bonded(ctx, 0, NULL, NULL, NULL, 0, NULL, NULL, NULL, NULL, 0, NULL, NULL, NULL, NULL, NULL, s->excludePotentialTerm);
   gid_order *ord = malloc(sizeof(gid_order) * n);
   for (int i = 0; i < n; i++) { ord[i].id = i; ord[i].gid = s->gid[i]; }
   qsort(ord, n, sizeof(gid_order), cmp_gid);
   /* count */
   size_t nb = 0, na = 0, nt = 0;
   for (int first = 0; first < n;)
   {
      uint64_t key = ord[first].gid & molResMask;
      int last = first;
      while (last < n && (ord[last].gid & molResMask) == key) last++;
      int rt = s->resitype[s->species[ord[first].id]];
      if (last - first != s->resi_natoms[rt]) { free(ord); return DDCMI_EINVAL; }
      nb += s->bond_off[rt + 1] - s->bond_off[rt]; na += s->angle_off[rt + 1] - s->angle_off[rt]; nt += s->tors_off[rt + 1] - s->tors_off[rt];
      first = last;
   }
   int *bij = malloc(sizeof(int) * (2 * nb + 2)), *aijk = malloc(sizeof(int) * (3 * na + 3)), *af = malloc(sizeof(int) * (na + 1));
   int *tijkl = malloc(sizeof(int) * (4 * nt + 4)), *tf = malloc(sizeof(int) * (nt + 1)), *tn = malloc(sizeof(int) * (nt + 1));
   /* more than one rank: the atoms of a term are named by gid (ddcmi_set_bonded_gid): beads migrate one by one */
   const int by_gid = par.world > 1;
   uint64_t *bg = by_gid ? malloc(sizeof(uint64_t) * (2 * nb + 2)) : NULL, *ag = by_gid ? malloc(sizeof(uint64_t) * (3 * na + 3)) : NULL,
            *tg = by_gid ? malloc(sizeof(uint64_t) * (4 * nt + 4)) : NULL;
   double *kb = malloc(sizeof(double) * (nb + 1)), *b0 = malloc(sizeof(double) * (nb + 1));
   double *ak = malloc(sizeof(double) * (na + 1)), *a0 = malloc(sizeof(double) * (na + 1));
   double *tk = malloc(sizeof(double) * (nt + 1)), *td = malloc(sizeof(double) * (nt + 1));
   nb = na = nt = 0;
   for (int first = 0; first < n;)
   {
      uint64_t key = ord[first].gid & molResMask;
      int last = first;
      while (last < n && (ord[last].gid & molResMask) == key) last++;
      int rt = s->resitype[s->species[ord[first].id]];
#define AT(off) (ord[first + (off)].id)
      for (int b = s->bond_off[rt]; b < s->bond_off[rt + 1]; b++, nb++)
      { bij[2 * nb] = AT(s->bondI[b]); bij[2 * nb + 1] = AT(s->bondJ[b]); kb[nb] = s->bond_kb[b]; b0[nb] = s->bond_b0[b]; }
      for (int a = s->angle_off[rt]; a < s->angle_off[rt + 1]; a++, na++)
      { aijk[3 * na] = AT(s->angleI[a]); aijk[3 * na + 1] = AT(s->angleJ[a]); aijk[3 * na + 2] = AT(s->angleK[a]); af[na] = s->angle_func[a]; ak[na] = s->angle_k[a]; a0[na] = s->angle_t0[a]; }
      for (int t = s->tors_off[rt]; t < s->tors_off[rt + 1]; t++, nt++)
      { tijkl[4 * nt] = AT(s->torsI[t]); tijkl[4 * nt + 1] = AT(s->torsJ[t]); tijkl[4 * nt + 2] = AT(s->torsK[t]); tijkl[4 * nt + 3] = AT(s->torsL[t]);
        tf[nt] = s->tors_func[t]; tn[nt] = s->tors_n[t]; tk[nt] = s->tors_k[t]; td[nt] = s->tors_delta[t]; }
#undef AT
      first = last;
   }
   int rc;
   if (by_gid)
   {
      for (size_t k = 0; k < 2 * nb; k++) bg[k] = s->gid[bij[k]];
      for (size_t k = 0; k < 3 * na; k++) ag[k] = s->gid[aijk[k]];
      for (size_t k = 0; k < 4 * nt; k++) tg[k] = s->gid[tijkl[k]];
      rc = ddcmi_set_bonded_gid(ctx, (int)nb, bg, kb, b0, (int)na, ag, af, ak, a0, (int)nt, tg, tf, tn, tk, td, s->excludePotentialTerm);
      free(bg); free(ag); free(tg);
   }
   else rc = ddcmi_set_bonded(ctx, (int)nb, bij, kb, b0, (int)na, aijk, af, ak, a0, (int)nt, tijkl, tf, tn, tk, td, s->excludePotentialTerm);
   free(ord); free(bij); free(aijk); free(af); free(tijkl); free(tf); free(tn); free(kb); free(b0); free(ak); free(a0); free(tk); free(td);
   return rc;
}

/* martini_parms (bioMartini.c:1210-1353): upload what martiniNonBondGPUParms /
 * martiniBondGPUParms upload in the reference */
static void martini_parms(POTENTIAL *potential, SIMULATE *simulate)
{
   const ddcmi_setup *s = simulate->setup;
   ACCELERATOR *accelerator = accelerator_getAccelerator(NULL);
   if (!accelerator) die("martini_parms", "no ACCELERATOR: this library has no CPU force path");
   ddcmi_ctx *ctx = accelerator->parms;
   int gtype[32];
   int rc = 0;
   rc |= ddcmi_set_box(ctx, s->h, s->pbc);
   rc |= ddcmi_set_species(ctx, s->nspecies, s->mass, s->charge, s->ljtype, s->moltype);
   rc |= ddcmi_set_nonbonded(ctx, s->nlj, s->sigma, s->eps, s->shift, s->rmax, s->keR, s->krf, s->crf);
   rc |= ddcmi_set_molecules(ctx, s->nmoltype, s->mol_nspecies, s->bpair_off, s->bpairI, s->bpairJ);
   rc |= ddcmi_set_neighbor(ctx, s->deltaR, s->updateRate);
   for (int g = 0; g < s->ngroup && g < 32; g++)
   {
      if (s->group_type[g] == DDCMI_GROUP_FREE) gtype[g] = DDCMI_FREE;
      else if (s->group_type[g] == DDCMI_GROUP_BERENDSEN) gtype[g] = DDCMI_BERENDSEN;
      else if (s->group_type[g] == DDCMI_GROUP_LANGEVIN) gtype[g] = DDCMI_LANGEVIN;
      else die("group_init", "only FREE, BERENDSEN and LANGEVIN groups are supported on this path");
   }
   rc |= ddcmi_set_groups(ctx, s->ngroup, gtype, s->group_Teq, s->group_tau, s->group_interval);
   if (s->group_vcm) rc |= ddcmi_set_group_vcm(ctx, s->ngroup, s->group_vcm);
   rc |= ddcmi_set_random(ctx, s->rng_seed);
   if (s->nrest > 0)      /* restraint_parms + restraintGPU_parms (restraint.c:177-208) */
      rc |= ddcmi_set_restraints(ctx, s->nrest, s->rest_gid, s->rest_fc, s->rest_r0, s->rest_kb, s->rest_origin);
   rc |= martiniBondHIPParms(ctx, s);
   rc |= ddcmi_set_clock(ctx, s->loop, s->time);
   if (rc) die("martini_parms", ddcmi_last_error(ctx));
   if (par.rank == 0) printf("using HIP martini parms\n");                 /* bioMartini.c:1339 prints "using gpu martini parms" */
   MARTINIHIP_PARMS *parms = calloc(1, sizeof(MARTINIHIP_PARMS));
   parms->ctx = ctx; parms->rmax = s->rmax; parms->simulate = simulate;
   potential->itype = MARTINI;
   potential->use_gpu_list = 1;
   potential->call_fsumX = 0;
   potential->commMode = POT_TWOSIDED;
   potential->neighborTableType = NEIGHBORTABLE_GPU;
   potential->eval_potential = (void (*)(void *, void *, void *))martiniHIP;
   potential->getCutoffs = (RCUT_TYPE * (*)(void *, void *, int *))martiniCutoff;
   potential->write_dynamics = NULL;
   potential->parms = parms;
}

/* charmmCutoff (bioCharmm.c:386-414): one cutoff, rmax, first vertex of a pair local */
RCUT_TYPE *martiniCutoff(SYSTEM *sys, void *parms_, int *n)
{
   (void)sys;
   MARTINIHIP_PARMS *parms = parms_;
   parms->rcut[0].value = parms->rmax; parms->rcut[0].mode = RCUT_LOCAL; parms->rcut[0].type = -1;
   parms->rcut[1].value = parms->rmax; parms->rcut[1].mode = RCUT_ALL; parms->rcut[1].type = -1;
   *n = 1;
   return parms->rcut;
}

POTENTIAL *potential_init(void *parent, const char *name, const char *type)
{
   if (strcmp(type, "MARTINI") != 0) die("potential_init", "only POTENTIAL type=MARTINI is implemented (potential.c:181-188)");
   POTENTIAL *p = calloc(1, sizeof(POTENTIAL));
   p->name = strdup(name); p->type = strdup(type); p->parent = parent;
   martini_parms(p, (SIMULATE *)parent);
   return p;
}

/* martiniGPU1 (bioMartini.cu:146-171) / martini (bioMartini.c:1357-1390):
 * accumulate into e->eion and e->virial */
void martiniHIP(SYSTEM *sys, void *parms_, ETYPE *e)
{
   MARTINIHIP_PARMS *parms = parms_;
   ddcmi_ctx *ctx = parms->ctx;
   SIMULATE *sim = parms->simulate;
   /* gpu_integrate == 0 (bioMartini.cu:153-166): the integrator runs on the host over STATE -- positions go up
    * before the evaluation (sendGPUState) and the forces come back, ACCUMULATED into state->f like every
    * potential's (the POTENTIAL contract: ddcenergy.c:212 after zeroAll) */
   const int gpu_integrate = (sim && sim->integrator) ? sim->integrator->uses_gpu : 1;
   STATE *st = sys->state;
   if (!gpu_integrate)
   {
      if (ddcmi_upload_positions(ctx, st->rx, st->ry, st->rz, st->vx, st->vy, st->vz) != DDCMI_OK) die("martiniHIP", ddcmi_last_error(ctx));
      /* constructList on rebuild steps (ddcUpdateAll.c:64-71,136-139) */
      if (sim->ddc->updateRate == 0 || sys->loop % sim->ddc->updateRate == 0)
         if (ddcmi_build_list(ctx) != DDCMI_OK) die("martiniHIP", ddcmi_last_error(ctx));
   }
   double en[DDCMI_NE], vir[6];
   if (ddcmi_eval_forces(ctx, en, vir) != DDCMI_OK) die("martiniHIP", ddcmi_last_error(ctx));
   if (!gpu_integrate)
   {
      int n = st->nlocal;
      double *f = malloc(sizeof(double) * 3 * (size_t)(n > 0 ? n : 1));
      if (ddcmi_download_state(ctx, DDCMI_FORCE, NULL, NULL, NULL, NULL, NULL, NULL, f, f + n, f + 2 * (size_t)n) != DDCMI_OK) die("martiniHIP", ddcmi_last_error(ctx));
      for (int i = 0; i < n; i++) { st->fx[i] += f[i]; st->fy[i] += f[n + i]; st->fz[i] += f[2 * (size_t)n + i]; }
      free(f);
   }
   e->eion += en[DDCMI_E_TOTAL];
   e->virial.xx += vir[DDCMI_XX]; e->virial.yy += vir[DDCMI_YY]; e->virial.zz += vir[DDCMI_ZZ];
   e->virial.xy += vir[DDCMI_XY]; e->virial.xz += vir[DDCMI_XZ]; e->virial.yz += vir[DDCMI_YZ];
}

/* nglfconstraint_parms (nglfconstraint.c:86-120) for the device: the constraint groups of every
 * residue instance (genConstraint, bioMartini.c:300-445: one group per CONSLISTPARMS, pairs in deck
 * order) and the molecule lists the barostat's molecular virial runs over (molecularPressure.c:23-56) */
static void nglfconstraintHIP_parms(ddcmi_ctx *ctx, const ddcmi_setup *s)
{
   const uint64_t molResMask = DDCMI_GID_MOLRESMASK, molMask = DDCMI_GID_MOLMASK;
   int n = s->natoms;
   gid_order *ord = malloc(sizeof(gid_order) * (n > 0 ? n : 1));
   for (int i = 0; i < n; i++) { ord[i].id = i; ord[i].gid = s->gid[i]; }
   qsort(ord, n, sizeof(gid_order), cmp_gid);
   if (s->nresicons > 0 && s->cons_off && s->cons_off[s->nresi] > 0)
   {
      size_t ng = 0, np = 0;
      for (int pass = 0; pass < 2; pass++)
      {
         int *poff = NULL, *pi = NULL, *pj = NULL; double *dd = NULL;
         if (pass == 1)
         {
            poff = malloc(sizeof(int) * (ng + 1)); pi = malloc(sizeof(int) * (np + 1)); pj = malloc(sizeof(int) * (np + 1)); dd = malloc(sizeof(double) * (np + 1));
            ng = np = 0;
         }
         for (int first = 0; first < n;)
         {
            uint64_t key = ord[first].gid & molResMask;
            int last = first;
            while (last < n && (ord[last].gid & molResMask) == key) last++;
            int rt = s->resitype[s->species[ord[first].id]];
            if (last - first != s->resi_natoms[rt]) die("nglfconstraint_parms", "incomplete residue in the particle set");
            for (int c0 = s->cons_off[rt]; c0 < s->cons_off[rt + 1];)
            {
               int c1 = c0;
               while (c1 < s->cons_off[rt + 1] && s->cons_grp[c1] == s->cons_grp[c0]) c1++;
               if (pass == 1)
               {
                  poff[ng] = (int)np;
                  for (int c = c0; c < c1; c++) { pi[np + c - c0] = ord[first + s->consI[c]].id; pj[np + c - c0] = ord[first + s->consJ[c]].id; dd[np + c - c0] = s->cons_r0[c]; }
               }
               ng++; np += c1 - c0;
               c0 = c1;
            }
            first = last;
         }
         if (pass == 1)
         {
            poff[ng] = (int)np;
            if (par.world > 1)
            {
               uint64_t *gi = malloc(sizeof(uint64_t) * (np + 1)), *gj = malloc(sizeof(uint64_t) * (np + 1));
               for (size_t k = 0; k < np; k++) { gi[k] = s->gid[pi[k]]; gj[k] = s->gid[pj[k]]; }
               if (ddcmi_set_constraints_gid(ctx, (int)ng, poff, gi, gj, dd) != DDCMI_OK) die("nglfconstraint_parms", ddcmi_last_error(ctx));
               free(gi); free(gj);
            }
            else if (ddcmi_set_constraints(ctx, (int)ng, poff, pi, pj, dd) != DDCMI_OK) die("nglfconstraint_parms", ddcmi_last_error(ctx));
            free(poff); free(pi); free(pj); free(dd);
         }
      }
   }
   if (s->npt_beta > 0.0)
   {
      long nmol = 0; int nmulti = 0, natm = 0;
      for (int first = 0; first < n;)
      {
         int last = first;
         while (last < n && (ord[last].gid & molMask) == (ord[first].gid & molMask)) last++;
         nmol++;
         if (last - first >= 2) { nmulti++; natm += last - first; }
         first = last;
      }
      int *moff = malloc(sizeof(int) * (nmulti + 1)), *matm = malloc(sizeof(int) * (natm + 1));
      nmulti = natm = 0;
      for (int first = 0; first < n;)
      {
         int last = first;
         while (last < n && (ord[last].gid & molMask) == (ord[first].gid & molMask)) last++;
         if (last - first >= 2) { moff[nmulti++] = natm; for (int k = first; k < last; k++) matm[natm++] = ord[k].id; }
         first = last;
      }
      moff[nmulti] = natm;
      if (ddcmi_set_barostat(ctx, s->npt_T, s->npt_P0, s->npt_beta, s->npt_tau) != DDCMI_OK) die("nglfconstraint_parms", ddcmi_last_error(ctx));
      if (par.world > 1)
      {
         uint64_t *mg = malloc(sizeof(uint64_t) * (natm + 1));
         double *mm = calloc(nmulti + 1, sizeof(double));      /* total mass of each multi-bead molecule */
         for (int k = 0; k < natm; k++) mg[k] = s->gid[matm[k]];
         for (int m = 0; m < nmulti; m++) for (int k = moff[m]; k < moff[m + 1]; k++) mm[m] += s->mass[s->species[matm[k]]];
         if (ddcmi_set_molecule_lists_gid(ctx, nmol, nmulti, moff, mg, mm) != DDCMI_OK) die("nglfconstraint_parms", ddcmi_last_error(ctx));
         free(mg); free(mm);
      }
      else if (ddcmi_set_molecule_lists(ctx, nmol, nmulti, moff, matm) != DDCMI_OK) die("nglfconstraint_parms", ddcmi_last_error(ctx));
      free(moff); free(matm);
   }
   free(ord);
}

/* integrator_init, integrator.c:37-167 */
INTEGRATOR *integrator_init(void *parent, const char *name, const char *type)
{
   INTEGRATOR *in = calloc(1, sizeof(INTEGRATOR));
   in->name = strdup(name); in->type = strdup(type); in->parent = parent;
   const ddcmi_setup *su = parent ? ((SIMULATE *)parent)->setup : NULL;
   const int gpulang = su && strcmp(type, "NGLFGPULANGEVIN") == 0;
   const int npt_ok = su && (strcmp(type, "NGLFCONSTRAINT") == 0 || gpulang);
   if (gpulang)
   {
      /* nglfGPULangevin (nglfGPU.cu:422-507): EVERY bead gets the Langevin update with the parameters of the
       * first group (:469-473), whatever the GROUP objects say; the barostat is the isotropic one.  The
       * reference leaves its T/P0/beta/tauBarostat uninitialised (nglf_parms, nglfGPU.cu:41-50): here they
       * are read from the INTEGRATOR object like nglfconstraint_parms does. */
      ddcmi_ctx *ctx = accelerator_getAccelerator(NULL)->parms;
      if (su->ngroup < 1 || su->group_type[0] != DDCMI_GROUP_LANGEVIN) die("integrator_init", "NGLFGPULANGEVIN needs the first GROUP to be of type LANGEVIN");
      int gt[32]; double gT[32], gtau[32]; int giv[32];
      for (int g = 0; g < su->ngroup && g < 32; g++) { gt[g] = DDCMI_LANGEVIN; gT[g] = su->group_Teq[0]; gtau[g] = su->group_tau[0]; giv[g] = 1; }
      if (ddcmi_set_groups(ctx, su->ngroup, gt, gT, gtau, giv) != DDCMI_OK || ddcmi_set_barostat_isotropic(ctx, 1) != DDCMI_OK) die("integrator_init", ddcmi_last_error(ctx));
   }
   if (npt_ok || strcmp(type, "NGLF") == 0 || strcmp(type, "NVTGLF") == 0 || strcmp(type, "NGLFGPU") == 0 || strcmp(type, "NGLFHIP") == 0)
   {
      /* NGLFCONSTRAINT = nglf + velocity constraints + the barostat of changeVolume (nglfconstraint.c:510-574) */
      if (npt_ok) nglfconstraintHIP_parms(accelerator_getAccelerator(NULL)->parms, su);
      in->itype = npt_ok ? NGLFCONSTRAINT : (strcmp(type, "NVTGLF") == 0 ? NVTGLF : NGLF);
      in->eval_integrator = (void (*)(void *, void *, void *))nglfHIP;
      in->uses_gpu = 1;                                    /* state stays on the device between print steps (masters.c:389-403) */
      const char *cpu = getenv("DDCMI_CPU_INTEGRATOR");
      if (cpu && atoi(cpu) != 0 && (strcmp(type, "NGLF") == 0 || strcmp(type, "NVTGLF") == 0))
      {
         /* the reference's own pairing for these type strings (integrator.c:59-64): nglf() on the host, the
          * potential on the accelerator.  The default keeps the whole step on the device. */
         in->eval_integrator = (void (*)(void *, void *, void *))nglf;
         in->uses_gpu = 0;
         printf("INTEGRATOR %s on the host (nglf.c), potential on the accelerator\n", type);
      }
   }
   else
   {
      char msg[256];
      snprintf(msg, sizeof(msg), "INTEGRATOR type %s is not on this path (NGLF, NVTGLF, NGLFGPU, NGLFHIP, NGLFGPULANGEVIN, NGLFCONSTRAINT are)", type);
      die("integrator_init", msg);
   }
   return in;
}

/* nglfGPU (nglfGPU.cu:511) with the nglf.c:67-112 contract: advance loop/time */
void nglfHIP(DDC *ddc, SIMULATE *simulate, void *parms)
{
   (void)parms;
   SYSTEM *sys = simulate->system;
   ddcmi_ctx *ctx = accelerator_getAccelerator(NULL)->parms;
   if (ddcmi_step_nglf(ctx, simulate->dt, 1) != DDCMI_OK) die("nglfHIP", ddcmi_last_error(ctx));
   ddc->update = 0;
   simulate->time += simulate->dt;
   simulate->loop++;
   sys->loop = simulate->loop;
   sys->time = simulate->time;
}

static int host_integrated(const SYSTEM *sys)
{
   const SIMULATE *sim = (sys->npotential > 0 && sys->potential[0]->parms) ? ((MARTINIHIP_PARMS *)sys->potential[0]->parms)->simulate : NULL;
   return sim && sim->integrator && !sim->integrator->uses_gpu;
}
/* free_velocityUpdate (free.c:13-28) / berendsen_velocityUpdate (berendsen.c:64-89) / berendsen_Update (:30-62) on STATE */
static void group_velocityUpdate(int front, int k, GROUP *g, STATE *st, double dt_half)
{
   const double a = dt_half / st->species[k]->mass;
   if (g->itype == BERENDSEN && front && g->doScaling) { st->vx[k] *= g->lambda; st->vy[k] *= g->lambda; st->vz[k] *= g->lambda; }
   st->vx[k] += a * st->fx[k]; st->vy[k] += a * st->fy[k]; st->vz[k] += a * st->fz[k];
}
static void group_Update_front(GROUP *g, int64_t loop, double dt_half)
{
   if (g->itype != BERENDSEN) return;
   g->Tsum += g->energyInfo.temperature; g->nT += 1;
   const double Tave = g->Tsum / g->nT, ratio = (Tave == 0) ? 0 : g->Teq / Tave;
   g->lambda = (g->tau != 0) ? sqrt(1 + (2.0 * dt_half / g->tau) * (ratio - 1)) : sqrt(ratio);
   g->doScaling = 0;
   if (loop % g->interval == 0) { g->Tsum = 0; g->nT = 0; g->doScaling = 1; }
}
/* nglf (nglf.c:67-112): FRONT half kick, drift, backInBox_fast, ddcenergy, BACK half kick, kinetic_terms, group Update */
void nglf(DDC *ddc, SIMULATE *simulate, void *parms)
{
   (void)parms;
   const double dt = simulate->dt;
   SYSTEM *sys = simulate->system;
   STATE *st = sys->state;
   const double L[3] = {sys->box->h0[0], sys->box->h0[4], sys->box->h0[8]};
   for (int g = 0; g < sys->ngroup; g++)
      if (sys->group[g]->itype != FREE && sys->group[g]->itype != BERENDSEN) die("nglf", "the host integrator handles FREE and BERENDSEN groups");
   for (int k = 0; k < st->nlocal; k++) group_velocityUpdate(1, k, st->group[k], st, 0.5 * dt);
   for (int k = 0; k < st->nlocal; k++)
   {
      double r[3] = {st->rx[k] + dt * st->vx[k], st->ry[k] + dt * st->vy[k], st->rz[k] + dt * st->vz[k]};
      for (int a = 0; a < 3; a++)      /* PreduceOrthorhombicB7_OneLatticeReduction (preduce.c:147-160) */
         if (sys->box->pbc >> a & 1) { if (r[a] > 0.5 * L[a]) r[a] -= L[a]; if (r[a] < -0.5 * L[a]) r[a] += L[a]; }
      st->rx[k] = r[0]; st->ry[k] = r[1]; st->rz[k] = r[2];
   }
   ddc->update = 0;
   simulate->time += dt;
   simulate->loop++;
   sys->loop = simulate->loop;
   sys->time = simulate->time;
   if (ddcenergy(ddc, sys, 0) != 0) return;
   for (int k = 0; k < st->nlocal; k++) group_velocityUpdate(0, k, st->group[k], st, 0.5 * dt);
   kinetic_terms(sys, 1);
   for (int g = 0; g < sys->ngroup; g++) group_Update_front(sys->group[g], simulate->loop, 0.5 * dt);
}

/* ddcenergy (ddcenergy.c:160-238) for the accelerated path: zero ETYPE, potentials */
int ddcenergy(DDC *ddc, SYSTEM *sys, int e_eval_flag)
{
   (void)ddc;
   ETYPE *e = &sys->energyInfo;
   e->eion = 0.0;
   memset(&e->virial, 0, sizeof(e->virial));
   if (host_integrated(sys))
   {
      /* zeroParticle: the reference skips it whenever an ACCELERATOR exists (ddcenergy.c:149-158) because its device
       * potentials never write the host forces; a host integrator reads them, so they start from zero here */
      STATE *st = sys->state;
      memset(st->fx, 0, sizeof(double) * st->nlocal); memset(st->fy, 0, sizeof(double) * st->nlocal); memset(st->fz, 0, sizeof(double) * st->nlocal);
   }
   for (int i = 0; i < sys->npotential; i++) sys->potential[i]->eval_potential(sys, sys->potential[i]->parms, e);
   if (e_eval_flag) { kinetic_terms(sys, 1); eval_energyInfo(sys); }
   return 0;
}

/* kinetic_terms (energy.c:48-163): rk, tion, the thermal flux and the per-group / per-species copies (rk, tion, mass,
 * number) from the device reductions.  The per-atom potentialEnergy and sion the reference reads for J are what the
 * potentials left per atom: nothing on this path (bioMartini.c:1111-1120 books e->eion / e->virial only), so J = sum K v
 * and the copies' eion stay 0. */
static void etype_from_row(ETYPE *g, const double *r)
{
   g->rk = r[0];
   g->tion.xx = r[1]; g->tion.yy = r[2]; g->tion.zz = r[3]; g->tion.xy = r[4]; g->tion.xz = r[5]; g->tion.yz = r[6];
   g->mass = r[7]; g->number = r[8];
   g->thermal_flux.x = r[9]; g->thermal_flux.y = r[10]; g->thermal_flux.z = r[11];
   g->eion = 0.0;
}
void kinetic_terms(SYSTEM *sys, int flag)
{
   (void)flag;
   ddcmi_ctx *ctx = accelerator_getAccelerator(NULL)->parms;
   ETYPE *e = &sys->energyInfo;
   double tion[6];
   const int ncl = sys->ngroup + sys->nspecies;
   double *rows = (double *)calloc((size_t)12 * (ncl > 0 ? ncl : 1), sizeof(double));      /* [groups..., species...][12] */
   if (!rows) die("kinetic_terms", "out of memory");
   if (host_integrated(sys))
   {
      /* energy.c:48-163 over STATE */
      STATE *st = sys->state;
      double rk = 0.0;
      memset(tion, 0, sizeof(tion));
      for (int k = 0; k < st->nlocal; k++)
      {
         const double m = st->species[k]->mass, x = st->vx[k], y = st->vy[k], z = st->vz[k];
         const double K = 0.5 * m * (x * x + y * y + z * z);
         rk += K;
         tion[DDCMI_XX] += m * x * x; tion[DDCMI_YY] += m * y * y; tion[DDCMI_ZZ] += m * z * z;
         tion[DDCMI_XY] += m * x * y; tion[DDCMI_XZ] += m * x * z; tion[DDCMI_YZ] += m * y * z;
         double *two[2] = {rows + 12 * st->group[k]->index, rows + 12 * (sys->ngroup + st->species[k]->index)};
         for (int w = 0; w < 2; w++)
         {
            double *r = two[w];
            r[0] += K; r[1] += m * x * x; r[2] += m * y * y; r[3] += m * z * z; r[4] += m * x * y; r[5] += m * x * z; r[6] += m * y * z;
            r[7] += m; r[8] += 1.0; r[9] += K * x; r[10] += K * y; r[11] += K * z;
         }
      }
      e->rk = rk;
   }
   else
   {
      if (ddcmi_kinetic(ctx, &e->rk, tion) != DDCMI_OK) die("kinetic_terms", ddcmi_last_error(ctx));
      if (sys->ngroup > 0 && ddcmi_kinetic_detail(ctx, 0, sys->ngroup, rows) != DDCMI_OK) die("kinetic_terms", ddcmi_last_error(ctx));
      if (sys->nspecies > 0 && ddcmi_kinetic_detail(ctx, 1, sys->nspecies, rows + 12 * sys->ngroup) != DDCMI_OK) die("kinetic_terms", ddcmi_last_error(ctx));
   }
   e->tion.xx = tion[DDCMI_XX]; e->tion.yy = tion[DDCMI_YY]; e->tion.zz = tion[DDCMI_ZZ];
   e->tion.xy = tion[DDCMI_XY]; e->tion.xz = tion[DDCMI_XZ]; e->tion.yz = tion[DDCMI_YZ];
   e->thermal_flux.x = e->thermal_flux.y = e->thermal_flux.z = 0.0; e->mass = 0.0;
   for (int g = 0; g < sys->ngroup; g++) etype_from_row(&sys->group[g]->energyInfo, rows + 12 * g);
   for (int q = 0; q < sys->nspecies; q++)
   {
      const double *r = rows + 12 * (sys->ngroup + q);
      etype_from_row(&sys->species[q]->energyInfo, r);
      e->thermal_flux.x += r[9]; e->thermal_flux.y += r[10]; e->thermal_flux.z += r[11]; e->mass += r[7];      /* every bead has one species */
   }
   free(rows);
   e->number = (par.world > 1) ? (double)ddcmi_nlocal(ctx) : (double)sys->nlocal;      /* this rank's beads (they migrate); eval_energyInfo sums the ranks */
   e->temperature = 2.0 * e->rk / (3.0 * (double)sys->nglobal);        /* energy.c:151 */
}

/* eval_energyInfo (energyInfo.c:75-148), one rank */
void eval_energyInfo(SYSTEM *sys)
{
   ETYPE *e = &sys->energyInfo;
   if (par.world > 1)
   {
      /* allreduce(energyInfo), energyInfo.c:9-63: the members this path fills */
      double b[19] = {e->rk, e->eion, e->virial.xx, e->virial.yy, e->virial.zz, e->virial.xy, e->virial.xz, e->virial.yz,
                      e->number, e->tion.xx, e->tion.yy, e->tion.zz, e->tion.xy, e->tion.xz, e->tion.yz,
                      e->thermal_flux.x, e->thermal_flux.y, e->thermal_flux.z, e->mass};
      ddcmi_ctx *c = accelerator_getAccelerator(NULL)->parms;
      if (ddcmi_comm_allreduce_sum(c, b, 19) != DDCMI_OK) die("eval_energyInfo", ddcmi_last_error(c));
      e->rk = b[0]; e->eion = b[1]; e->virial.xx = b[2]; e->virial.yy = b[3]; e->virial.zz = b[4]; e->virial.xy = b[5]; e->virial.xz = b[6]; e->virial.yz = b[7];
      e->number = b[8]; e->tion.xx = b[9]; e->tion.yy = b[10]; e->tion.zz = b[11]; e->tion.xy = b[12]; e->tion.xz = b[13]; e->tion.yz = b[14];
      e->thermal_flux.x = b[15]; e->thermal_flux.y = b[16]; e->thermal_flux.z = b[17]; e->mass = b[18];
      /* the group and species copies (energyInfo.c:118-141 sums the groups' blocks the same way) */
      const int ncl = sys->ngroup + sys->nspecies;
      double *rows = (double *)malloc(sizeof(double) * 12 * (size_t)(ncl > 0 ? ncl : 1));
      if (!rows) die("eval_energyInfo", "out of memory");
      for (int q = 0; q < ncl; q++)
      {
         const ETYPE *g = q < sys->ngroup ? &sys->group[q]->energyInfo : &sys->species[q - sys->ngroup]->energyInfo;
         double *r = rows + 12 * q;
         r[0] = g->rk; r[1] = g->tion.xx; r[2] = g->tion.yy; r[3] = g->tion.zz; r[4] = g->tion.xy; r[5] = g->tion.xz; r[6] = g->tion.yz;
         r[7] = g->mass; r[8] = g->number; r[9] = g->thermal_flux.x; r[10] = g->thermal_flux.y; r[11] = g->thermal_flux.z;
      }
      for (int q0 = 0; q0 < ncl; q0 += 5)      /* (the all-reduce takes up to 64 doubles a call) */
         if (ddcmi_comm_allreduce_sum(c, rows + 12 * q0, 12 * (ncl - q0 < 5 ? ncl - q0 : 5)) != DDCMI_OK) die("eval_energyInfo", ddcmi_last_error(c));
      for (int q = 0; q < ncl; q++) etype_from_row(q < sys->ngroup ? &sys->group[q]->energyInfo : &sys->species[q - sys->ngroup]->energyInfo, rows + 12 * q);
      free(rows);
   }
   /* the barostat moves the box on the device */
   if (ddcmi_get_box(accelerator_getAccelerator(NULL)->parms, sys->box->h0) == DDCMI_OK)
      sys->box->volume = sys->box->h0[0] * sys->box->h0[4] * sys->box->h0[8];
   double vol = sys->box->volume;
   const double *v = &e->virial.xx, *t = &e->tion.xx;
   double *s = &e->sion.xx;
   for (int k = 0; k < 6; k++) s[k] = (v[k] + t[k]) * (1.0 / (-vol));   /* SMATACUM, SMATNORM(-vol) :108-110 */
   e->pion = -(e->sion.xx + e->sion.yy + e->sion.zz) / 3.0;             /* :114 */
   e->temperature = 2.0 * e->rk / (3.0 * e->number - sys->nConstraints); /* :115 */
   sys->energy = e->eion + e->rk;                                       /* :116 */
   e->energy = sys->energy;
   /* group branch :118-141: the temperatures BERENDSEN reads */
   ddcmi_ctx *ctx = accelerator_getAccelerator(NULL)->parms;
   double Tg[32];
   if (host_integrated(sys))
   {
      for (int g = 0; g < sys->ngroup; g++)      /* energyInfo.c:139 */
         if (sys->group[g]->energyInfo.number > 0.0) sys->group[g]->energyInfo.temperature = 2.0 * sys->group[g]->energyInfo.rk / (3.0 * sys->group[g]->energyInfo.number);
      return;
   }
   if (ddcmi_group_temperatures(ctx, Tg) != DDCMI_OK) die("eval_energyInfo", ddcmi_last_error(ctx));
   for (int g = 0; g < sys->ngroup && g < 32; g++) sys->group[g]->energyInfo.temperature = Tg[g];
}

/* more than one rank: the beads every rank owns now (they migrate), identified by gid, gathered on rank 0 in rank order;
 * collective.  Species and group come back as the deck's objects (the group of a bead is looked up by its gid). */
typedef struct { uint64_t gid; int species, pad; double r[3], v[3], f[3]; uint64_t lcg_state; uint32_t lcg_multID, lcg_prime; } PREC;
static const ddcmi_setup *gather_setup = NULL;
static gid_order *gather_tab = NULL;
static int gather_state(SYSTEM *sys)
{
   ddcmi_ctx *ctx = accelerator_getAccelerator(NULL)->parms;
   STATE *st = sys->state;
   const ddcmi_setup *s = gather_setup;
   const int cap = (int)sys->nglobal + 16;
   int n = 0;
   uint64_t *gid = malloc(sizeof(uint64_t) * (size_t)cap);
   int *sp = malloc(sizeof(int) * (size_t)cap);
   double *a = malloc(sizeof(double) * 9 * (size_t)cap);
   if (ddcmi_download_particles(ctx, cap, &n, gid, sp, a, a + cap, a + 2 * (size_t)cap, a + 3 * (size_t)cap, a + 4 * (size_t)cap, a + 5 * (size_t)cap,
                                a + 6 * (size_t)cap, a + 7 * (size_t)cap, a + 8 * (size_t)cap) != DDCMI_OK) die("sendHostState", ddcmi_last_error(ctx));
   sys->nlocal = sys->nion = (unsigned)n;
   int *cnt = malloc(sizeof(int) * (size_t)par.world);
   if (ddcmi_rdzv_allgather(par.rdzv, &n, cnt, sizeof(int)) != DDCMI_OK) die("sendHostState", ddcmi_rdzv_last_error(par.rdzv));
   long tot = 0;
   for (int r = 0; r < par.world; r++) tot += cnt[r];
   if (tot != (long)sys->nglobal) die("sendHostState", "the ranks' bead counts do not add up to the global count");
   PREC *rec = malloc(sizeof(PREC) * (size_t)(par.rank == 0 ? tot : n) + 8);
   for (int i = 0; i < n; i++)
   {
      PREC *q = rec + i;
      q->gid = gid[i]; q->species = sp[i]; q->pad = 0;
      for (int k = 0; k < 3; k++) { q->r[k] = a[(size_t)k * cap + i]; q->v[k] = a[(size_t)(3 + k) * cap + i]; q->f[k] = a[(size_t)(6 + k) * cap + i]; }
   }
   if (s->random_lcg64 && n > 0)
   {
      /* the LCG64 records of the beads this rank owns now, in the order of ddcmi_download_particles */
      uint64_t *ls = malloc(sizeof(uint64_t) * (size_t)n); uint32_t *lm = malloc(sizeof(uint32_t) * 2 * (size_t)n);
      if (ddcmi_get_random_lcg64(ctx, n, ls, lm, lm + n) != DDCMI_OK) die("sendHostState", ddcmi_last_error(ctx));
      for (int i = 0; i < n; i++) { rec[i].lcg_state = ls[i]; rec[i].lcg_multID = lm[i]; rec[i].lcg_prime = lm[n + i]; }
      free(ls); free(lm);
   }
   free(gid); free(sp); free(a);
   if (par.rank != 0)
   {
      const int peer = 0; const void *sb = rec; const size_t sbytes = sizeof(PREC) * (size_t)n;
      if (ddcmi_rdzv_exchange(par.rdzv, n > 0 ? 1 : 0, &peer, &sb, &sbytes, 0, NULL, NULL, NULL) != DDCMI_OK) die("sendHostState", ddcmi_rdzv_last_error(par.rdzv));
   }
   else
   {
      int *peer = malloc(sizeof(int) * (size_t)par.world); void **rb = malloc(sizeof(void *) * (size_t)par.world); size_t *rbytes = malloc(sizeof(size_t) * (size_t)par.world);
      int nr = 0; size_t off = (size_t)n;
      for (int r = 1; r < par.world; r++)
      {
         if (cnt[r] > 0) { peer[nr] = r; rb[nr] = rec + off; rbytes[nr] = sizeof(PREC) * (size_t)cnt[r]; nr++; }
         off += (size_t)cnt[r];
      }
      if (ddcmi_rdzv_exchange(par.rdzv, 0, NULL, NULL, NULL, nr, peer, rb, rbytes) != DDCMI_OK) die("sendHostState", ddcmi_rdzv_last_error(par.rdzv));
      free(peer); free(rb); free(rbytes);
      for (long i = 0; i < tot; i++)
      {
         const PREC *q = rec + i;
         gid_order key = {q->gid, 0};
         const gid_order *hit = bsearch(&key, gather_tab, (size_t)s->natoms, sizeof(gid_order), cmp_gid);
         if (!hit) die("sendHostState", "a bead came back with a gid the deck does not have");
         st->label[i] = q->gid; st->species[i] = sys->species[q->species]; st->group[i] = sys->group[s->group[hit->id]]; st->q[i] = st->species[i]->charge;
         st->rx[i] = q->r[0]; st->ry[i] = q->r[1]; st->rz[i] = q->r[2]; st->vx[i] = q->v[0]; st->vy[i] = q->v[1]; st->vz[i] = q->v[2];
         st->fx[i] = q->f[0]; st->fy[i] = q->f[1]; st->fz[i] = q->f[2];
         if (s->random_lcg64) { ((ddcmi_setup *)s)->lcg_state[i] = q->lcg_state; ((ddcmi_setup *)s)->lcg_multID[i] = q->lcg_multID; ((ddcmi_setup *)s)->lcg_prime[i] = q->lcg_prime; }      /* the writer's order */
      }
      st->nlocal = st->nion = (int)tot;
   }
   free(rec); free(cnt);
   return DDCMI_OK;
}
int sendHostState(SYSTEM *sys)
{
   ddcmi_ctx *ctx = accelerator_getAccelerator(NULL)->parms;
   STATE *st = sys->state;
   if (par.world > 1) return gather_state(sys);
   if (host_integrated(sys)) return DDCMI_OK;      /* STATE is the master copy */
   return ddcmi_download_state(ctx, DDCMI_POS | DDCMI_VEL | DDCMI_FORCE, st->rx, st->ry, st->rz, st->vx, st->vy, st->vz, st->fx, st->fy, st->fz);
}

/* ------------------------------------------------------------------------- */
/* restart writer */
static int writeRestart_rank0(SIMULATE *simulate, const char *dir, int restartLink);
int writeRestart(SIMULATE *simulate, const char *dir, int restartLink)
{
   SYSTEM *sys = simulate->system;
   if (sendHostState(sys) != DDCMI_OK) return -1;
   if (par.world > 1)
   {
      /* rank 0 holds the gathered state and writes the files of a one-rank run; the others wait until they exist */
      int rc0 = (par.rank == 0) ? writeRestart_rank0(simulate, dir, restartLink) : 0;
      double flag = (double)rc0;
      if (ddcmi_rdzv_allreduce_f64(par.rdzv, &flag, 1, 0) != DDCMI_OK) return -1;
      return flag != 0.0 ? -1 : 0;
   }
   return writeRestart_rank0(simulate, dir, restartLink);
}
static int writeRestart_rank0(SIMULATE *simulate, const char *dir, int restartLink)
{
   SYSTEM *sys = simulate->system;
   STATE *st = sys->state;
   if (dir) snprintf(simulate->snapshotdir, sizeof(simulate->snapshotdir), "%s", dir);
   else snprintf(simulate->snapshotdir, sizeof(simulate->snapshotdir), "snapshot.%012" PRId64, simulate->loop);    /* loopFormat, io.c:128-129 */
   if (mkdir(simulate->snapshotdir, 0777) != 0 && errno != EEXIST) return -1;
   /* the files appear under their final names only once they are complete: a crash while writing leaves the previous
    * snapshot and the previous ./restart intact */
   char path[1024], tmppath[1100];
   snprintf(path, sizeof(path), "%s/atoms#000000", simulate->snapshotdir);
   snprintf(tmppath, sizeof(tmppath), "%s.tmp", path);
   FILE *f = fopen(tmppath, "w");
   if (!f) return -1;
   const double cLen = units_convert(1.0, NULL, "l"), cVel = cLen / units_convert(1.0, NULL, "t");
   const char *fmt = "%08x %12.12" PRIu64 " %s %s %s %21.13e %21.13e %21.13e %21.13e %21.13e %21.13e";
   /* record length: longest possible record, terminator, padded to 8 bytes (collection_write.c:87-96) */
   size_t maxsp = 1, maxgr = 1;
   for (int i = 0; i < sys->nspecies; i++) if (strlen(sys->species[i]->name) > maxsp) maxsp = strlen(sys->species[i]->name);
   for (int g = 0; g < sys->ngroup; g++) if (strlen(sys->group[g]->name) > maxgr) maxgr = strlen(sys->group[g]->name);
   char line[1024];
   int lrec = snprintf(line, sizeof(line), fmt, 0u, (uint64_t)0, "ATOM", " ", " ", -1.0e-100, -1.0e-100, -1.0e-100, -1.0e-100, -1.0e-100, -1.0e-100);
   lrec += (int)(maxsp - 1) + (int)(maxgr - 1) + 1;
   /* the random field (collection_write.c:81-89,157-161): lcg64_write's "%16.16llx %1u %8.8x" behind the velocities */
   ddcmi_setup *su = simulate->setup;
   const int rnd = su->random_lcg64 && st->nlocal == su->natoms;
   const int randomFieldSize = rnd ? 16 + 1 + 1 + 1 + 8 : 0;
   if (rnd)
   {
      /* one rank: the advanced states in caller = file order, like STATE; several: gather_state left them in STATE's order */
      ddcmi_ctx *ctx = accelerator_getAccelerator(NULL)->parms;
      if (par.world == 1 && ddcmi_get_random_lcg64(ctx, st->nlocal, su->lcg_state, su->lcg_multID, su->lcg_prime) != DDCMI_OK) { fclose(f); return -1; }
      lrec += randomFieldSize + 1;
   }
   lrec = 8 * ((lrec + 7) / 8);
   const double *h = sys->box->h0;
   time_t now = time(NULL);
   char stamp[64];
   strftime(stamp, sizeof(stamp), "%Y-%m-%d-%H:%M:%S", localtime(&now));
   int key;
   memcpy(&key, "1234", 4);
   fprintf(f, "particle FILEHEADER {type=MULTILINE; datatype=FIXRECORDASCII; checksum=CRC32; create_time=%s; run_id=0x%08x;\n", stamp, 0u);
   fprintf(f, "code_version=%s; srcpath=libddcmi;\n", ddcmi_version());
   fprintf(f, "loop=%" PRId64 "; time=%f fs;\n", simulate->loop, units_convert(simulate->time, NULL, "t"));
   fprintf(f, "nfiles=1; nrecord=%d; lrec=%d; nfields=11; endian_key=%d;\n", st->nlocal, lrec, key);
   fprintf(f, "field_names=checksum id class type group rx ry rz vx vy vz;\n");
   fprintf(f, "field_types=u u s s s f f f f f f;\n");
   fprintf(f, "field_units=1 1 1 1 1 Ang Ang Ang Ang/fs Ang/fs Ang/fs;\n");
   fprintf(f, "field_format=%s;\n", fmt);
   fprintf(f, "reducedcorner=%21.14f %21.14f %21.14f;\n", -0.5, -0.5, -0.5);
   fprintf(f, "h=%21.14f %21.14f %21.14f\n  %21.14f %21.14f %21.14f\n  %21.14f %21.14f %21.14f Ang;\n",
           h[0] * cLen, h[1] * cLen, h[2] * cLen, h[3] * cLen, h[4] * cLen, h[5] * cLen, h[6] * cLen, h[7] * cLen, h[8] * cLen);
   fprintf(f, "random = %s;\nrandomFieldSize = %d;\ngroups =", rnd ? su->random_name : "NONE", randomFieldSize);
   for (int g = 0; g < sys->ngroup; g++) fprintf(f, " %s", sys->group[g]->name);
   fprintf(f, ";\nspecies =");
   for (int i = 0; i < sys->nspecies; i++) fprintf(f, " %s", sys->species[i]->name);
   fprintf(f, ";\ntypes = ATOM;\n}\n\n");
   for (int i = 0; i < st->nlocal; i++)
   {
      /* positions come back wrapped into the box (ddcmi_download_state = backInBox) */
      int len = snprintf(line, sizeof(line), fmt, 0u, (uint64_t)st->label[i], "ATOM", st->species[i]->name, st->group[i]->name,
                         st->rx[i] * cLen, st->ry[i] * cLen, st->rz[i] * cLen, st->vx[i] * cVel, st->vy[i] * cVel, st->vz[i] * cVel);
      if (rnd && len > 0 && len < (int)sizeof(line))
         len += snprintf(line + len, sizeof(line) - (size_t)len, " %16.16llx %1u %8.8x", (unsigned long long)su->lcg_state[i], (unsigned)su->lcg_multID[i], (unsigned)su->lcg_prime[i]);
      if (len > lrec - 1) { fclose(f); return -1; }
      for (int l = len; l < lrec; l++) line[l] = ' ';
      line[lrec - 1] = '\n';
      char tmp[16];
      snprintf(tmp, sizeof(tmp), "%08x", ddcmi_crc32((const unsigned char *)line + 8, (size_t)lrec - 8));
      memcpy(line, tmp, 8);
      if (fwrite(line, 1, (size_t)lrec, f) != (size_t)lrec) { fclose(f); return -1; }
   }
   if (fclose(f) != 0) return -1;
   if (rename(tmppath, path) != 0) return -1;
   snprintf(path, sizeof(path), "%s/restart", simulate->snapshotdir);
   snprintf(tmppath, sizeof(tmppath), "%s.tmp", path);
   f = fopen(tmppath, "w");
   if (!f) return -1;
   fprintf(f, "%s SIMULATE { run_id=0x%08x; loop=%" PRId64 "; time=%f fs;}\n", simulate->name, 0u, simulate->loop, units_convert(simulate->time, NULL, "t"));
   /* box_write, box.c:91-108 */
   fprintf(f, "box BOX {\n h  = %21.14e %21.14e %21.14e\n      %21.14e %21.14e %21.14e\n      %21.14e %21.14e %21.14e;\n}\n",
           h[0] * cLen, h[1] * cLen, h[2] * cLen, h[3] * cLen, h[4] * cLen, h[5] * cLen, h[6] * cLen, h[7] * cLen, h[8] * cLen);
   fprintf(f, "collection COLLECTION { size=%" PRIu64 "; files=%s/atoms#;}\n", (uint64_t)sys->nglobal, simulate->snapshotdir);
   if (fclose(f) != 0) return -1;
   if (rename(tmppath, path) != 0) return -1;
   if (restartLink)
   {
      /* a new link under a temporary name, renamed over ./restart: there is a valid restart at every instant */
      char lnk[64];
      snprintf(lnk, sizeof(lnk), "restart.tmp.%d", (int)getpid());
      unlink(lnk);
      if (symlink(path, lnk) != 0) return -1;
      if (rename(lnk, "restart") != 0) { unlink(lnk); return -1; }
   }
   return 0;
}

/* ------------------------------------------------------------------------- */
/* simulate_init (simulate.c:104-297) + system_init (system.c:79-214) from the deck */
SIMULATE *simulate_init(const char *object_file, const char *restart_file, const char *extra, char *err, int errlen)
{
   ddcmi_setup *s = ddcmi_deck_load_with(object_file, restart_file, extra, err, errlen);
   if (!s) return NULL;
   parallel_init(s);
   SIMULATE *sim = calloc(1, sizeof(SIMULATE));
   sim->name = strdup("simulate");
   sim->setup = s;
   sim->loop = s->loop; sim->maxloop = s->maxloop; sim->time = s->time; sim->dt = s->dt; sim->printrate = s->printrate > 0 ? s->printrate : 1;
   sim->snapshotrate = s->snapshotrate; sim->checkpointrate = s->checkpointrate;
   SYSTEM *sys = sim->system = calloc(1, sizeof(SYSTEM));
   sys->name = strdup("system");
   sys->nspecies = s->nspecies;
   sys->species = calloc(s->nspecies + 1, sizeof(SPECIES *));
   for (int i = 0; i < s->nspecies; i++)
   {
      SPECIES *sp = sys->species[i] = calloc(1, sizeof(SPECIES));
      sp->name = strdup(s->species_name[i]); sp->index = i; sp->mass = s->mass[i]; sp->charge = s->charge[i];
   }
   sys->ngroup = s->ngroup;
   sys->group = calloc(s->ngroup + 1, sizeof(GROUP *));
   for (int g = 0; g < s->ngroup; g++)
   {
      GROUP *gp = sys->group[g] = calloc(1, sizeof(GROUP));
      gp->name = strdup(s->group_name[g]); gp->index = g;
      gp->itype = s->group_type[g] == DDCMI_GROUP_FREE ? FREE : s->group_type[g] == DDCMI_GROUP_BERENDSEN ? BERENDSEN : s->group_type[g] == DDCMI_GROUP_LANGEVIN ? LANGEVIN_GROUP : OTHER_GROUP;
      gp->Teq = s->group_Teq[g]; gp->tau = s->group_tau[g]; gp->interval = s->group_interval[g];
   }
   BOX_STRUCT *box = sys->box = calloc(1, sizeof(BOX_STRUCT));
   memcpy(box->h0, s->h, sizeof(double) * 9); box->pbc = s->pbc; box->volume = s->h[0] * s->h[4] * s->h[8];
   int n = s->natoms;
   STATE *st = sys->state = calloc(1, sizeof(STATE));
   st->nlocal = st->nion = n;
   if (par.world > 1)
   {
      /* STATE is the gathered global state on rank 0 (sendHostState): its own arrays, in gather order */
      double **a6[] = {&st->rx, &st->ry, &st->rz, &st->vx, &st->vy, &st->vz};
      const double *src[] = {s->rx, s->ry, s->rz, s->vx, s->vy, s->vz};
      for (int k = 0; k < 6; k++) { *a6[k] = malloc(sizeof(double) * (size_t)(n > 0 ? n : 1)); memcpy(*a6[k], src[k], sizeof(double) * (size_t)n); }
      st->label = malloc(sizeof(gid_type) * (size_t)(n > 0 ? n : 1));
      memcpy(st->label, s->gid, sizeof(gid_type) * (size_t)n);
   }
   else
   {
      st->rx = s->rx; st->ry = s->ry; st->rz = s->rz; st->vx = s->vx; st->vy = s->vy; st->vz = s->vz;    /* aliases of the deck arrays */
      st->label = s->gid;
   }
   st->fx = calloc(n, sizeof(double)); st->fy = calloc(n, sizeof(double)); st->fz = calloc(n, sizeof(double));
   st->q = calloc(n, sizeof(double));
   st->species = calloc(n, sizeof(SPECIES *)); st->group = calloc(n, sizeof(GROUP *));
   for (int i = 0; i < n; i++) { st->species[i] = sys->species[s->species[i]]; st->group[i] = sys->group[s->group[i]]; st->q[i] = st->species[i]->charge; }
   sys->nlocal = sys->nion = n; sys->nglobal = n; sys->loop = s->loop; sys->time = s->time; sys->nConstraints = s->nConstraints; sys->deltaR = s->deltaR;
   DDC *ddc = sim->ddc = calloc(1, sizeof(DDC));
   ddc->updateRate = s->updateRate; ddc->lx = s->lx; ddc->ly = s->ly; ddc->lz = s->lz;
   if (par.world > 1) { ddc->lx = par.grid[0]; ddc->ly = par.grid[1]; ddc->lz = par.grid[2]; } ddc->rcut = s->rmax + s->deltaR;   /* ddcenergy.c:43-55 cutoffs() */
   /* simulate.c:172: accelerator_init when the deck names one; this library always needs one */
   if (!s->has_accelerator && par.rank == 0) printf("no ACCELERATOR object in the deck: running on HIP device 0 (this library has no CPU force path)\n");
   sim->accelerator = accelerator_init(sim, "accelerator", s->has_accelerator ? s->accelerator_type : "HIP");
   sys->npotential = 1;
   sys->potential = calloc(2, sizeof(POTENTIAL *));
   sys->potential[0] = potential_init(sim, "martini", "MARTINI");
   {
      /* cutoffs() (ddcenergy.c:43-55): the largest cutoff any potential reports, plus the skin */
      double rmax = 0.0;
      for (int i = 0; i < sys->npotential; i++)
      {
         int nc = 0;
         RCUT_TYPE *rc = sys->potential[i]->getCutoffs(sys, sys->potential[i]->parms, &nc);
         for (int k = 0; k < nc; k++) if (rc[k].value > rmax) rmax = rc[k].value;
      }
      ddc->rcut = rmax + s->deltaR;
   }
   sim->integrator = integrator_init(sim, "nglf", s->integrator_type);
   /* sendGPUState + sendForceVelocityToGPU (masters.c:389-393) */
   ddcmi_ctx *ctx = sim->accelerator->parms;
   if (par.world > 1)
   {
      gather_setup = s;
      gather_tab = malloc(sizeof(gid_order) * (size_t)(n > 0 ? n : 1));
      for (int i = 0; i < n; i++) { gather_tab[i].gid = s->gid[i]; gather_tab[i].id = i; }
      qsort(gather_tab, (size_t)n, sizeof(gid_order), cmp_gid);
      /* ddcAssignment: every rank read the whole particle set and keeps the beads of its brick */
      if (sim->integrator->uses_gpu == 0) die("simulate_init", "DDCMI_CPU_INTEGRATOR runs on one rank only");
      parallel_comm_init(ctx);
      int m = 0;
      int *sp = malloc(sizeof(int) * (size_t)(n > 0 ? n : 1)), *gr = malloc(sizeof(int) * (size_t)(n > 0 ? n : 1));
      double *a[6];
      for (int k = 0; k < 6; k++) a[k] = malloc(sizeof(double) * (size_t)(n > 0 ? n : 1));
      uint64_t *lab = malloc(sizeof(uint64_t) * (size_t)(n > 0 ? n : 1));
      for (int i = 0; i < n; i++)
         if (brick_of(s, s->rx[i], s->ry[i], s->rz[i]) == par.rank)
         {
            a[0][m] = s->rx[i]; a[1][m] = s->ry[i]; a[2][m] = s->rz[i]; a[3][m] = s->vx[i]; a[4][m] = s->vy[i]; a[5][m] = s->vz[i];
            lab[m] = s->gid[i]; sp[m] = s->species[i]; gr[m] = s->group[i]; m++;
         }
      int rcu = ddcmi_upload_state(ctx, m, a[0], a[1], a[2], a[3], a[4], a[5], lab, sp, gr);
      if (rcu == DDCMI_OK && s->random_lcg64 && m > 0)
      {
         /* the streams of the beads this rank starts with; from here on they migrate with them */
         uint64_t *ls = malloc(sizeof(uint64_t) * (size_t)m); uint32_t *lm = malloc(sizeof(uint32_t) * 2 * (size_t)m);
         int k = 0;
         for (int i = 0; i < n; i++)
            if (brick_of(s, s->rx[i], s->ry[i], s->rz[i]) == par.rank) { ls[k] = s->lcg_state[i]; lm[k] = s->lcg_multID[i]; lm[m + k] = s->lcg_prime[i]; k++; }
         rcu = ddcmi_set_random_lcg64(ctx, m, ls, lm, lm + m);
         free(ls); free(lm);
      }
      for (int k = 0; k < 6; k++) free(a[k]);
      free(lab); free(sp); free(gr);
      if (rcu != DDCMI_OK) die("simulate_init", ddcmi_last_error(ctx));      /* (a rank that returned would leave the others in their collectives) */
      sys->nlocal = sys->nion = (unsigned)m;
      if (s->random_lcg64 && par.rank == 0)
         printf("RANDOM %s: LCG64 streams of %d particles %s; they migrate with their beads\n", s->random_name, n,
                s->lcg_from_file ? "from the atoms file" : "at their default values, as one task assigns them (no random field in the atoms file)");
      if (par.rank == 0) printf("%d ranks on a %d x %d x %d grid of domains (%s transport); rank 0 owns %d of %d beads\n", par.world, par.grid[0], par.grid[1], par.grid[2],
                                par.host_transport ? "host" : "RCCL", m, n);
      return sim;
   }
   if (ddcmi_upload_state(ctx, n, st->rx, st->ry, st->rz, st->vx, st->vy, st->vz, st->label, s->species, s->group) != DDCMI_OK)
   {
      snprintf(err, errlen, "%s", ddcmi_last_error(ctx));
      return NULL;
   }
   if (s->random_lcg64 && n > 0)
   {
      /* RANDOM type LCG64: the particles' streams follow them onto the device (langevin.c:95-96 reads random_getParms(random, k)) */
      if (ddcmi_set_random_lcg64(ctx, n, s->lcg_state, s->lcg_multID, s->lcg_prime) != DDCMI_OK) { snprintf(err, errlen, "%s", ddcmi_last_error(ctx)); return NULL; }
      printf("RANDOM %s: LCG64 streams of %d particles %s\n", s->random_name, n, s->lcg_from_file ? "from the atoms file" : "at their default values (no random field in the atoms file)");
   }
   return sim;
}

/* printinfoA, printinfo.c:125-232: one line of the `data` file */
/* molecularPressure (molecularPressure.c:22-67) + convertToMolecularPressures (printinfo.c:233-240):
 * the atomic virial minus sum_atoms (r_a - R_mol).f_a per molecule (R_mol = centre of mass, nearest
 * images about the molecule's first atom), plus N_mol kB T, over the volume.  Host side, from the
 * state copied back at print steps.  Diagonal only, like the reference. */
static int cmp_label_idx(const void *a, const void *b)
{
   const uint64_t *x = a, *y = b;
   return x[0] < y[0] ? -1 : x[0] > y[0] ? 1 : 0;
}
static void convertToMolecularPressures(SIMULATE *simulate, ETYPE *e)
{
   SYSTEM *sys = simulate->system;
   STATE *st = sys->state;
   if (sendHostState(sys) != DDCMI_OK) die("convertToMolecularPressures", "state download failed");      /* (collective) */
   if (par.rank != 0) return;
   int n = st->nlocal;
   uint64_t *key = malloc(sizeof(uint64_t) * 2 * (size_t)(n > 0 ? n : 1));
   for (int i = 0; i < n; i++) { key[2 * i] = st->label[i]; key[2 * i + 1] = (uint64_t)i; }
   qsort(key, n, 2 * sizeof(uint64_t), cmp_label_idx);
   double L[3] = {sys->box->h0[0], sys->box->h0[4], sys->box->h0[8]};
   double vxx = e->virial.xx, vyy = e->virial.yy, vzz = e->virial.zz;
   int nmol = 0;
   for (int k0 = 0; k0 < n;)
   {
      int k1 = k0;
      while (k1 < n && (key[2 * k1] >> 32) == (key[2 * k0] >> 32)) k1++;
      int i0 = (int)key[2 * k0 + 1];
      double M = 0.0, R[3] = {0, 0, 0};
      for (int k = k0; k < k1; k++)
      {
         int i = (int)key[2 * k + 1];
         double m = st->species[i]->mass, d[3] = {st->rx[i] - st->rx[i0], st->ry[i] - st->ry[i0], st->rz[i] - st->rz[i0]};
         for (int a = 0; a < 3; a++) { if (sys->box->pbc >> a & 1) d[a] -= L[a] * rint(d[a] / L[a]); R[a] += m * d[a]; }
         M += m;
      }
      for (int a = 0; a < 3; a++) R[a] /= M;
      for (int k = k0; k < k1; k++)
      {
         int i = (int)key[2 * k + 1];
         double d[3] = {st->rx[i] - st->rx[i0], st->ry[i] - st->ry[i0], st->rz[i] - st->rz[i0]};
         for (int a = 0; a < 3; a++) { if (sys->box->pbc >> a & 1) d[a] -= L[a] * rint(d[a] / L[a]); d[a] -= R[a]; }
         vxx -= d[0] * st->fx[i]; vyy -= d[1] * st->fy[i]; vzz -= d[2] * st->fz[i];
      }
      nmol++;
      k0 = k1;
   }
   free(key);
   double vol = sys->box->volume, NkT = nmol * e->temperature;      /* kB = 1 */
   double pxx = (vxx + NkT) / vol, pyy = (vyy + NkT) / vol, pzz = (vzz + NkT) / vol;
   e->pion = (pxx + pyy + pzz) / 3.0;
   e->sion.xx = -pxx; e->sion.yy = -pyy; e->sion.zz = -pzz;
   e->sion.xy = -e->virial.xy / vol; e->sion.xz = -e->virial.xz / vol; e->sion.yz = -e->virial.yz / vol;
}

void printinfo(SIMULATE *simulate, ETYPE *e_in, int header)
{
   const ddcmi_setup *s = simulate->setup;
   SYSTEM *sys = simulate->system;
   ETYPE ecopy = *e_in, *e = &ecopy;                                   /* printinfoAll works on a copy (printinfo.c:250-253) */
   if (s->printMolecularPressure) convertToMolecularPressures(simulate, e);
   if (par.rank != 0) return;
   double cE = units_convert(1.0, NULL, s->u_energy), cT = units_convert(1.0, NULL, s->u_temperature), cP = units_convert(1.0, NULL, s->u_pressure);
   double cV = units_convert(1.0, NULL, s->u_volume), ct = units_convert(1.0, NULL, s->u_time), cL = units_convert(1.0, NULL, s->u_length);
   double ng = (double)sys->nglobal;
   double time = ct * simulate->time;
   double ekinetic = cE * (e->rk / ng), etot = cE * ((e->eion + e->rk + e->eBath) / ng), epot = cE * (e->eion / ng);
   double temperature = cT * e->temperature, pressure = cP * e->pion, voln = cV * sys->box->volume / ng;
   FILE *out[2] = {stdout, simulate->datafile};
   for (int k = 0; k < 2; k++)
   {
      FILE *f = out[k];
      if (!f) continue;
      if (header)
      {
         char b[6][64];
         snprintf(b[0], 64, "time(%s)", s->u_time); snprintf(b[1], 64, "Etotal(%s)", s->u_energy); snprintf(b[2], 64, "Ekin(%s)", s->u_energy);
         snprintf(b[3], 64, "Epot(%s)", s->u_energy); snprintf(b[4], 64, "Temp(%s)", s->u_temperature); snprintf(b[5], 64, "Press(%s)", s->u_pressure);
         char v[4][64];
         snprintf(v[0], 64, "Volume(%s)", s->u_volume); snprintf(v[1], 64, "lx(%s)", s->u_length); snprintf(v[2], 64, "ly(%s)", s->u_length); snprintf(v[3], 64, "lz(%s)", s->u_length);
         fprintf(f, "%-12s %16s %18s %18s %18s %18s %18s %18s %15s %15s %15s\n", "#loop", b[0], b[1], b[2], b[3], b[4], b[5], v[0], v[1], v[2], v[3]);
      }
      fprintf(f, "%12" PRId64 " %16.6f %18.12f %18.12f %18.12f %18.8f %18.12f %18.12f %15.8f %15.8f %15.8f\n", simulate->loop, time, etot, ekinetic, epot,
              temperature, pressure, voln, cL * sys->box->h0[0], cL * sys->box->h0[4], cL * sys->box->h0[8]);
      fflush(f);
   }
   /* print_stress (printinfo.c:282-318): the stress tensor sion and the thermal flux -- of the UNconverted energyInfo, as
    * printinfoAll passes it (printinfo.c:255) -- and print_hmat (:320-350) */
   if (simulate->stressfile)
   {
      FILE *f = simulate->stressfile;
      const double cJ = units_convert(1.0, NULL, s->u_energyflux);
      if (header)
      {
         char b[10][64];
         const char *nm[9] = {"Sigma_xx", "Sigma_yy", "Sigma_zz", "Sigma_xy", "Sigma_xz", "Sigma_yz", "Jx", "Jy", "Jz"};
         snprintf(b[0], 64, "time(%s)", s->u_time);
         for (int k = 0; k < 9; k++) snprintf(b[1 + k], 64, "%s(%s)", nm[k], k < 6 ? s->u_pressure : s->u_energyflux);
         fprintf(f, "%-12s %16s %16s %16s %16s %16s %16s %16s %16s %16s %16s\n", "#loop", b[0], b[1], b[2], b[3], b[4], b[5], b[6], b[7], b[8], b[9]);
      }
      const ETYPE *u = e_in;
      fprintf(f, "%12" PRId64 " %16.6f %16.12f %16.12f %16.12f %16.12f %16.12f %16.12f %16.12f %16.12f %16.12f\n", simulate->loop, time,
              cP * u->sion.xx, cP * u->sion.yy, cP * u->sion.zz, cP * u->sion.xy, cP * u->sion.xz, cP * u->sion.yz,
              cJ * u->thermal_flux.x, cJ * u->thermal_flux.y, cJ * u->thermal_flux.z);
      fflush(f);
   }
   if (simulate->hmatfile)
   {
      FILE *f = simulate->hmatfile;
      if (header)
      {
         char b[10][64];
         const char *nm[9] = {"h_xx", "h_xy", "h_xz", "h_yx", "h_yy", "h_yz", "h_zx", "h_zy", "h_zz"};
         snprintf(b[0], 64, "time(%s)", s->u_time);
         for (int k = 0; k < 9; k++) snprintf(b[1 + k], 64, "%s(%s)", nm[k], s->u_length);
         fprintf(f, "%-12s %16s %16s %16s %16s %16s %16s %16s %16s %16s %16s\n", "#loop", b[0], b[1], b[2], b[3], b[4], b[5], b[6], b[7], b[8], b[9]);
      }
      const double *h = sys->box->h0;
      fprintf(f, "%12" PRId64 " %16.6f %16.10f %16.10f %16.10f %16.10f %16.10f %16.10f %16.10f %16.10f %16.10f\n", simulate->loop, time,
              cL * h[0], cL * h[1], cL * h[2], cL * h[3], cL * h[4], cL * h[5], cL * h[6], cL * h[7], cL * h[8]);
      fflush(f);
   }
}

/* simulateMaster, masters.c:369-559: firstEnergyCall, then batches of steps up to
 * the next print step (findEndLoop :263-281), energies after each batch */
int simulateMaster(SIMULATE *simulate, const char *datafile_path)
{
   SYSTEM *sys = simulate->system;
   ddcmi_ctx *ctx = simulate->accelerator->parms;
   simulate->datafile = (datafile_path && par.rank == 0) ? fopen(datafile_path, "a") : NULL;
   simulate->stressfile = simulate->hmatfile = NULL;
   if (datafile_path && par.rank == 0 && (simulate->setup->printStress || simulate->setup->printHmatrix))
   {
      /* next to the data file, under the reference's names (printinfo.c:248-249) */
      char path[1200], dir[1024];
      snprintf(dir, sizeof(dir), "%s", datafile_path);
      char *slash = strrchr(dir, '/');
      if (slash) slash[1] = 0; else dir[0] = 0;
      if (simulate->setup->printStress) { snprintf(path, sizeof(path), "%sstress.data", dir); simulate->stressfile = fopen(path, "a"); }
      if (simulate->setup->printHmatrix) { snprintf(path, sizeof(path), "%shmatrix.data", dir); simulate->hmatfile = fopen(path, "a"); }
   }
   simulate->ddc->update = 3;                                     /* firstEnergyCall :579-620 */
   ddcenergy(simulate->ddc, sys, 1);
   printinfo(simulate, &sys->energyInfo, 1);
   while (simulate->loop < simulate->maxloop)
   {
      int64_t endLoop = (simulate->loop / simulate->printrate + 1) * simulate->printrate;
      if (simulate->checkpointrate > 0)
      {
         int64_t nextCk = (simulate->loop / simulate->checkpointrate + 1) * simulate->checkpointrate;      /* masters.c:275-276 */
         if (nextCk < endLoop) endLoop = nextCk;
      }
      if (simulate->snapshotrate > 0)
      {
         int64_t nextSn = (simulate->loop / simulate->snapshotrate + 1) * simulate->snapshotrate;          /* masters.c:275 */
         if (nextSn < endLoop) endLoop = nextSn;
      }
      if (endLoop > simulate->maxloop) endLoop = simulate->maxloop;
      if (simulate->integrator->eval_integrator == (void (*)(void *, void *, void *))nglfHIP)
      {
         /* the device integrator takes the whole batch in one call: inside it the BACK kick of a step and the FRONT
          * kick + drift of the next run as one pass (the path bench.py measures) */
         if (ddcmi_step_nglf(ctx, simulate->dt, (int)(endLoop - simulate->loop)) != DDCMI_OK) die("nglfHIP", ddcmi_last_error(ctx));
         simulate->ddc->update = 0;
         for (int64_t k = simulate->loop; k < endLoop; k++) simulate->time += simulate->dt;      /* the sum nglf forms step by step */
         simulate->loop = endLoop;
         sys->loop = simulate->loop; sys->time = simulate->time;
      }
      else
         while (simulate->loop < endLoop)
            simulate->integrator->eval_integrator(simulate->ddc, simulate, simulate->integrator->parms);
      /* uses_gpu: sendForceEnergyToHost (masters.c:448-453), then kinetic_terms + eval_energyInfo (:454-455) */
      double en[DDCMI_NE], vir[6], rk, tion[6];
      if (ddcmi_get_energies(ctx, en, vir, &rk, tion) != DDCMI_OK) die("simulateMaster", ddcmi_last_error(ctx));
      ETYPE *e = &sys->energyInfo;
      e->eion = en[DDCMI_E_TOTAL];
      e->virial.xx = vir[DDCMI_XX]; e->virial.yy = vir[DDCMI_YY]; e->virial.zz = vir[DDCMI_ZZ];
      e->virial.xy = vir[DDCMI_XY]; e->virial.xz = vir[DDCMI_XZ]; e->virial.yz = vir[DDCMI_YZ];
      kinetic_terms(sys, 1);
      eval_energyInfo(sys);
      if (!isfinite(e->eion))                                     /* masters.c:470-475 */
      {
         if (par.rank == 0) printf("eion = %e is bad. Simulation is being killed at loop = %" PRId64 "\n", e->eion, simulate->loop);
         break;
      }
      if (simulate->loop % simulate->printrate == 0) printinfo(simulate, e, 0);
      if (simulate->checkpointrate > 0 && simulate->loop % simulate->checkpointrate == 0)      /* masters.c:318-322 */
      { if (writeRestart(simulate, NULL, 1) != 0) die("simulateMaster", "writeRestart failed"); }
      else if (simulate->snapshotrate > 0 && simulate->loop % simulate->snapshotrate == 0)     /* doSnapshot, masters.c:340-352: the particle files, no ./restart */
      { if (writeRestart(simulate, NULL, 0) != 0) die("simulateMaster", "snapshot write failed"); }
   }
   sendHostState(sys);
   if (simulate->datafile) fclose(simulate->datafile);
   simulate->datafile = NULL;
   if (simulate->stressfile) fclose(simulate->stressfile);
   if (simulate->hmatfile) fclose(simulate->hmatfile);
   simulate->stressfile = simulate->hmatfile = NULL;
   return 0;
}

void simulate_free(SIMULATE *sim)
{
   if (!sim) return;
   if (sim->accelerator) { ddcmi_destroy(sim->accelerator->parms); free(sim->accelerator->name); free(sim->accelerator->type); free(sim->accelerator); the_accelerator = NULL; }
   SYSTEM *sys = sim->system;
   if (sys)
   {
      STATE *st = sys->state;
      if (st && par.world > 1) { free(st->rx); free(st->ry); free(st->rz); free(st->vx); free(st->vy); free(st->vz); free(st->label); }      /* own arrays, not the deck's */
      if (st) { free(st->fx); free(st->fy); free(st->fz); free(st->q); free(st->species); free(st->group); free(st); }
      for (int i = 0; i < sys->nspecies; i++) { free(sys->species[i]->name); free(sys->species[i]); }
      for (int g = 0; g < sys->ngroup; g++) { free(sys->group[g]->name); free(sys->group[g]); }
      if (sys->potential) { free(sys->potential[0]->parms); free(sys->potential[0]->name); free(sys->potential[0]->type); free(sys->potential[0]); free(sys->potential); }
      free(sys->species); free(sys->group); free(sys->box); free(sys->name); free(sys);
   }
   if (sim->integrator) { free(sim->integrator->name); free(sim->integrator->type); free(sim->integrator); }
   free(sim->ddc);
   if (par.rdzv) { ddcmi_rdzv_barrier(par.rdzv); ddcmi_rdzv_destroy(par.rdzv); par.rdzv = NULL; }      /* (the context that used it is gone) */
   free(gather_tab); gather_tab = NULL; gather_setup = NULL;
   par.world = 1; par.rank = 0;
   ddcmi_setup_free(sim->setup);
   free(sim->name);
   free(sim);
}
