/*
 * ddc_oracle.c -- CPU restatement of ddcMD's Martini MD inner loop.
 * TEST INFRASTRUCTURE ONLY (see ddc_oracle.h).  PARITY UNPINNED: the reference
 * cannot be built here and ships no golden outputs; this restatement is
 * cross-validated by orc_brute_force, finite differences and conservation
 * tests under tests/.
 *
 * Every function cites the /root/reference/src lines it follows.
 */
#include "ddc_oracle.h"
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <stdio.h>
#include <assert.h>
#include <limits.h>

/* bioGid.h:13-22 */
static const uint64_t molMask = (4294967295ull << 32);
static const uint64_t atmgrpMask = 65535ull;
static const uint64_t molResMask = 0xffffffffffff0000ull;

/* bioCharmm.c:32-35 */
#define FLOAT_EPS 1e-08
#define NEAR_ZERO_ANGLE 0.017453292519943295
#define NEAR_180_ANGLE 3.12413936106985

/* bioCharmmParms.h:25-28 exclude masks */
enum { bondMask = 1, angleMask = 2, cosangleMask = 4, ureybradleyMask = 8, torsionMask = 16,
       improperMask = 32, cMapMask = 64, nonBondMask = 128, rebangleMask = 256 };

/* ------------------------------------------------------------------ */
/* preduce.c:147-160 PreduceOrthorhombicB7_OneLatticeReduction, selected as
 * nearestImage_fast/backInBox_fast for an orthorhombic box (PsetMethod,
 * preduce.c:449-480); pbc<7 variants reduce only the periodic axes
 * (preduce.c:176-260). */
static inline void nearestImage_fast(const orc_params *p, double *x, double *y, double *z)
{
   if (p->pbc & 1) { if (*x > 0.5 * p->hxx) *x += -p->hxx; if (*x < -0.5 * p->hxx) *x += p->hxx; }
   if (p->pbc & 2) { if (*y > 0.5 * p->hyy) *y += -p->hyy; if (*y < -0.5 * p->hyy) *y += p->hyy; }
   if (p->pbc & 4) { if (*z > 0.5 * p->hzz) *z += -p->hzz; if (*z < -0.5 * p->hzz) *z += p->hzz; }
}
/* preduce.c:282-338 Preduce (rint based), = nearestImage for an orthorhombic box */
static inline void nearestImage(const orc_params *p, double *x, double *y, double *z)
{
   if (p->pbc & 1) { double da = -rint((1.0 / p->hxx) * (*x)); *x += p->hxx * da; }
   if (p->pbc & 2) { double db = -rint((1.0 / p->hyy) * (*y)); *y += p->hyy * db; }
   if (p->pbc & 4) { double dc = -rint((1.0 / p->hzz) * (*z)); *z += p->hzz * dc; }
}
static double minspan(const orc_params *p)
{
   double m = p->hxx;
   if (p->hyy < m) m = p->hyy;
   if (p->hzz < m) m = p->hzz;
   return m;
}
void orc_back_in_box(const orc_params *p, int n, double *rx, double *ry, double *rz)
{
   /* nglf.c:90 backInBox_fast on every local particle */
   for (int k = 0; k < n; k++) nearestImage_fast(p, rx + k, ry + k, rz + k);
}

/* ------------------------------------------------------------------ */
struct orc_nbr
{
   int n;
   int *start[2];  /* CSR row starts [n+1] : 0 kept list (ifirst[0]), 1 pruned list (ifirst[1]) */
   int *j[2];
   long npairs[2];
   /* neighborRef (neighbor.c:209-246): reference positions and their centroid for neighborCheck */
   double *r0[3];
   double rbar[3];
};

void orc_nbr_free(orc_nbr *nb)
{
   if (!nb) return;
   for (int l = 0; l < 2; l++) { free(nb->start[l]); free(nb->j[l]); }
   for (int a = 0; a < 3; a++) free(nb->r0[a]);
   free(nb);
}
long orc_nbr_npairs(const orc_nbr *nb, int which) { return nb->npairs[which]; }
void orc_nbr_csr(const orc_nbr *nb, int which, const int **start, const int **j)
{
   *start = nb->start[which];
   *j = nb->j[which];
}

/* reOrgPairs test (bioMartini.c:1443-1463): is the pair (i,j) pruned from the
 * LJ list?  Same molecule AND (molecule type has one species OR
 * (atmI,atmJ) in the ownership residue's bpairList). */
static int pair_is_pruned(const orc_params *p, const uint64_t *gid, const int *species, int i, int j)
{
   if (p->nmoltype == 0) return 0;  /* sys->moleculeClass == NULL: bioMartini.c:1374 */
   if ((gid[i] & molMask) != (gid[j] & molMask)) return 0;
   int mt = p->moltype[species[i]];
   if (p->mol_nspecies[mt] > 1)
   {
      unsigned atmI = (unsigned)(gid[i] & atmgrpMask);
      unsigned atmJ = (unsigned)(gid[j] & atmgrpMask);
      for (int k = p->bpair_off[mt]; k < p->bpair_off[mt + 1]; k++)
      {
         unsigned eI = (unsigned)p->bpairI[k], eJ = (unsigned)p->bpairJ[k];
         if ((atmI == eI && atmJ == eJ) || (atmJ == eI && atmI == eJ)) return 1;
      }
      return 0;
   }
   return 1;
}

/* pairlist1 (pairlist.c:205-314): for every i, every j in the 27 neighbouring
 * cells with gid_i < gid_j; min-image if r2 > R2cut; keep if r2 < (rcut+deltaR)^2.
 * The cell grid stands in for GeomBox (geom.c:311); the resulting pair SET is
 * the same, the in-row order differs (the reference prepends to a linked list). */
static long orc_builds = 0;
long orc_nbr_build_count(void) { return orc_builds; }
/* census of the dihedral code's rarely taken branches (tests: same slots as ddcmi_debug_branch_census) */
static long orc_census[8];
void orc_branch_census(long out[8], int reset) { for (int k = 0; k < 8; k++) { out[k] = orc_census[k]; if (reset) orc_census[k] = 0; } }
orc_nbr *orc_nbr_build(const orc_params *p, int n, const double *rx, const double *ry, const double *rz,
                       const uint64_t *gid, const int *species)
{
   orc_builds++;
   double rlist = p->rmax + p->deltaR;
   double rmax_plus_delta2 = rlist * rlist;
   double ms = minspan(p);
   double R2cut = 0.25 * ms * ms;
   double L[3] = {p->hxx, p->hyy, p->hzz};
   int nc[3];
   for (int a = 0; a < 3; a++)
   {
      nc[a] = (int)floor(L[a] / rlist);
      if (nc[a] < 1) nc[a] = 1;
      if (!((p->pbc >> a) & 1)) { if (nc[a] < 1) nc[a] = 1; }
   }
   long ncell = (long)nc[0] * nc[1] * nc[2];
   int *head = malloc(sizeof(int) * ncell);
   int *next = malloc(sizeof(int) * (n > 0 ? n : 1));
   int *cellOf = malloc(sizeof(int) * (n > 0 ? n : 1));
   for (long c = 0; c < ncell; c++) head[c] = -1;
   for (int i = n - 1; i >= 0; i--)
   {
      double r[3] = {rx[i], ry[i], rz[i]};
      int ic[3];
      for (int a = 0; a < 3; a++)
      {
         double s = r[a] / L[a] + 0.5;       /* reduced coordinate, box centred on origin */
         if ((p->pbc >> a) & 1) s -= floor(s);
         else { if (s < 0.0) s = 0.0; if (s >= 1.0) s = 0.999999999999; }   /* open axis: beads outside sit in the edge cell;
                                                                              * only exact while they stay within one cell width of the box */
         ic[a] = (int)(s * nc[a]);
         if (ic[a] >= nc[a]) ic[a] = nc[a] - 1;
         if (ic[a] < 0) ic[a] = 0;
      }
      int c = (ic[2] * nc[1] + ic[1]) * nc[0] + ic[0];
      cellOf[i] = c;
      next[i] = head[c];
      head[c] = i;
   }
   orc_nbr *nb = calloc(1, sizeof(orc_nbr));
   nb->n = n;
   size_t cap[2] = {(size_t)n * 16 + 64, 64};
   for (int l = 0; l < 2; l++)
   {
      nb->start[l] = malloc(sizeof(int) * (n + 1));
      nb->j[l] = malloc(sizeof(int) * cap[l]);
      nb->start[l][0] = 0;
   }
   long np[2] = {0, 0};
   for (int i = 0; i < n; i++)
   {
      int c = cellOf[i];
      int ic[3] = {c % nc[0], (c / nc[0]) % nc[1], c / (nc[0] * nc[1])};
      int visited[27], nvis = 0;
      uint64_t gi_i = gid[i];
      for (int dz = -1; dz <= 1; dz++)
         for (int dy = -1; dy <= 1; dy++)
            for (int dx = -1; dx <= 1; dx++)
            {
               int jc[3] = {ic[0] + dx, ic[1] + dy, ic[2] + dz};
               int ok = 1;
               for (int a = 0; a < 3; a++)
               {
                  if (jc[a] < 0 || jc[a] >= nc[a])
                  {
                     if ((p->pbc >> a) & 1) jc[a] = (jc[a] + nc[a]) % nc[a];
                     else ok = 0;
                  }
               }
               if (!ok) continue;
               int cj = (jc[2] * nc[1] + jc[1]) * nc[0] + jc[0];
               int dup = 0;
               for (int v = 0; v < nvis; v++) if (visited[v] == cj) dup = 1;
               if (dup) continue;
               visited[nvis++] = cj;
               for (int j = head[cj]; j != -1; j = next[j])
               {
                  if (gi_i < gid[j])                                   /* pairlist.c:279 */
                  {
                     double x = rx[i] - rx[j], y = ry[i] - ry[j], z = rz[i] - rz[j];
                     double r2 = x * x + y * y + z * z;
                     if (r2 > R2cut) { nearestImage_fast(p, &x, &y, &z); r2 = x * x + y * y + z * z; }
                     if (r2 < rmax_plus_delta2)
                     {
                        int l = pair_is_pruned(p, gid, species, i, j);  /* reOrgPairs */
                        if ((size_t)np[l] + 1 > cap[l])
                        {
                           cap[l] *= 2;
                           nb->j[l] = realloc(nb->j[l], sizeof(int) * cap[l]);
                        }
                        nb->j[l][np[l]++] = j;
                     }
                  }
               }
            }
      nb->start[0][i + 1] = (int)np[0];
      nb->start[1][i + 1] = (int)np[1];
   }
   nb->npairs[0] = np[0];
   nb->npairs[1] = np[1];
   /* neighborRef: r0 = nearest image of r about the box centre (0 here), rbar = centroid of the locals */
   for (int a = 0; a < 3; a++) { nb->r0[a] = malloc(sizeof(double) * (n > 0 ? n : 1)); nb->rbar[a] = 0.0; }
   for (int i = 0; i < n; i++)
   {
      double x = rx[i], y = ry[i], z = rz[i];
      nearestImage(p, &x, &y, &z);
      nb->r0[0][i] = x; nb->r0[1][i] = y; nb->r0[2][i] = z;
      nb->rbar[0] += x; nb->rbar[1] += y; nb->rbar[2] += z;
   }
   for (int a = 0; a < 3; a++) nb->rbar[a] /= (n > 0 ? n : 1);
   free(head); free(next); free(cellOf);
   return nb;
}

/* ------------------------------------------------------------------ */
/* martiniNonBond, bioMartini.c:989-1122 */
void orc_nonbond(const orc_params *p, const orc_nbr *nb, int n,
                 const double *rx, const double *ry, const double *rz, const int *species,
                 double *fx, double *fy, double *fz, double *pvLJ, double *pvEle, double *virial)
{
   int nspecies = p->nlj;                       /* :1000 parms->nspecies = mmff->nAtomType */
   double r2cut = p->rmax * p->rmax;            /* :1002 */
   double krf = p->krf, crf = p->crf;
   double ms = minspan(p);
   double R2cut = 0.25 * ms * ms;               /* :1017 */
   double vxx = 0, vyy = 0, vzz = 0, vxy = 0, vxz = 0, vyz = 0;
   double q2 = 0.0;
   for (int i = 0; i < n; i++) { double qi = p->charge[species[i]]; q2 += qi * qi; }  /* :1031 */
   double keR = p->keR;                         /* :1033 */
   double vLJ = 0.0;
   double vEle = -0.5 * q2 * keR * crf;         /* :1035 self term */
   const int *start = nb->start[0], *jl = nb->j[0];
   for (int i = 0; i < n; i++)
   {
      int si = p->ljtype[species[i]];
      double kqi = keR * p->charge[species[i]];
      double xi = rx[i], yi = ry[i], zi = rz[i];
      double fxi = 0.0, fyi = 0.0, fzi = 0.0;
      for (int k = start[i]; k < start[i + 1]; k++)
      {
         int j = jl[k];
         int sj = p->ljtype[species[j]];
         int sij = sj + nspecies * si;
         double x = xi - rx[j], y = yi - ry[j], z = zi - rz[j];
         double r2 = x * x + y * y + z * z;
         if (r2 > R2cut) { nearestImage_fast(p, &x, &y, &z); r2 = x * x + y * y + z * z; }
         if (r2 < r2cut)
         {
            double sigma = p->sigma[sij], eps = p->eps[sij];
            double ir = sqrt(1.0 / r2);
            double ir2 = ir * ir;
            double sigma_r = sigma * ir;
            double s2 = sigma_r * sigma_r;
            double s4 = s2 * s2;
            double s6 = s4 * s2;
            double s12 = s6 * s6;
            vLJ += 4.0 * eps * (s12 - s6) + p->shift[sij];
            double dvdr = 24.0 * eps * (s6 - 2.0 * s12) * ir2;
            double kqij = kqi * p->charge[species[j]];
            vEle += kqij * (ir + krf * r2 - crf);
            dvdr += kqij * (2 * krf - ir2 * ir);
            double fxij = -dvdr * x, fyij = -dvdr * y, fzij = -dvdr * z;
            fxi += fxij; fyi += fyij; fzi += fzij;
            fx[j] -= fxij; fy[j] -= fyij; fz[j] -= fzij;
            vxx += fxij * x; vyy += fyij * y; vzz += fzij * z;
            vxy += fxij * y; vxz += fxij * z; vyz += fyij * z;
         }
      }
      fx[i] += fxi; fy[i] += fyi; fz[i] += fzi;
   }
   *pvLJ += vLJ;
   *pvEle += vEle;
   virial[0] += vxx; virial[1] += vyy; virial[2] += vzz;
   virial[3] += vxy; virial[4] += vxz; virial[5] += vyz;
}

/* martiniIntraMoleReaction, bioMartini.c:1124-1208: RF correction on pruned pairs */
void orc_intramol(const orc_params *p, const orc_nbr *nb, int n,
                  const double *rx, const double *ry, const double *rz, const int *species,
                  double *fx, double *fy, double *fz, double *pvEle, double *virial)
{
   double r2cut = p->rmax * p->rmax;
   double krf = p->krf, crf = p->crf;
   double ms = minspan(p);
   double R2cut = 0.25 * ms * ms;
   double keR = p->keR;
   double vEle = 0.0;
   double vxx = 0, vyy = 0, vzz = 0, vxy = 0, vxz = 0, vyz = 0;
   const int *start = nb->start[1], *jl = nb->j[1];
   for (int i = 0; i < n; i++)
   {
      double kqi = keR * p->charge[species[i]];
      double xi = rx[i], yi = ry[i], zi = rz[i];
      double fxi = 0.0, fyi = 0.0, fzi = 0.0;
      for (int k = start[i]; k < start[i + 1]; k++)
      {
         int j = jl[k];
         double x = xi - rx[j], y = yi - ry[j], z = zi - rz[j];
         double r2 = x * x + y * y + z * z;
         if (r2 > R2cut) { nearestImage_fast(p, &x, &y, &z); r2 = x * x + y * y + z * z; }
         if (r2 < r2cut)
         {
            double kqij = kqi * p->charge[species[j]];
            vEle += kqij * (krf * r2 - crf);
            double dvdr = kqij * (2 * krf);
            double fxij = -dvdr * x, fyij = -dvdr * y, fzij = -dvdr * z;
            fxi += fxij; fyi += fyij; fzi += fzij;
            fx[j] -= fxij; fy[j] -= fyij; fz[j] -= fzij;
            vxx += fxij * x; vyy += fyij * y; vzz += fzij * z;
            vxy += fxij * y; vxz += fxij * z; vyz += fyij * z;
         }
      }
      fx[i] += fxi; fy[i] += fyi; fz[i] += fzi;
   }
   *pvEle += vEle;
   virial[0] += vxx; virial[1] += vyy; virial[2] += vzz;
   virial[3] += vxy; virial[4] += vxz; virial[5] += vyz;
}

/* ------------------------------------------------------------------ */
/* Bonded terms.  charmmConvalent (bioCharmmCovalent.c:95-251) sorts the local
 * atoms by gid, cuts residue runs at changes of (gid & molResMask)
 * (charmmResidues :48-93), copies them into a gid-ordered scratch state and
 * calls connectiveEnergy (bioCharmmCovalentEnergies.c:754-795) per residue.
 * Term atom indices are offsets inside the residue (...Sorted.c:34-35). */
typedef struct { uint64_t gid; int id; } gid_order;
static int cmp_gid(const void *a, const void *b)
{
   uint64_t ga = ((const gid_order *)a)->gid, gb = ((const gid_order *)b)->gid;
   return (ga > gb) - (ga < gb);
}
typedef struct { double x, y, z; } vec3;
/* bioVec, bioCharmmCovalentEnergies.c:34-48: r1-r2 reduced with nearestImage */
static inline vec3 bioVec(const orc_params *p, const double *rx, const double *ry, const double *rz, int a1, int a2)
{
   vec3 v = {rx[a1] - rx[a2], ry[a1] - ry[a2], rz[a1] - rz[a2]};
   nearestImage(p, &v.x, &v.y, &v.z);
   return v;
}
static inline double bioNorm(vec3 v) { return sqrt(v.x * v.x + v.y * v.y + v.z * v.z); }
#define DOT3(a, b) ((a).x * (b).x + (a).y * (b).y + (a).z * (b).z)
static inline vec3 cross3(vec3 a, vec3 b)
{
   vec3 c = {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
   return c;
}

/* bioDihedralFast, bioCharmmCovalentEnergies.c:266-351 (the fopen("ddd.data")
 * side effect at :344-345 is deliberately not reproduced) */
static void bioDihedralFast(const orc_params *p, const double *rx, const double *ry, const double *rz,
                            int I, int J, int K, int L, double *angleX, double *sinXptr,
                            vec3 *dI, vec3 *dJ, vec3 *dK, vec3 *dL, double *vir)
{
   double eps = 1e-12;
   vec3 v_ij = bioVec(p, rx, ry, rz, I, J);
   vec3 v_jk = bioVec(p, rx, ry, rz, J, K);
   vec3 v_kl = bioVec(p, rx, ry, rz, K, L);
   double a2 = DOT3(v_ij, v_ij), b2 = DOT3(v_jk, v_jk), c2 = DOT3(v_kl, v_kl);
   double ab = DOT3(v_ij, v_jk), bc = DOT3(v_jk, v_kl), ac = DOT3(v_ij, v_kl);
   double f = ab * bc - ac * b2;
   double g1 = a2 * b2 - ab * ab + eps;
   double g2 = b2 * c2 - bc * bc + eps;
   double y = 1.0 / sqrt(g1 * g2);
   double x = y * f;
   double xab = y * bc + x / g1 * ab;
   double xbc = y * ab + x / g2 * bc;
   double xac = -y * b2;
   double xaa = -0.5 * x * b2 / g1;
   double xcc = -0.5 * x * b2 / g2;
   double xbb = -y * ac - 0.5 * x * (a2 / g1 + c2 / g2);
   vec3 ca, cb, cc;
   ca.x = xab * v_jk.x + xac * v_kl.x + (2 * xaa) * v_ij.x;
   ca.y = xab * v_jk.y + xac * v_kl.y + (2 * xaa) * v_ij.y;
   ca.z = xab * v_jk.z + xac * v_kl.z + (2 * xaa) * v_ij.z;
   cb.x = xab * v_ij.x + xbc * v_kl.x + (2 * xbb) * v_jk.x;
   cb.y = xab * v_ij.y + xbc * v_kl.y + (2 * xbb) * v_jk.y;
   cb.z = xab * v_ij.z + xbc * v_kl.z + (2 * xbb) * v_jk.z;
   cc.x = xbc * v_jk.x + xac * v_ij.x + (2 * xcc) * v_kl.x;
   cc.y = xbc * v_jk.y + xac * v_ij.y + (2 * xcc) * v_kl.y;
   cc.z = xbc * v_jk.z + xac * v_ij.z + (2 * xcc) * v_kl.z;
   vec3 m = cross3(v_ij, v_jk), nn = cross3(v_jk, v_kl), mXn = cross3(m, nn);
   double signnum = DOT3(v_jk, mXn);
   double sign = (signnum < 0.0) ? -1.0 : 1.0;
   x = fmax(fmin(x, 1.0), -1.0);
   *angleX = sign * acos(x);
   *sinXptr = sin(*angleX);
   *dI = ca;
   dJ->x = cb.x - ca.x; dJ->y = cb.y - ca.y; dJ->z = cb.z - ca.z;
   dK->x = cc.x - cb.x; dK->y = cc.y - cb.y; dK->z = cc.z - cb.z;
   dL->x = -cc.x; dL->y = -cc.y; dL->z = -cc.z;
   vir[0] = -1 * (ca.x * v_ij.x + cb.x * v_jk.x + cc.x * v_kl.x);   /* xx */
   vir[3] = -1 * (ca.x * v_ij.y + cb.x * v_jk.y + cc.x * v_kl.y);   /* xy */
   vir[4] = -1 * (ca.x * v_ij.z + cb.x * v_jk.z + cc.x * v_kl.z);   /* xz */
   vir[1] = -1 * (ca.y * v_ij.y + cb.y * v_jk.y + cc.y * v_kl.y);   /* yy */
   vir[5] = -1 * (ca.y * v_ij.z + cb.y * v_jk.z + cc.y * v_kl.z);   /* yz */
   vir[2] = -1 * (ca.z * v_ij.z + cb.z * v_jk.z + cc.z * v_kl.z);   /* zz */
}

void orc_bonded(const orc_params *p, int n,
                const double *rx, const double *ry, const double *rz,
                const uint64_t *gid, const int *species,
                double *fx, double *fy, double *fz, double *e4, double *virial)
{
   if (n == 0 || p->nresi == 0) return;
   int ex = p->excludePotentialTerm;
   /* charmmResidues: qsort by gid */
   gid_order *ord = malloc(sizeof(gid_order) * n);
   for (int i = 0; i < n; i++) { ord[i].id = i; ord[i].gid = gid[i]; }
   qsort(ord, n, sizeof(gid_order), cmp_gid);
   double ebond = 0, eangle = 0, etors = 0, eimpr = 0;
   int first = 0;
   while (first < n)
   {
      uint64_t rkey = ord[first].gid & molResMask;
      int last = first;
      while (last < n && (ord[last].gid & molResMask) == rkey) last++;
      int rt = p->resitype[species[ord[first].id]];
      int na = p->resi_natoms[rt];
      /* padded residues with holes (bioCharmmCovalent.c:147-185) only occur when a
       * residue is split across ranks; the single-rank oracle requires whole ones */
      if (last - first != na)
      {
         fprintf(stderr, "orc_bonded: incomplete residue (have %d atoms, residue type %d needs %d)\n", last - first, rt, na);
         abort();
      }
#define AT(off) (ord[first + (off)].id)
      /* resBondSorted, ...Sorted.c:18-116 */
      if (!(ex & bondMask))
         for (int b = p->bond_off[rt]; b < p->bond_off[rt + 1]; b++)
         {
            int I = AT(p->bondI[b]), J = AT(p->bondJ[b]);
            vec3 bVec = bioVec(p, rx, ry, rz, I, J);
            double bl = bioNorm(bVec);
            double bDelta = bl - p->bond_b0[b];
            ebond += p->bond_kb[b] * bDelta * bDelta;
            vec3 u = {bVec.x / bl, bVec.y / bl, bVec.z / bl};
            double kforce = -2 * p->bond_kb[b] * bDelta;
            double fxD = kforce * u.x, fyD = kforce * u.y, fzD = kforce * u.z;
            fx[I] += fxD; fy[I] += fyD; fz[I] += fzD;
            fx[J] -= fxD; fy[J] -= fyD; fz[J] -= fzD;
            virial[0] += fxD * bVec.x; virial[3] += fxD * bVec.y; virial[4] += fxD * bVec.z;
            virial[1] += fyD * bVec.y; virial[5] += fyD * bVec.z; virial[2] += fzD * bVec.z;
         }
      /* resAngleSorted :118-242 (func 1), resAngleCosineSorted :244-363 (func 2),
       * resAngleRestrainSorted :365-487 (func 10) */
      for (int a = p->angle_off[rt]; a < p->angle_off[rt + 1]; a++)
      {
         int func = p->angle_func[a];
         if (func == 1 && (ex & angleMask)) continue;
         if (func == 2 && (ex & cosangleMask)) continue;
         if (func == 10 && (ex & rebangleMask)) continue;
         int I = AT(p->angleI[a]), J = AT(p->angleJ[a]), K = AT(p->angleK[a]);
         vec3 vij = bioVec(p, rx, ry, rz, I, J);
         double b_ij = bioNorm(vij);
         vec3 uij = {vij.x / b_ij, vij.y / b_ij, vij.z / b_ij};
         vec3 vkj = bioVec(p, rx, ry, rz, K, J);
         double b_kj = bioNorm(vkj);
         vec3 ukj = {vkj.x / b_kj, vkj.y / b_kj, vkj.z / b_kj};
         double cosT = uij.x * ukj.x + uij.y * ukj.y + uij.z * ukj.z;
         double kt = p->angle_k[a], t0 = p->angle_t0[a];
         double coef_i, coef_k;
         if (func == 1)
         {
            double ang = acos(cosT);
            double aDelta = ang - t0;
            eangle += kt * aDelta * aDelta;
            double sinabs = sin(ang);
            coef_i = 2 * kt * aDelta / (b_ij * sinabs);
            coef_k = 2 * kt * aDelta / (b_kj * sinabs);
         }
         else if (func == 2)
         {
            double aDelta = cosT - t0;
            eangle += kt * aDelta * aDelta;
            coef_i = -2 * kt * aDelta / b_ij;
            coef_k = -2 * kt * aDelta / b_kj;
         }
         else
         {
            double sinAsq = 1 - cosT * cosT;
            double aDelta = cosT - t0;
            eangle += kt * aDelta * aDelta / sinAsq;
            double coef_reb = -2 * kt * aDelta * (1 - cosT * t0) / (sinAsq * sinAsq);
            coef_i = coef_reb / b_ij;
            coef_k = coef_reb / b_kj;
         }
         double fxI = coef_i * (ukj.x - uij.x * cosT), fyI = coef_i * (ukj.y - uij.y * cosT), fzI = coef_i * (ukj.z - uij.z * cosT);
         double fxK = coef_k * (uij.x - ukj.x * cosT), fyK = coef_k * (uij.y - ukj.y * cosT), fzK = coef_k * (uij.z - ukj.z * cosT);
         fx[I] += fxI; fy[I] += fyI; fz[I] += fzI;
         fx[K] += fxK; fy[K] += fyK; fz[K] += fzK;
         fx[J] -= (fxI + fxK); fy[J] -= (fyI + fyK); fz[J] -= (fzI + fzK);
         virial[0] += fxI * vij.x + fxK * vkj.x;
         virial[3] += fxI * vij.y + fxK * vkj.y;
         virial[4] += fxI * vij.z + fxK * vkj.z;
         virial[1] += fyI * vij.y + fyK * vkj.y;
         virial[5] += fyI * vij.z + fyK * vkj.z;
         virial[2] += fzI * vij.z + fzK * vkj.z;
      }
      /* resTorsionSorted :577-721 (func 1), resImproperSorted :723-848 (func 2) */
      for (int t = p->tors_off[rt]; t < p->tors_off[rt + 1]; t++)
      {
         int func = p->tors_func[t];
         if (func == 1 && (ex & torsionMask)) continue;
         if (func == 2 && (ex & improperMask)) continue;
         int I = AT(p->torsI[t]), J = AT(p->torsJ[t]), K = AT(p->torsK[t]), L = AT(p->torsL[t]);
         double ang, sinX, vir[6];
         vec3 dI, dJ, dK, dL;
         bioDihedralFast(p, rx, ry, rz, I, J, K, L, &ang, &sinX, &dI, &dJ, &dK, &dL, vir);
         double kk;
         if (func == 1)
         {
            double kchi = p->tors_k[t], delta = p->tors_delta[t];
            int nn = p->tors_n[t];
            etors += kchi * (1 + cos(nn * ang - delta));
            double absX = fabs(sinX);
            if (absX > FLOAT_EPS) kk = kchi * nn * sin(nn * ang - delta) / sinX;
            else
            {
               double nX = nn * ang, nX2 = nX * nX, nX4 = nX2 * nX2, nX6 = nX4 * nX2, nX8 = nX4 * nX4, nX10 = nX8 * nX2;
               double X2 = ang * ang, X4 = X2 * X2, X6 = X4 * X2, X8 = X4 * X4, X10 = X8 * X2;
               double ratio = nn * (1 - nX2 / 6 + nX4 / 120 - nX6 / 5040 + nX8 / 362880 - nX10 / 39916800) /
                              (1 - X2 / 6 + X4 / 120 - X6 / 5040 + X8 / 362880 - X10 / 39916800);
               orc_census[0]++;
               if (delta < NEAR_ZERO_ANGLE) { kk = kchi * nn * ratio; orc_census[1]++; }
               else if (delta > NEAR_180_ANGLE) { kk = -kchi * nn * ratio; orc_census[2]++; }
               else { kk = kchi * nn * ratio; orc_census[3]++; }
            }
         }
         else
         {
            double kpsi = p->tors_k[t], psi0 = p->tors_delta[t];
            double PI2 = 2 * M_PI, PI_1 = -1 * M_PI;
            double d = ang - psi0;
            if (d < PI_1) { d = d + PI2; orc_census[5]++; } else if (d > M_PI) { d = d - PI2; orc_census[5]++; }
            eimpr += kpsi * d * d;
            double absX = sinX < 0 ? -sinX : sinX;
            if (absX > FLOAT_EPS) kk = -2 * kpsi * d / sinX;
            else
            {
               orc_census[4]++;
               double i2 = ang * ang, i4 = i2 * i2, i6 = i4 * i2, i8 = i4 * i4, i10 = i8 * i2;
               kk = -2 * kpsi / (1 - i2 / 6 + i4 / 120 - i6 / 5040 + i8 / 362880 - i10 / 39916800);
            }
         }
         fx[I] -= dI.x * kk; fy[I] -= dI.y * kk; fz[I] -= dI.z * kk;
         fx[J] -= dJ.x * kk; fy[J] -= dJ.y * kk; fz[J] -= dJ.z * kk;
         fx[K] -= dK.x * kk; fy[K] -= dK.y * kk; fz[K] -= dK.z * kk;
         fx[L] -= dL.x * kk; fy[L] -= dL.y * kk; fz[L] -= dL.z * kk;
         for (int c = 0; c < 6; c++) virial[c] += vir[c] * kk;
      }
#undef AT
      first = last;
   }
   free(ord);
   e4[0] += ebond; e4[1] += eangle; e4[2] += etors; e4[3] += eimpr;
}

/* ------------------------------------------------------------------ */
/* ddcenergy (ddcenergy.c:160-238) single rank: zeroAll, martini() =
 * martiniNonBond + martiniIntraMoleReaction + charmmConvalent (bioMartini.c:1357-1390) */
void orc_forces(const orc_params *p, const orc_nbr *nb, int n,
                const double *rx, const double *ry, const double *rz,
                const uint64_t *gid, const int *species,
                double *fx, double *fy, double *fz, double *e, double *virial)
{
   for (int i = 0; i < n; i++) fx[i] = fy[i] = fz[i] = 0.0;          /* zeroAll :119-158 */
   for (int c = 0; c < 6; c++) virial[c] = 0.0;
   for (int c = 0; c < ORC_NE; c++) e[c] = 0.0;
   if ((p->excludePotentialTerm & nonBondMask) == 0)
   {
      orc_nonbond(p, nb, n, rx, ry, rz, species, fx, fy, fz, &e[ORC_E_LJ], &e[ORC_E_ELE], virial);
      if (p->nmoltype > 0) orc_intramol(p, nb, n, rx, ry, rz, species, fx, fy, fz, &e[ORC_E_ELE], virial);
   }
   double e4[4] = {0, 0, 0, 0};
   orc_bonded(p, n, rx, ry, rz, gid, species, fx, fy, fz, e4, virial);
   e[ORC_E_BOND] = e4[0]; e[ORC_E_ANGLE] = e4[1]; e[ORC_E_TORS] = e4[2]; e[ORC_E_IMPR] = e4[3];
   e[ORC_E_TOTAL] = e[ORC_E_LJ] + e[ORC_E_ELE] + e4[0] + e4[1] + e4[2] + e4[3];   /* e->eion */
   /* restraint() (restraint.c:259-361): a second POTENTIAL adding to eion, f and the virial */
   e[ORC_E_RESTRAINT] = 0.0;
   for (int r = 0; r < p->nrest; r++)
   {
      int ii = -1;
      for (int k = 0; k < n; k++) if (gid[k] == p->rest_gid[r]) { ii = k; break; }      /* restraintMap */
      if (ii < 0) continue;
      double L[3] = {p->hxx, p->hyy, p->hzz}, d[3], c[3], pos[3] = {rx[ii], ry[ii], rz[ii]};
      for (int a = 0; a < 3; a++)
      {
         double x0 = p->rest_r0[3 * r + a] * L[a];
         if (p->rest_origin == 0) x0 -= 0.5 * L[a];
         d[a] = pos[a] - x0;
      }
      int wrap = 0;
      for (int a = 0; a < 3; a++) if (p->rest_fc[3 * r + a] > 0 && fabs(d[a]) > 0.5 * L[a]) wrap = 1;
      if (wrap) nearestImage(p, &d[0], &d[1], &d[2]);
      for (int a = 0; a < 3; a++) c[a] = p->rest_fc[3 * r + a] * d[a];
      double kb = p->rest_kb[r];
      e[ORC_E_RESTRAINT] += kb * (c[0] * d[0] + c[1] * d[1] + c[2] * d[2]);
      double kforce = -2 * kb, f[3] = {kforce * c[0], kforce * c[1], kforce * c[2]};
      fx[ii] += f[0]; fy[ii] += f[1]; fz[ii] += f[2];
      virial[0] += f[0] * c[0]; virial[1] += f[1] * c[1]; virial[2] += f[2] * c[2];
      virial[3] += f[0] * c[1]; virial[4] += f[0] * c[2]; virial[5] += f[1] * c[2];
   }
   e[ORC_E_TOTAL] += e[ORC_E_RESTRAINT];
}

/* ------------------------------------------------------------------ */
void orc_brute_force(const orc_params *p, int n,
                     const double *rx, const double *ry, const double *rz,
                     const uint64_t *gid, const int *species,
                     double *fx, double *fy, double *fz, double *pvLJ, double *pvEle, double *virial,
                     long *npair_in_cut)
{
   double rc2 = p->rmax * p->rmax;
   double vLJ = 0, vEle = 0, q2 = 0;
   long np = 0;
   for (int i = 0; i < n; i++) { fx[i] = fy[i] = fz[i] = 0; double qi = p->charge[species[i]]; q2 += qi * qi; }
   for (int c = 0; c < 6; c++) virial[c] = 0;
   vEle = -0.5 * q2 * p->keR * p->crf;
   for (int i = 0; i < n; i++)
      for (int j = i + 1; j < n; j++)
      {
         double d[3] = {rx[i] - rx[j], ry[i] - ry[j], rz[i] - rz[j]};
         nearestImage(p, &d[0], &d[1], &d[2]);
         double r2 = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
         if (!(r2 < rc2)) continue;
         int pruned = pair_is_pruned(p, gid, species, i, j);
         double qq = p->keR * p->charge[species[i]] * p->charge[species[j]];
         double r = sqrt(r2);
         double dvdr_over_r;
         if (!pruned)
         {
            int sij = p->ljtype[species[j]] + p->nlj * p->ljtype[species[i]];
            double sr6 = pow(p->sigma[sij] / r, 6.0);
            vLJ += 4.0 * p->eps[sij] * (sr6 * sr6 - sr6) + p->shift[sij];
            vEle += qq * (1.0 / r + p->krf * r2 - p->crf);
            dvdr_over_r = 24.0 * p->eps[sij] * (sr6 - 2.0 * sr6 * sr6) / r2 + qq * (2.0 * p->krf - 1.0 / (r2 * r));
            np++;
         }
         else
         {
            vEle += qq * (p->krf * r2 - p->crf);
            dvdr_over_r = qq * 2.0 * p->krf;
         }
         double f[3] = {-dvdr_over_r * d[0], -dvdr_over_r * d[1], -dvdr_over_r * d[2]};
         fx[i] += f[0]; fy[i] += f[1]; fz[i] += f[2];
         fx[j] -= f[0]; fy[j] -= f[1]; fz[j] -= f[2];
         virial[0] += f[0] * d[0]; virial[1] += f[1] * d[1]; virial[2] += f[2] * d[2];
         virial[3] += f[0] * d[1]; virial[4] += f[0] * d[2]; virial[5] += f[1] * d[2];
      }
   *pvLJ = vLJ; *pvEle = vEle;
   if (npair_in_cut) *npair_in_cut = np;
}

/* ------------------------------------------------------------------ */
/* kinetic_terms, energy.c:48-163 */
void orc_kinetic(const orc_params *p, int n, const double *vx, const double *vy, const double *vz,
                 const int *species, double *prk, double *tion)
{
   double rk = 0.0, t[6] = {0, 0, 0, 0, 0, 0};
   for (int k = 0; k < n; k++)
   {
      double mass = p->mass[species[k]];
      double vxx = vx[k] * vx[k], vyy = vy[k] * vy[k], vzz = vz[k] * vz[k];
      double vxy = vx[k] * vy[k], vxz = vx[k] * vz[k], vyz = vy[k] * vz[k];
      double K = 0.5 * mass * (vxx + vyy + vzz);
      t[0] += mass * vxx; t[1] += mass * vyy; t[2] += mass * vzz;
      t[3] += mass * vxy; t[4] += mass * vxz; t[5] += mass * vyz;
      rk += K;
   }
   *prk = rk;
   for (int c = 0; c < 6; c++) tion[c] = t[c];
}

/* kinetic_terms, energy.c:104-147: the per-group / per-species copies and the thermal flux.  out[12 c + ..] for class c =
 * {rk, tion xx yy zz xy xz yz, mass, number, J x y z}.  U = potentialEnergy[k] and S = sion[k] are what the potentials
 * left per atom: the Martini path books e->eion / e->virial only (bioMartini.c:1111-1120), so both are zero here and
 * ge->eion / se->eion stay zero. */
void orc_kinetic_detail(const orc_params *p, int n, const double *vx, const double *vy, const double *vz,
                        const int *species, const int *group, int by_species, int nclass, double *out)
{
   for (int q = 0; q < 12 * nclass; q++) out[q] = 0.0;
   for (int k = 0; k < n; k++)
   {
      double mass = p->mass[species[k]];
      double vxx = vx[k] * vx[k], vyy = vy[k] * vy[k], vzz = vz[k] * vz[k];
      double vxy = vx[k] * vy[k], vxz = vx[k] * vz[k], vyz = vy[k] * vz[k];
      double K = 0.5 * mass * (vxx + vyy + vzz);
      double U = 0.0, Sxx = 0.0, Syy = 0.0, Szz = 0.0, Sxy = 0.0, Sxz = 0.0, Syz = 0.0;
      int c = by_species ? species[k] : group[k];
      if (c < 0 || c >= nclass) continue;
      double *e = out + 12 * c;
      e[0] += K;
      e[1] += mass * vxx; e[2] += mass * vyy; e[3] += mass * vzz; e[4] += mass * vxy; e[5] += mass * vxz; e[6] += mass * vyz;
      e[7] += mass; e[8] += 1.0;
      e[9] += (K + U) * vx[k] - 0.5 * (Sxx * vx[k] + Sxy * vy[k] + Sxz * vz[k]);
      e[10] += (K + U) * vy[k] - 0.5 * (Sxy * vx[k] + Syy * vy[k] + Syz * vz[k]);
      e[11] += (K + U) * vz[k] - 0.5 * (Sxz * vx[k] + Syz * vy[k] + Szz * vz[k]);
   }
}

/* eval_energyInfo, energyInfo.c:75-116 (global branch, single rank) */
void orc_energyinfo(const orc_params *p, double natoms, int nConstraints, double eion, double rk,
                    const double *virial, const double *tion, double *out)
{
   double vol = p->hxx * p->hyy * p->hzz;
   double sion[6];
   for (int c = 0; c < 6; c++) sion[c] = (virial[c] + tion[c]) * (1.0 / (-vol));   /* SMATACUM, SMATNORM(-vol) */
   out[0] = 2.0 * rk / (3.0 * natoms - nConstraints);
   out[1] = -(sion[0] + sion[1] + sion[2]) / 3.0;
   for (int c = 0; c < 6; c++) out[2 + c] = sion[c];
   out[8] = eion + rk;
}

void orc_group_temperature(const orc_params *p, int n, const double *vx, const double *vy, const double *vz,
                           const int *species, const int *group, int ngroup, orc_group *groups)
{
   /* energy.c:124-133 per-group rk/number, energyInfo.c:139 eg->temperature = 2 rk/(3 number) */
   double rk[64], num[64];
   assert(ngroup <= 64);
   for (int g = 0; g < ngroup; g++) { rk[g] = 0; num[g] = 0; }
   for (int k = 0; k < n; k++)
   {
      double mass = p->mass[species[k]];
      rk[group[k]] += 0.5 * mass * (vx[k] * vx[k] + vy[k] * vy[k] + vz[k] * vz[k]);
      num[group[k]] += 1.0;
   }
   for (int g = 0; g < ngroup; g++)
      if (num[g] > 0.0) groups[g].temperature = 2.0 * rk[g] / (3.0 * num[g]);
}

/* berendsen_Update FRONT_TIMESTEP, berendsen.c:30-62 */
static void berendsen_Update(orc_group *g, long loop, double dt_half)
{
   g->Tsum += g->temperature;
   g->nT += 1;
   double Tave = g->Tsum / g->nT;
   double ratio = (Tave == 0) ? 0 : g->Teq / Tave;
   if (g->tau != 0) g->lambda = sqrt(1 + (2.0 * dt_half / g->tau) * (ratio - 1));
   else g->lambda = sqrt(ratio);
   g->doScaling = 0;
   if (loop % g->interval == 0) { g->Tsum = 0; g->nT = 0; g->doScaling = 1; }
}

/* nglf, nglf.c:67-112 */
void orc_barostat(orc_params *p, int n, double *rx, double *ry, double *rz, const double virial[6],
                  double T, double P0, double beta, double tau, double dt)
{
   double vol = p->hxx * p->hyy * p->hzz, NkT = (double)n * T;        /* N molecules = n beads; kB = 1 */
   double pxx = (virial[0] + NkT) / vol - P0, pyy = (virial[1] + NkT) / vol - P0, pzz = (virial[2] + NkT) / vol - P0;
   double btt = beta * dt / tau;
   double Pxx = 0.5 * (pxx + pyy);                                     /* semi-isotropic: changeVolume */
   if (p->baro_isotropic) Pxx = pzz = (1.0 / 3.0) * (pxx + pyy + pzz);
   double l[3] = {cbrt(1.0 + Pxx * btt), cbrt(1.0 + Pxx * btt), cbrt(1.0 + pzz * btt)};
   for (int a = 0; a < 3; a++) if (fabs(l[a] - 1.0) < 1e-14) l[a] = 1.0;   /* box.c:44 */
   p->hxx *= l[0]; p->hyy *= l[1]; p->hzz *= l[2];
   for (int i = 0; i < n; i++) { rx[i] *= l[0]; ry[i] *= l[1]; rz[i] *= l[2]; }
}

/* neighborCheck (neighbor.c:117-208) for a constant box (the strain term is zero): the list
 * must be rebuilt once 2*max_i |(r_i - rbar) - (r0_i - rbar0)| reaches the skin deltaR.
 * Used when updateRate == 0 (ddcUpdateAll.c:64-71). */
int orc_neighbor_check(const orc_params *p, const orc_nbr *nb, int n, const double *rx, const double *ry, const double *rz)
{
   if (!nb || nb->n != n) return 1;
   double rbar[3] = {0, 0, 0};
   for (int i = 0; i < n; i++)
   {
      double x = rx[i], y = ry[i], z = rz[i];
      nearestImage(p, &x, &y, &z);
      rbar[0] += x; rbar[1] += y; rbar[2] += z;
   }
   for (int a = 0; a < 3; a++) rbar[a] /= (n > 0 ? n : 1);
   double d2max = 0.0;
   for (int i = 0; i < n; i++)
   {
      double x1 = rx[i] - rbar[0], y1 = ry[i] - rbar[1], z1 = rz[i] - rbar[2];
      nearestImage_fast(p, &x1, &y1, &z1);
      double x0 = nb->r0[0][i] - nb->rbar[0], y0 = nb->r0[1][i] - nb->rbar[1], z0 = nb->r0[2][i] - nb->rbar[2];
      nearestImage_fast(p, &x0, &y0, &z0);
      double x = x1 - x0, y = y1 - y0, z = z1 - z0;
      nearestImage_fast(p, &x, &y, &z);
      double dr2 = x * x + y * y + z * z;
      if (dr2 > d2max) d2max = dr2;
   }
   return (2.0 * sqrt(d2max) < p->deltaR) ? 0 : 1;
}

/* Langevin noise.  ddcMD draws three unit normals per particle and half step from a per-particle
 * LCG64 stream (lcg64.c, gasdev3d random.c:135-160) whose state travels with the particle.  The
 * device implementation's default replaces the stream by a counter-based one -- normals are a pure function
 * of (seed, gid, counter = 2*loop + {0 FRONT, 1 BACK}) -- and this restates exactly that function:
 * splitmix64 hashes, Box-Muller (statistical parity with the reference; the same numbers under every
 * decomposition).  The reference's own stream is below (orc_lcg64 ...): groups with `lcg` set draw from it. */
static unsigned long long smix64(unsigned long long z)
{
   z += 0x9E3779B97F4A7C15ull;
   z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
   z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
   return z ^ (z >> 31);
}
void orc_gauss3(unsigned long long seed, unsigned long long gid, unsigned long long counter, double g[3])
{
   const unsigned long long key = smix64(seed ^ smix64(gid)) + 4ull * counter;
   const double two53 = 1.0 / 9007199254740992.0;
   double u1 = ((double)(smix64(key) >> 11) + 0.5) * two53, u2 = ((double)(smix64(key + 1) >> 11) + 0.5) * two53;
   double u3 = ((double)(smix64(key + 2) >> 11) + 0.5) * two53, u4 = ((double)(smix64(key + 3) >> 11) + 0.5) * two53;
   double r = sqrt(-2.0 * log(u1)), t = 6.283185307179586476925 * u2;
   g[0] = r * cos(t); g[1] = r * sin(t);
   g[2] = sqrt(-2.0 * log(u3)) * cos(6.283185307179586476925 * u4);
}

/* RANDOM type LCG64: the reference's own per-particle stream.
 * lcg64_2 (lcg64.c:137-146): state = MULT[multID] * state + prime (mod 2^64), uniform = state * 2^-64, twice. */
static const unsigned long long LCG64_MULT[3] = {0x27bb2ee687b0b0fdull, 0x2c6fe96ee78b6955ull, 0x369dea0f31a53f85ull};
double orc_lcg64(orc_lcg64_parm *q)
{
   q->state = LCG64_MULT[q->multID] * q->state + q->prime;
   return (double)q->state * 5.4210108624275222e-20;      /* TWO_M64 */
}
/* gasdev3d (random.c:135-160): two polar (Marsaglia) draws; x, y from the first accepted pair, z from the second */
void orc_gasdev3d(orc_lcg64_parm *q, double g[3])
{
   double x, y, rsq, fac;
   do { x = orc_lcg64(q); y = orc_lcg64(q); x = 2.0 * x - 1.0; y = 2.0 * y - 1.0; rsq = x * x + y * y; } while (rsq >= 1.0 || rsq == 0.0);
   fac = sqrt(-2.0 * log(rsq) / rsq);
   g[0] = x * fac; g[1] = y * fac;
   do { x = orc_lcg64(q); y = orc_lcg64(q); x = 2.0 * x - 1.0; y = 2.0 * y - 1.0; rsq = x * x + y * y; } while (rsq >= 1.0 || rsq == 0.0);
   fac = sqrt(-2.0 * log(rsq) / rsq);
   g[2] = x * fac;
}
/* primes.c: nextPrime hands every task odd primes out of its own blocks of `blockSize` numbers above 2^31 + 1;
 * isPrime1 (:120-155) is a strong-probable-prime test whose modular products are Montgomery products (MMUL :74-86)
 * of operands that were never brought into Montgomery form -- i.e. a Miller-Rabin test to the bases
 * {2,3,5,7,11,13,17} * 2^-n mod N.  Restated as it stands. */
static unsigned long long pr_mmul(unsigned long long a, unsigned long long b, unsigned long long N, unsigned n)
{
   unsigned long long sum = 0;
   for (unsigned i = 0; i < n; i++) { sum += (a & 1) * b; sum += (sum & 1) * N; sum >>= 1; a >>= 1; }
   if (sum > N) sum -= N;
   return sum;
}
static unsigned long long pr_two2n(unsigned n, unsigned long long N)
{
   unsigned long long q = 1ull << n;
   if (q > N) q -= N;
   for (unsigned s = 0; s < n; s++) { q = 2 * q; if (q > N) q -= N; }
   return q;
}
static unsigned long long pr_ipow1(unsigned long long base, unsigned long long ex, unsigned long long N, unsigned n)
{
   unsigned long long result = 1;
   while (ex) { if (ex & 1) result = pr_mmul(result, base, N, n); ex >>= 1; base = pr_mmul(base, base, N, n); }
   return result;
}
int orc_is_prime1(unsigned long long N)
{
   static const unsigned long long small[7] = {2, 3, 5, 7, 11, 13, 17};
   if (N % 3 == 0 || N % 5 == 0 || N % 7 == 0 || N % 11 == 0 || N % 13 == 0) return 0;
   unsigned n = 0;
   for (unsigned long long m = N; m; m >>= 1) n++;      /* 1 + llog2(N) */
   unsigned long long s = N - 1, p2 = pr_two2n(n, N);
   unsigned r = 0;
   for (; s % 2 == 0; r++) s >>= 1;
   for (unsigned i = 0; i < 7 && small[i] < N - 1; i++)
   {
      unsigned long long x = pr_ipow1(small[i], s, N, n);
      if (x == 1 || x == N - 1) continue;
      int ok = 0;
      for (unsigned j = 1; j < r; j++)
      {
         x = pr_mmul(x, x, N, n); x = pr_mmul(x, p2, N, n);
         if (x == 1) return 0;
         if (x == N - 1) { ok = 1; break; }
      }
      if (!ok) return 0;
   }
   return 1;
}
void orc_prime_init(orc_primes *g, unsigned blockSize, unsigned taskId, unsigned nTasks)
{
   g->blockSize = blockSize; g->taskId = taskId; g->nTasks = nTasks;
   g->upperBound = 0; g->prime = 1; g->iBlock = 0;
}
unsigned long long orc_next_prime(orc_primes *g)      /* primes.c:35-63 */
{
   do
   {
      g->prime += 2;
      if (g->prime >= g->upperBound)
      {
         g->upperBound = (g->iBlock * g->nTasks + g->taskId) * g->blockSize + ((2ull << 30) + 1ull);
         g->prime = g->upperBound - g->blockSize;
         if (g->upperBound % 2 == 0) g->upperBound -= 1;
         if (g->prime % 2 == 0) g->prime += 1;
         g->iBlock++;
      }
   } while (!orc_is_prime1(g->prime));
   return g->prime;
}
/* collection.c:95-109 + lcg64_default (lcg64.c:98-110): particles that come without a random field get, in the order they
 * lie on the task, state = INIT_SEED ^ label, multID = 0,1,2,0,... and a fresh prime for every third particle
 * (ddcMD.c:70: prime_init(30000, rank, size)) */
void orc_lcg64_default(int n, const uint64_t *label, unsigned taskId, unsigned nTasks, orc_lcg64_parm *out)
{
   orc_primes g;
   orc_prime_init(&g, 30000, taskId, nTasks);
   unsigned long long prime = 0;
   for (int i = 0; i < n; i++)
   {
      if (i % 3 == 0) prime = orc_next_prime(&g);
      out[i].multID = (unsigned)(i % 3);
      out[i].state = 0x2bc6ffff8cfe166dull ^ label[i];
      out[i].prime = (unsigned)prime;
   }
}

void orc_nglf_step(const orc_params *p, orc_nbr **pnb, int updateRate, double dt,
                   long *loop, double *time, int n,
                   double *rx, double *ry, double *rz, double *vx, double *vy, double *vz,
                   double *fx, double *fy, double *fz,
                   const uint64_t *gid, const int *species, const int *group,
                   int ngroup, orc_group *groups,
                   double *e, double *virial, double *rk, double *tion)
{
   /* :74-78 FRONT velocityUpdate(dt/2): free.c:13-28 / berendsen.c:64-89 */
   for (int k = 0; k < n; k++)
   {
      const orc_group *g = &groups[group[k]];
      double mass = p->mass[species[k]];
      if (g->type == 2)
      {
         /* langevin_velocityUpdate FRONT_TIMESTEP (langevin.c:92-128), called with dt/2 */
         double dth = 0.5 * dt, al = exp(-dth / g->tau), c = dth / mass, d = sqrt(2.0 * dth * g->Teq / (mass * g->tau)), gg[3];
         if (g->lcg) orc_gasdev3d(&g->lcg[k], gg); else
         orc_gauss3(g->seed, gid[k], 2ull * (unsigned long long)(*loop), gg);
         /* :111-113  vx[k] = v.x + a*(vx[k]-v.x) + c*fx[k] + d*g.x */
         vx[k] = g->vcm[0] + al * (vx[k] - g->vcm[0]) + c * fx[k] + d * gg[0];
         vy[k] = g->vcm[1] + al * (vy[k] - g->vcm[1]) + c * fy[k] + d * gg[1];
         vz[k] = g->vcm[2] + al * (vz[k] - g->vcm[2]) + c * fz[k] + d * gg[2];
         continue;
      }
      if (g->type == 1 && g->doScaling == 1) { vx[k] *= g->lambda; vy[k] *= g->lambda; vz[k] *= g->lambda; }
      double a = (0.5 * dt) / mass;
      vx[k] += a * fx[k]; vy[k] += a * fy[k]; vz[k] += a * fz[k];
   }
   /* nglfconstraint.c:545 velocityConstraintOld(FRONT_TIMESTEP) */
   if (p->cons_off) orc_velocity_constraint(p, n, dt, 0, rx, ry, rz, vx, vy, vz, gid, species);
   /* :80-87 drift */
   for (int k = 0; k < n; k++) { rx[k] += dt * vx[k]; ry[k] += dt * vy[k]; rz[k] += dt * vz[k]; }
   /* :90 backInBox_fast */
   orc_back_in_box(p, n, rx, ry, rz);
   /* :92-95 */
   *time += dt;
   *loop += 1;
   /* :97 ddcenergy; rebuild test ddcUpdateAll.c:64-71 */
   if ((updateRate > 0 && (*loop % updateRate) == 0) || (updateRate == 0 && orc_neighbor_check(p, *pnb, n, rx, ry, rz)))
   {
      orc_nbr_free(*pnb);
      *pnb = orc_nbr_build(p, n, rx, ry, rz, gid, species);
   }
   orc_forces(p, *pnb, n, rx, ry, rz, gid, species, fx, fy, fz, e, virial);
   /* :100-104 BACK velocityUpdate(dt/2) */
   for (int k = 0; k < n; k++)
   {
      const orc_group *g = &groups[group[k]];
      double mass = p->mass[species[k]];
      if (g->type == 2)
      {
         /* BACK_TIMESTEP: v = a (v + c f + d g), loop already advanced */
         double dth = 0.5 * dt, al = exp(-dth / g->tau), c = dth / mass, d = sqrt(2.0 * dth * g->Teq / (mass * g->tau)), gg[3];
         if (g->lcg) orc_gasdev3d(&g->lcg[k], gg); else
         orc_gauss3(g->seed, gid[k], 2ull * (unsigned long long)(*loop) + 1ull, gg);
         /* :116-118  vx[k] = v.x + a*((vx[k]-v.x) + c*fx[k] + d*g.x) */
         vx[k] = g->vcm[0] + al * ((vx[k] - g->vcm[0]) + c * fx[k] + d * gg[0]);
         vy[k] = g->vcm[1] + al * ((vy[k] - g->vcm[1]) + c * fy[k] + d * gg[1]);
         vz[k] = g->vcm[2] + al * ((vz[k] - g->vcm[2]) + c * fz[k] + d * gg[2]);
         continue;
      }
      double a = (0.5 * dt) / mass;
      vx[k] += a * fx[k]; vy[k] += a * fy[k]; vz[k] += a * fz[k];
   }
   /* nglfconstraint.c:569 velocityConstraintOld(BACK_TIMESTEP) */
   if (p->cons_off) orc_velocity_constraint(p, n, dt, 1, rx, ry, rz, vx, vy, vz, gid, species);
   /* :105 kinetic_terms */
   orc_kinetic(p, n, vx, vy, vz, species, rk, tion);
   /* :108 group->Update(FRONT_TIMESTEP) */
   for (int g = 0; g < ngroup; g++)
      if (groups[g].type == 1) berendsen_Update(&groups[g], *loop, 0.5 * dt);
}

/* ------------------------------------------------------------------ */
/* nglfconstraint: velocity constraints and the molecular-pressure barostat */
int orc_velocity_constraint(const orc_params *p, int n, double dt, int location,
                            const double *rx, const double *ry, const double *rz, double *vx, double *vy, double *vz,
                            const uint64_t *gid, const int *species)
{
   const double tol = 1.0e-12;
   const int maxit = 500;
   int worst = 0;
   if (!p->cons_off || p->nresi <= 0 || p->cons_off[p->nresi] == 0) return 0;
   gid_order *ord = malloc(sizeof(gid_order) * (n > 0 ? n : 1));
   for (int i = 0; i < n; i++) { ord[i].id = i; ord[i].gid = gid[i]; }
   qsort(ord, n, sizeof(gid_order), cmp_gid);
   for (int first = 0; first < n;)
   {
      /* residue runs as charmmResidues cuts them (bioCharmmCovalent.c:48-93) */
      uint64_t key = ord[first].gid & molResMask;
      int last = first;
      while (last < n && (ord[last].gid & molResMask) == key) last++;
      int rt = p->resitype[species[ord[first].id]];
      int c0 = p->cons_off[rt], c1 = p->cons_off[rt + 1];
      for (int g0 = c0; g0 < c1;)
      {
         /* one CONSTRAINT = the pairs of one constraint list, in deck order */
         int g1 = g0;
         while (g1 < c1 && p->cons_grp[g1] == p->cons_grp[g0]) g1++;
         int np = g1 - g0;
         vec3 *rab = malloc(sizeof(vec3) * np);
         for (int ab = 0; ab < np; ab++)
         {
            int a = ord[first + p->consI[g0 + ab]].id, b = ord[first + p->consJ[g0 + ab]].id;
            rab[ab].x = rx[a] - rx[b]; rab[ab].y = ry[a] - ry[b]; rab[ab].z = rz[a] - rz[b];
            nearestImage(p, &rab[ab].x, &rab[ab].y, &rab[ab].z);
         }
         int it = 0;
         for (; it < maxit; it++)
         {
            double errMax = 0.0;
            for (int ab = 0; ab < np; ab++)
            {
               int a = ord[first + p->consI[g0 + ab]].id, b = ord[first + p->consJ[g0 + ab]].id;
               double dist2 = p->cons_r0[g0 + ab] * p->cons_r0[g0 + ab];
               double vabx = vx[a] - vx[b], vaby = vy[a] - vy[b], vabz = vz[a] - vz[b];
               double rma = 1.0 / p->mass[species[a]], rmb = 1.0 / p->mass[species[b]];
               double rvab;
               if (location == 0)
               {
                  double px = rab[ab].x + dt * vabx, py = rab[ab].y + dt * vaby, pz = rab[ab].z + dt * vabz;
                  rvab = (px * px + py * py + pz * pz - dist2) / (2 * dt);
               }
               else rvab = rab[ab].x * vabx + rab[ab].y * vaby + rab[ab].z * vabz;
               rvab /= dist2;
               double gab = -rvab / (rma + rmb);
               double err = fabs(rvab * dt);
               if (err > errMax) errMax = err;
               vx[a] += (rma * gab) * rab[ab].x; vy[a] += (rma * gab) * rab[ab].y; vz[a] += (rma * gab) * rab[ab].z;
               vx[b] -= (rmb * gab) * rab[ab].x; vy[b] -= (rmb * gab) * rab[ab].y; vz[b] -= (rmb * gab) * rab[ab].z;
            }
            if (errMax < tol) break;
         }
         if (it + 1 > worst) worst = it + 1;
         free(rab);
         g0 = g1;
      }
      first = last;
   }
   free(ord);
   return worst;
}

void orc_barostat_mol(orc_params *p, int n, double *rx, double *ry, double *rz,
                      const double *fx, const double *fy, const double *fz, const uint64_t *gid, const int *species,
                      const double virial[6], double T, double P0, double beta, double tau, double dt, double pmol[3])
{
   double v[3] = {virial[0], virial[1], virial[2]};
   gid_order *ord = malloc(sizeof(gid_order) * (n > 0 ? n : 1));
   for (int i = 0; i < n; i++) { ord[i].id = i; ord[i].gid = gid[i]; }
   qsort(ord, n, sizeof(gid_order), cmp_gid);
   long nmol = 0;
   for (int first = 0; first < n;)
   {
      uint64_t key = ord[first].gid & molMask;
      int last = first;
      while (last < n && (ord[last].gid & molMask) == key) last++;
      int i0 = ord[first].id, na = last - first;
      double M = 0.0, R[3] = {0, 0, 0};
      vec3 *d = malloc(sizeof(vec3) * na);
      for (int a = 0; a < na; a++)
      {
         int i = ord[first + a].id;
         double mass = p->mass[species[i]];
         d[a].x = rx[i] - rx[i0]; d[a].y = ry[i] - ry[i0]; d[a].z = rz[i] - rz[i0];
         nearestImage(p, &d[a].x, &d[a].y, &d[a].z);
         R[0] += mass * d[a].x; R[1] += mass * d[a].y; R[2] += mass * d[a].z;
         M += mass;
      }
      R[0] /= M; R[1] /= M; R[2] /= M;
      for (int a = 0; a < na; a++)
      {
         int i = ord[first + a].id;
         v[0] -= (d[a].x - R[0]) * fx[i]; v[1] -= (d[a].y - R[1]) * fy[i]; v[2] -= (d[a].z - R[2]) * fz[i];
      }
      free(d);
      nmol++;
      first = last;
   }
   free(ord);
   double vol = p->hxx * p->hyy * p->hzz, NkT = (double)nmol * T;
   double pxx = (v[0] + NkT) / vol, pyy = (v[1] + NkT) / vol, pzz = (v[2] + NkT) / vol;
   if (pmol) { pmol[0] = pxx; pmol[1] = pyy; pmol[2] = pzz; }
   pxx -= P0; pyy -= P0; pzz -= P0;
   double btt = beta * dt / tau;
   double Pxx = 0.5 * (pxx + pyy);
   if (p->baro_isotropic) Pxx = pzz = (1.0 / 3.0) * (pxx + pyy + pzz);
   double l[3] = {cbrt(1.0 + Pxx * btt), cbrt(1.0 + Pxx * btt), cbrt(1.0 + pzz * btt)};
   for (int a = 0; a < 3; a++) if (fabs(l[a] - 1.0) < 1e-14) l[a] = 1.0;
   p->hxx *= l[0]; p->hyy *= l[1]; p->hzz *= l[2];
   for (int i = 0; i < n; i++) { rx[i] *= l[0]; ry[i] *= l[1]; rz[i] *= l[2]; }
}
