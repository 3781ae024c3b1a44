/*
 * bonded.hip -- Martini bonded terms, the RESTRAINT potential and nglfconstraint on the device.
 * Bonded terms are evaluated bead-parallel, one lane per atom looping over its terms -- the layout of
 * the reference's CUDA path (bondedGPU.cu:1267-2672) -- with the formulas of the CPU reference:
 *   resBondSorted            bioCharmmCovalentEnergiesSorted.c:18-116
 *   resAngleSorted           :118-242   (func 1)
 *   resAngleCosineSorted     :244-363   (func 2)
 *   resAngleRestrainSorted   :365-487   (func 10)
 *   resTorsionSorted         :577-721   (func 1) via bioDihedralFast
 *   resImproperSorted        :723-848   (func 2)   (bioCharmmCovalentEnergies.c:266-351)
 * Separations use the rint-based nearestImage (Preduce, preduce.c:282-338) like bioVec.  A lane keeps
 * the force on its own atom (no atomics, fixed summation order); energies and virial use per-block
 * partials + a fixed-order second stage.
 */
#include "ddcmi_internal.h"
#include <math.h>
#include <array>
#include <map>
#include <string>
#include <unordered_map>

#define FLOAT_EPS 1e-08
#define NEAR_ZERO_ANGLE 0.017453292519943295
#define NEAR_180_ANGLE 3.12413936106985

struct BoxArgs { double L[3]; double Linv[3]; int pbc; };

__device__ __forceinline__ void bioVec(const BoxArgs &b, const double4 &p1, const double4 &p2, double &x, double &y, double &z)
{
   x = p1.x - p2.x; y = p1.y - p2.y; z = p1.z - p2.z;
   if (b.pbc & 1) { double da = -rint(b.Linv[0] * x); x += b.L[0] * da; }
   if (b.pbc & 2) { double db = -rint(b.Linv[1] * y); y += b.L[1] * db; }
   if (b.pbc & 4) { double dc = -rint(b.Linv[2] * z); z += b.L[2] * dc; }
}
__device__ __forceinline__ double wsum(double v)
{
#pragma unroll
   for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
   return v;
}
template <int NV>
__device__ __forceinline__ void block_store(double (&v)[NV], double *out)
{
   __shared__ double s_red[4][NV];
   int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
   for (int k = 0; k < NV; k++)
   {
      double s = wsum(v[k]);
      if (lane == 0) s_red[w][k] = s;
   }
   __syncthreads();
   if (threadIdx.x < NV) out[threadIdx.x] = (s_red[0][threadIdx.x] + s_red[1][threadIdx.x]) + (s_red[2][threadIdx.x] + s_red[3][threadIdx.x]);
}
/* resBondSorted (bioCharmmCovalentEnergiesSorted.c:18-116): fD = force on atom I, -fD on J */
__device__ __forceinline__ void bond_eval(const BoxArgs &box, const double4 &pI, const double4 &pJ, double kb, double b0,
                                          double &e, double (&fD)[3], double (&vir)[6])
{
   double x, y, z;
   bioVec(box, pI, pJ, x, y, z);
   /* b = r^2 / sqrt(r^2) and the unit vector by the reciprocal root: no division, no square root */
   const double r2 = x * x + y * y + z * z, ib = rsqrt_f64(r2), b = r2 * ib;
   double bDelta = b - b0;
   e = kb * bDelta * bDelta;
   double kforce = -2 * kb * bDelta * ib;
   fD[0] = kforce * x; fD[1] = kforce * y; fD[2] = kforce * z;
   vir[0] = fD[0] * x; vir[1] = fD[1] * y; vir[2] = fD[2] * z;
   vir[3] = fD[0] * y; vir[4] = fD[0] * z; vir[5] = fD[1] * z;
}
/* resAngleSorted / resAngleCosineSorted / resAngleRestrainSorted (...Sorted.c:118-487): forces on I and K, J gets -(fI+fK).
 * false: the kind is switched off by excludePotentialTerm */
template <bool WITH_F1 = true>      /* false: the caller holds no func-1 (acos/sin) angles -- a third fewer registers */
__device__ __forceinline__ bool angle_eval(const BoxArgs &box, const double4 &pI, const double4 &pJ, const double4 &pK, int f, double kt, double t0, int excl_mask,
                                           double &e, double (&fI)[3], double (&fK)[3], double (&vir)[6])
{
   if ((f == 1 && (excl_mask & 2)) || (f == 2 && (excl_mask & 4)) || (f == 10 && (excl_mask & 256))) return false;
   double ax, ay, az, cx, cy, cz;
   bioVec(box, pI, pJ, ax, ay, az);
   bioVec(box, pK, pJ, cx, cy, cz);
   /* 1/b_ij, 1/b_kj by reciprocal roots; the unit vectors and coefficients multiply by them */
   const double ib_ij = rsqrt_f64(ax * ax + ay * ay + az * az), ib_kj = rsqrt_f64(cx * cx + cy * cy + cz * cz);
   double uix = ax * ib_ij, uiy = ay * ib_ij, uiz = az * ib_ij;
   double ukx = cx * ib_kj, uky = cy * ib_kj, ukz = cz * ib_kj;
   double cosT = uix * ukx + uiy * uky + uiz * ukz;
   double coef_i, coef_k;
   if (WITH_F1 && f == 1)
   {
      double a = acos(cosT);
      double aDelta = a - t0;
      e = kt * aDelta * aDelta;
      double c = 2 * kt * aDelta / sin(a);
      coef_i = c * ib_ij;
      coef_k = c * ib_kj;
   }
   else if (f == 2)
   {
      double aDelta = cosT - t0;
      e = kt * aDelta * aDelta;
      coef_i = -2 * kt * aDelta * ib_ij;
      coef_k = -2 * kt * aDelta * ib_kj;
   }
   else
   {
      double isin2 = rcp_f64(1 - cosT * cosT);
      double aDelta = cosT - t0;
      e = kt * aDelta * aDelta * isin2;
      double coef_reb = -2 * kt * aDelta * (1 - cosT * t0) * (isin2 * isin2);
      coef_i = coef_reb * ib_ij;
      coef_k = coef_reb * ib_kj;
   }
   fI[0] = coef_i * (ukx - uix * cosT); fI[1] = coef_i * (uky - uiy * cosT); fI[2] = coef_i * (ukz - uiz * cosT);
   fK[0] = coef_k * (uix - ukx * cosT); fK[1] = coef_k * (uiy - uky * cosT); fK[2] = coef_k * (uiz - ukz * cosT);
   vir[0] = fI[0] * ax + fK[0] * cx; vir[1] = fI[1] * ay + fK[1] * cy; vir[2] = fI[2] * az + fK[2] * cz;
   vir[3] = fI[0] * ay + fK[0] * cy; vir[4] = fI[0] * az + fK[0] * cz; vir[5] = fI[1] * az + fK[1] * cz;
   return true;
}
/* bioDihedralFast (bioCharmmCovalentEnergies.c:266-351) + resTorsionSorted / resImproperSorted (...Sorted.c:577-848):
 * e_t = proper, e_i = improper energy; forces on I, J, K, L */
/* census of the rarely taken branches of the dihedral code (tests show with it that their inputs really drive
 * them): [0] torsion series (|sin phi| <= 1e-8), of these [1] delta ~ 0, [2] delta ~ pi, [3] any other delta;
 * [4] improper series; [5] improper difference wrapped by 2 pi; [6] cos phi clamped to +-1.  The adds sit inside
 * those branches: they cost nothing on the common path. */
__device__ unsigned long long g_branch_census[8];
extern "C" int ddcmi_debug_branch_census(unsigned long long out[8], int reset)
{
   if (!out) return DDCMI_EINVAL;
   if (hipDeviceSynchronize() != hipSuccess) return DDCMI_ENODEVICE;
   if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_branch_census), 8 * sizeof(unsigned long long)) != hipSuccess) return DDCMI_ENODEVICE;
   if (reset)
   {
      unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      if (hipMemcpyToSymbol(HIP_SYMBOL(g_branch_census), z, sizeof(z)) != hipSuccess) return DDCMI_ENODEVICE;
   }
   return DDCMI_OK;
}
__device__ __forceinline__ bool tors_eval(const BoxArgs &box, const double4 &pI, const double4 &pJ, const double4 &pK, const double4 &pL,
                                          int f, int n, double kpar, double dpar, int excl_mask,
                                          double &e_t, double &e_i, double (&fI)[3], double (&fJ)[3], double (&fK)[3], double (&fL)[3], double (&vir)[6])
{
   if ((f == 1 && (excl_mask & 16)) || (f == 2 && (excl_mask & 32))) return false;
   const double eps = 1e-12;
   double ax, ay, az, bx, by, bz, cx, cy, cz;
   bioVec(box, pI, pJ, ax, ay, az);
   bioVec(box, pJ, pK, bx, by, bz);
   bioVec(box, pK, pL, cx, cy, cz);
   double a2 = ax * ax + ay * ay + az * az, b2 = bx * bx + by * by + bz * bz, c2 = cx * cx + cy * cy + cz * cz;
   double ab = ax * bx + ay * by + az * bz, bc = bx * cx + by * cy + bz * cz, ac = ax * cx + ay * cy + az * cz;
   double ff = ab * bc - ac * b2;
   double g1 = a2 * b2 - ab * ab + eps;
   double g2 = b2 * c2 - bc * bc + eps;
   const double ig1 = rcp_f64(g1), ig2 = rcp_f64(g2);
   double yy = rsqrt_f64(g1 * g2);
   double xx = yy * ff;
   double xab = yy * bc + xx * ig1 * ab;
   double xbc = yy * ab + xx * ig2 * bc;
   double xac = -yy * b2;
   double xaa = -0.5 * xx * b2 * ig1;
   double xcc = -0.5 * xx * b2 * ig2;
   double xbb = -yy * ac - 0.5 * xx * (a2 * ig1 + c2 * ig2);
   double cax = xab * bx + xac * cx + (2 * xaa) * ax, cay = xab * by + xac * cy + (2 * xaa) * ay, caz = xab * bz + xac * cz + (2 * xaa) * az;
   double cbx = xab * ax + xbc * cx + (2 * xbb) * bx, cby = xab * ay + xbc * cy + (2 * xbb) * by, cbz = xab * az + xbc * cz + (2 * xbb) * bz;
   double ccx = xbc * bx + xac * ax + (2 * xcc) * cx, ccy = xbc * by + xac * ay + (2 * xcc) * cy, ccz = xbc * bz + xac * az + (2 * xcc) * cz;
   double mx = ay * bz - az * by, my = az * bx - ax * bz, mz = ax * by - ay * bx;
   double nx = by * cz - bz * cy, ny = bz * cx - bx * cz, nz = bx * cy - by * cx;
   double qx = my * nz - mz * ny, qy = mz * nx - mx * nz, qz = mx * ny - my * nx;
   double signnum = bx * qx + by * qy + bz * qz;
   double sign = (signnum < 0.0) ? -1.0 : 1.0;
   if (xx > 1.0 || xx < -1.0) atomicAdd(&g_branch_census[6], 1ull);
   xx = fmax(fmin(xx, 1.0), -1.0);
   double ang = sign * acos(xx);
   double sinX = sin(ang);
   double v0 = -(cax * ax + cbx * bx + ccx * cx), v3 = -(cax * ay + cbx * by + ccx * cy), v4 = -(cax * az + cbx * bz + ccx * cz);
   double v1 = -(cay * ay + cby * by + ccy * cy), v5 = -(cay * az + cby * bz + ccy * cz), v2 = -(caz * az + cbz * bz + ccz * cz);
   double kk;
   e_t = 0.0; e_i = 0.0;
   if (f == 1)
   {
      double kchi = kpar, delta = dpar;
      e_t = kchi * (1 + cos(n * ang - delta));
      if (fabs(sinX) > FLOAT_EPS) kk = kchi * n * sin(n * ang - delta) / sinX;
      else
      {
         double nX = n * ang, nX2 = nX * nX, nX4 = nX2 * nX2, nX6 = nX4 * nX2, nX8 = nX4 * nX4, nX10 = nX8 * nX2;
         double X2 = ang * ang, X4 = X2 * X2, X6 = X4 * X2, X8 = X4 * X4, X10 = X8 * X2;
         double ratio = n * (1 - nX2 / 6 + nX4 / 120 - nX6 / 5040 + nX8 / 362880 - nX10 / 39916800) /
                        (1 - X2 / 6 + X4 / 120 - X6 / 5040 + X8 / 362880 - X10 / 39916800);
         atomicAdd(&g_branch_census[0], 1ull);
         if (delta < NEAR_ZERO_ANGLE) { kk = kchi * n * ratio; atomicAdd(&g_branch_census[1], 1ull); }
         else if (delta > NEAR_180_ANGLE) { kk = -kchi * n * ratio; atomicAdd(&g_branch_census[2], 1ull); }
         else { kk = kchi * n * ratio; atomicAdd(&g_branch_census[3], 1ull); }
      }
   }
   else
   {
      double kpsi = kpar, psi0 = dpar;
      double d = ang - psi0;
      if (d < -M_PI) { d += 2 * M_PI; atomicAdd(&g_branch_census[5], 1ull); } else if (d > M_PI) { d -= 2 * M_PI; atomicAdd(&g_branch_census[5], 1ull); }
      e_i = kpsi * d * d;
      double absX = sinX < 0 ? -sinX : sinX;
      if (absX > FLOAT_EPS) kk = -2 * kpsi * d / sinX;
      else
      {
         atomicAdd(&g_branch_census[4], 1ull);
         double i2 = ang * ang, i4 = i2 * i2, i6 = i4 * i2, i8 = i4 * i4, i10 = i8 * i2;
         kk = -2 * kpsi / (1 - i2 / 6 + i4 / 120 - i6 / 5040 + i8 / 362880 - i10 / 39916800);
      }
   }
   fI[0] = -cax * kk; fI[1] = -cay * kk; fI[2] = -caz * kk;
   fJ[0] = -(cbx - cax) * kk; fJ[1] = -(cby - cay) * kk; fJ[2] = -(cbz - caz) * kk;
   fK[0] = -(ccx - cbx) * kk; fK[1] = -(ccy - cby) * kk; fK[2] = -(ccz - cbz) * kk;
   fL[0] = ccx * kk; fL[1] = ccy * kk; fL[2] = ccz * kk;
   vir[0] = v0 * kk; vir[1] = v1 * kk; vir[2] = v2 * kk; vir[3] = v3 * kk; vir[4] = v4 * kk; vir[5] = v5 * kk;
   return true;
}

/* bead-parallel kernel: one lane per bead walks the terms the bead takes part in, evaluates each
 * and keeps the force on its own atom -- a term is evaluated once per atom it has, but no force is added
 * atomically (the term-parallel kernels above are bound by the rate of double-precision atomic adds: 6 to
 * 12 per term).  Rows (built once in ddcmi_set_bonded, by caller-order atom index, terms ascending: a fixed
 * summation order) name the OTHER atoms of the term, the lane's role in it and a row of the table of
 * distinct parameter sets: a bond costs one 8-byte row read, one index translation and one bead record.
 * Energy and virial are booked by the lane holding role 0.
 * `slot` translates atom numbers to device slots: one domain -- caller-order index -> slot (slot_of_orig);
 * decomposed run -- atoms are numbered by their place in the sorted list of gids that occur in terms, and
 * slot_of_atom (refilled at every rebuild) holds the lowest slot carrying that gid here, owned copies
 * first, or INT_MAX.  A rank works on the atoms it owns (slot < nown): forces need no return traffic, and
 * every term's energy and virial are booked exactly once over all ranks, by the owner of its first atom. */
#define GB_NV 10      /* e_bond, e_angle, e_tors, e_impr, virial xx yy zz xy xz yz */
/* the lanes of one launch and their row patterns.  tab: ONE block of 16-byte pieces -- the patterns' headers {first A row, A rows, first B row,
 * B rows}, the A rows, the B rows ({other atoms - atom in term order, pid << 2 | role}), the two parameter tables; the offsets in pieces.  When
 * it has at most GB_TAB_PIECES of them (a force field's lipids: under 2 KB) every workgroup copies it to LDS and the trips of its lanes read
 * nothing from memory but their partners' records (k_bonded_gather<., true>). */
struct PatSet
{
   int nlanes; const int2 *desc;         /* [nlanes] {atom, pattern}: the launch's atoms in caller order, with filler lanes {nrow, 0} in front of a molecule that would straddle two workgroups (its partners are then all in the workgroup's LDS) */
   const int4 *tab; int pieces, rowA, rowB, parA, parB;
};
struct GatherRows
{
   int nrow;
   const int *boff, *aoff, *haoff, *toff; /* [nrow + 1] each; aoff: func 2/10 angles, haoff: func-1 angles */
   const int2 *brow;                     /* {partner, pid << 2 | role} */
   const int4 *arow, *harow;             /* {other atoms in term order, pid << 2 | role, 0} */
   const int4 *trow;                     /* {other atoms in term order, pid << 2 | role} */
   const double2 *bpar;                  /* {kb, b0} */
   const double4 *apar;                  /* {k, theta0, func, 0} */
   const double4 *tpar;                  /* {k, delta, func, n} */
   int nheavy; const int *hatoms;        /* atoms with func-1 angles or dihedrals */
   int nlight; const int *latoms;        /* atoms with bonds or func 2/10 angles, caller order: a molecule's atoms are neighbouring lanes */
   /* both launches read their rows as PATTERNS (round 5): a row names its partners by their distance in atom numbers, and atoms whose rows
    * then read the same -- the n-th atom of every copy of a molecule -- share one copy of them.  A bilayer's 1.28 M lipid atoms have 12
    * patterns: the rows (55 B per atom, a third of the light launch's traffic and 25 of its 72 us) no longer come from the HBM.  A system
    * without repetition has as many patterns as atoms and reads what it read before. */
   PatSet lp, hp;                        /* light launch: A = bond rows (int2), B = func 2/10 angle rows, parameters bpar | apar; heavy: A = func-1 angle rows, B = dihedral rows, apar | tpar */
};
#define GB_TAB_PIECES 384      /* 6 KB */
template <bool HEAVY, bool TABL = false>      /* HEAVY false: bonds and func 2/10 angles; true: func-1 angles and dihedrals (few terms, three times the registers).  TABL: row patterns and parameters in LDS */
__global__ __launch_bounds__(256) void k_bonded_gather(GatherRows gr, const int *__restrict__ slot, int nown, int ntot, BoxArgs box, int excl_mask,
                                                       const double4 *__restrict__ pos, double *fx, double *fy, double *fz, double4 *fb, double *partials, int pstride,
                                                       const double *__restrict__ hrecv3, const int *__restrict__ halo_src)
{
   /* a decomposed rank whose halo is staged straight from the exchange's receive buffer (ddcmi_ctx::halo_in_recv, round 6 for systems with
    * bonded terms too): a received bead's current position lies at hrecv3[3 k], k = -1 - halo_src[slot - nown]; its record in pos[] is the
    * rebuild's.  The terms read x y z only. */
   auto bead = [&](const int idx) -> double4
   {
      if (hrecv3 && idx >= nown)
      {
         const int k = halo_src[idx - nown];
         if (k < 0) { const double *r = hrecv3 + 3 * (size_t)(-1 - k); return make_double4(r[0], r[1], r[2], 0.0); }
      }
      return pos[idx];
   };
   /* lane = entry of the list of atoms that have terms of this launch, in caller order: the lanes of a
    * molecule sit together, so their rows are read with unit stride and the partners' bead records are the
    * neighbouring lanes' own -- taken from there (round 4): every lane leaves its own record in LDS, and a partner that is one of
    * the workgroup's atoms (for a molecule in the middle of the block: all of them) is read from LDS instead of through a slot
    * look-up and a scattered 32-byte gather each (five record gathers per bead became one: the light launch of the 2 M-bead
    * bilayer moved 0.35 GB in 32-byte pieces) */
   /* the lanes' bead records; behind the terms the same bytes take the lanes' GB_NV sums, [value][lane] */
   __shared__ double4 s_big[(GB_NV * 256 * sizeof(double)) / sizeof(double4)];
   double4 *const s_rec = s_big;
   __shared__ int s_atom[256];
   __shared__ int4 s_tab[TABL ? GB_TAB_PIECES : 1];
   const PatSet &ps = HEAVY ? gr.hp : gr.lp;
   if (TABL) for (int k = threadIdx.x; k < ps.pieces; k += 256) s_tab[k] = ps.tab[k];      /* (ordered by the barrier below) */
   const int4 *const t_hdr = TABL ? s_tab : ps.tab;
   const int4 *const t_rowA = t_hdr + ps.rowA, *const t_rowB = t_hdr + ps.rowB;
   const int4 *const t_parA = t_hdr + ps.parA, *const t_parB = t_hdr + ps.parB;
   const int2 *const t_brow = (const int2 *)t_rowA;
   const double2 *const t_bpar = (const double2 *)t_parA;
   const double4 *const t_apar = (const double4 *)(HEAVY ? t_parA : t_parB), *const t_tpar = (const double4 *)t_parB;
   const int j = blockIdx.x * 256 + threadIdx.x;
   double acc[GB_NV];
#pragma unroll
   for (int k = 0; k < GB_NV; k++) acc[k] = 0.0;
   int i = 0x7fffffff, o = gr.nrow, pat = 0, near = 1;
   if (j < ps.nlanes)
   {
      const int2 d = ps.desc[j];
      o = d.x; pat = d.y & 0x3fffffff; near = (d.y >> 30) & 1;
      if (o < gr.nrow) i = slot[o];
   }
   /* every partner of every lane of this wave is a lane of this workgroup (the rule for a molecule's run that fits one: PatSet::desc): their
    * records come out of LDS unasked (the look, compare and branch per partner were 6 of the light launch's 51 us) */
   const bool all_near = __all(near) != 0;
   const bool here = o < gr.nrow && (unsigned)i < (unsigned)ntot;      /* owned, or a halo copy on this rank */
   const double4 me = here ? bead(i) : make_double4(0.0, 0.0, 0.0, 0.0);
   s_atom[threadIdx.x] = here ? o : -1;
   s_rec[threadIdx.x] = me;
   __syncthreads();
   auto rec = [&](const int pa) -> double4
   {
      const int t = (int)threadIdx.x + (pa - o);
      if (all_near) return s_rec[t];
      if ((unsigned)t < 256u && s_atom[t] == pa) return s_rec[t];
      return bead(slot[pa]);
   };
   if (here && i < nown)
   {
      const int4 h = t_hdr[pat];
      /* light: bonds = the A rows, angles = the B rows; heavy: angles = the A rows, dihedrals = the B rows */
      const int b0 = HEAVY ? 0 : h.x, b1 = HEAVY ? 0 : h.x + h.y, a0 = HEAVY ? h.x : h.z, a1 = HEAVY ? h.x + h.y : h.z + h.w;
      const int t0 = HEAVY ? h.z : 0, t1 = HEAVY ? h.z + h.w : 0;
      if (b1 + a1 + t1 > b0 + a0 + t0)
      {
         double fxi = 0, fyi = 0, fzi = 0;
         /* three loops, each of one kind: the lanes of a wave run the same code in every trip */
         /* the row and the partner's slot of the NEXT trip are fetched while this trip's bead record is in
          * flight and its term is evaluated: one exposed memory latency per trip instead of three */
         if (!HEAVY && b1 > b0)
         {
         int2 row_n = t_brow[b0];
         for (int r = b0; r < b1; r++)
         {
            const int2 row = row_n;
            const double4 q = rec(o + row.x);
            if (r + 1 < b1) row_n = t_brow[r + 1];
            const int role = row.y & 3;
            const double2 par = t_bpar[row.y >> 2];
            double e, fD[3], vir[6];
            bond_eval(box, role == 0 ? me : q, role == 0 ? q : me, par.x, par.y, e, fD, vir);
            const double sg = role == 0 ? 1.0 : -1.0;
            fxi += sg * fD[0]; fyi += sg * fD[1]; fzi += sg * fD[2];
            if (role == 0)
            {
               acc[0] += e;
#pragma unroll
               for (int k = 0; k < 6; k++) acc[4 + k] += vir[k];
            }
         }
         }
         if (a1 > a0)
         {
         const int4 *arows = HEAVY ? t_rowA : t_rowB;
         const int rel = o;      /* (pattern rows name their atoms relative to the lane's) */
         int4 arow_n = arows[a0];
         for (int r = a0; r < a1; r++)
         {
            const int4 row = arow_n;
            const double4 q1 = rec(rel + row.x), q2 = rec(rel + row.y);
            if (r + 1 < a1) arow_n = arows[r + 1];
            const int role = row.z & 3;
            const double4 par = t_apar[row.z >> 2];
            double e, fI[3], fK[3], vir[6];
            if (angle_eval<HEAVY>(box, role == 0 ? me : q1, role == 0 ? q1 : (role == 1 ? me : q2), role == 2 ? me : q2, (int)par.z, par.x, par.y, excl_mask, e, fI, fK, vir))
            {
               if (role == 0)
               {
                  fxi += fI[0]; fyi += fI[1]; fzi += fI[2]; acc[1] += e;
#pragma unroll
                  for (int k = 0; k < 6; k++) acc[4 + k] += vir[k];
               }
               else if (role == 2) { fxi += fK[0]; fyi += fK[1]; fzi += fK[2]; }
               else { fxi -= fI[0] + fK[0]; fyi -= fI[1] + fK[1]; fzi -= fI[2] + fK[2]; }
            }
         }
         }
         if (HEAVY)
         for (int r = t0; r < t1; r++)
         {
            const int4 row = t_rowB[r];
            const int role = row.w & 3;
            const double4 par = t_tpar[row.w >> 2];
            const double4 q1 = rec(o + row.x), q2 = rec(o + row.y), q3 = rec(o + row.z);
            double et, ei, fI[3], fJ[3], fK[3], fL[3], vir[6];
            if (tors_eval(box, role == 0 ? me : q1, role == 0 ? q1 : (role == 1 ? me : q2), role <= 1 ? q2 : (role == 2 ? me : q3), role == 3 ? me : q3,
                          (int)par.z, (int)par.w, par.x, par.y, excl_mask, et, ei, fI, fJ, fK, fL, vir))
            {
               if (role == 0)
               {
                  fxi += fI[0]; fyi += fI[1]; fzi += fI[2]; acc[2] += et; acc[3] += ei;
#pragma unroll
                  for (int k = 0; k < 6; k++) acc[4 + k] += vir[k];
               }
               else if (role == 1) { fxi += fJ[0]; fyi += fJ[1]; fzi += fJ[2]; }
               else if (role == 2) { fxi += fK[0]; fyi += fK[1]; fzi += fK[2]; }
               else { fxi += fL[0]; fyi += fL[1]; fzi += fL[2]; }
            }
         }
         /* fb: the record array is all zero when the bonded kernels start and the light launch comes first: it stores (0 + f = f), the
          * heavy launch adds to the record of its few beads */
         if (fb) { if (HEAVY) { double4 b = fb[i]; b.x += fxi; b.y += fyi; b.z += fzi; fb[i] = b; } else fb[i] = make_double4(fxi, fyi, fzi, 0.0); }
         else { fx[i] += fxi; fy[i] += fyi; fz[i] += fzi; }
      }
   }
   /* block sums, GB_NV values per block at stride 16.  Every lane leaves its sums in LDS ([value][lane]: conflict-free), then wave w adds
    * up values w, w + 4, w + 8: four columns per lane in a fixed order and ONE wave reduction per value (ten reductions per wave of ten
    * lane values each were a sixth of the light launch's time) */
   double *const s_acc = (double *)s_big;
   __syncthreads();      /* (the last partner record has been read) */
#pragma unroll
   for (int k = 0; k < GB_NV; k++)
      if (HEAVY ? k != 0 : (k < 2 || k > 3)) s_acc[k * 256 + threadIdx.x] = acc[k];      /* the light launch has no dihedrals, the heavy one no bonds: those sums are zero */
   __syncthreads();
   const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
   for (int kk = 0; kk < (GB_NV + 3) / 4; kk++)
   {
      const int k = w + 4 * kk;
      if (k >= GB_NV) break;
      double sv = 0.0;
      if (HEAVY ? k != 0 : (k < 2 || k > 3))
         sv = wsum((s_acc[k * 256 + lane] + s_acc[k * 256 + 64 + lane]) + (s_acc[k * 256 + 128 + lane] + s_acc[k * 256 + 192 + lane]));
      if (lane == 0) partials[(size_t)k * pstride + blockIdx.x] = sv;      /* [value][workgroup]: k_reduce_gather reads whole cache lines */
   }
}
/* workgroup k sums column k of the gather kernel's partials in a fixed order and files it where
 * finish_energy expects the per-kind sums (the whole bonded virial goes to the bond block) */
#define RG_T 1024
/* row k of the [value][workgroup] sums of both launches (the light launch's workgroups, then the heavy one's): unit stride (rows of 16 values per
 * workgroup made every load instruction touch 64 cache lines: 8 us for the 5000 workgroups of the 2 M-bead bilayer); independent partial sums */
__device__ __forceinline__ double reduce_gather_row(const double *__restrict__ row, int nblocks, double *s)
{
   double p[4] = {0.0, 0.0, 0.0, 0.0};
   int b = threadIdx.x;
   for (; b + 3 * RG_T < nblocks; b += 4 * RG_T)
   {
#pragma unroll
      for (int u = 0; u < 4; u++) p[u] += row[b + u * RG_T];
   }
   for (int u = 0; b < nblocks; b += RG_T, u++) p[u] += row[b];
   s[threadIdx.x] = (p[0] + p[1]) + (p[2] + p[3]);
   __syncthreads();
   for (int off = RG_T / 2; off > 0; off >>= 1)
   {
      if (threadIdx.x < off) s[threadIdx.x] += s[threadIdx.x + off];
      __syncthreads();
   }
   return s[0];
}
__global__ __launch_bounds__(RG_T) void k_reduce_gather(const double *__restrict__ partials, int nblocks, int pstride, double *results)
{
   __shared__ double s[RG_T];
   const int k = blockIdx.x;
   const double sum = reduce_gather_row(partials + (size_t)k * pstride, nblocks, s);
   if (threadIdx.x == 0)
   {
      const int dst = k == 0 ? R_SCR_BOND : k == 1 ? R_SCR_ANGLE : k == 2 ? R_SCR_TORS : k == 3 ? R_SCR_TORS + 1 : R_SCR_BOND + 1 + (k - 4);
      results[dst] = sum;
      if (k == 1) for (int q = 1; q < 7; q++) results[R_SCR_ANGLE + q] = 0.0;
      if (k == 2) for (int q = 2; q < 8; q++) results[R_SCR_TORS + q] = 0.0;
   }
}
/* the lean steps' bonded sums, every pending step in one launch (ddcmi_lean_flush): step q's rows lie q * bstride doubles behind the first step's;
 * its GB_NV sums -- e_bond, e_angle, e_tors, e_impr, virial xx yy zz xy xz yz, the same additions in the same order as k_reduce_gather's --
 * go to hist[LEAN_HW q + 16 ...] */
__global__ __launch_bounds__(RG_T) void k_reduce_gather_hist(const double *__restrict__ partials, int nblocks, int pstride, size_t bstride, double *hist)
{
   __shared__ double s[RG_T];
   const int k = blockIdx.x, q = blockIdx.y;
   const double sum = reduce_gather_row(partials + (size_t)q * bstride + (size_t)k * pstride, nblocks, s);
   if (threadIdx.x == 0) hist[(size_t)LEAN_HW * q + 16 + k] = sum;
}
int ddcmi_lean_flush_bonded(ddcmi_ctx *ctx, int np)
{
   if (ctx->lean_bnblk <= 0 || !ctx->lean_bpart.p) return DDCMI_OK;
   hipLaunchKernelGGL(k_reduce_gather_hist, dim3(GB_NV, np), dim3(RG_T), 0, ctx->stream, ctx->lean_bpart.p, ctx->lean_bnblk, ctx->lean_bpstride, ctx->lean_bstride, ctx->lean_hist.p);
   return DDCMI_OK;
}

/* ---- decomposed runs: atoms of terms are named by gid and located among the owned + halo beads - */
#define GID_EMPTY 0xffffffffffffffffull
__device__ __forceinline__ unsigned gid_hash(uint64_t g, unsigned mask)
{
   g ^= g >> 33; g *= 0xff51afd7ed558ccdull; g ^= g >> 33; g *= 0xc4ceb9fe1a85ec53ull; g ^= g >> 33;
   return (unsigned)g & mask;
}
/* open-addressing table gid -> lowest device slot holding that gid.  Owned beads have
 * the lowest slots, so an owned copy wins over halo copies, and among periodic images
 * the choice does not depend on timing. */
__global__ void k_gid_insert(int n, const uint64_t *__restrict__ gid, unsigned mask, unsigned long long *keys, int *vals)
{
   int i = blockIdx.x * blockDim.x + threadIdx.x;
   if (i >= n) return;
   uint64_t g = gid[i];
   unsigned h = gid_hash(g, mask);
   for (;;)
   {
      unsigned long long old = atomicCAS(&keys[h], (unsigned long long)GID_EMPTY, (unsigned long long)g);
      if (old == GID_EMPTY || old == g) { atomicMin(&vals[h], i); return; }
      h = (h + 1) & mask;
   }
}
__device__ __forceinline__ int gid_find(uint64_t g, unsigned mask, const unsigned long long *keys, const int *vals)
{
   unsigned h = gid_hash(g, mask);
   for (;;)
   {
      unsigned long long k = keys[h];
      if (k == g) return vals[h];
      if (k == GID_EMPTY) return -1;
      h = (h + 1) & mask;
   }
}
/* rebuild: slot_of_atom[a] = lowest slot holding the a-th gid of the term atoms, INT_MAX if it is not here */
__global__ void k_atom_slots(int natom, const uint64_t *__restrict__ agid, unsigned mask, const unsigned long long *keys, const int *vals, int *slot_of_atom)
{
   int a = blockIdx.x * blockDim.x + threadIdx.x;
   if (a >= natom) return;
   int s = gid_find(agid[a], mask, keys, vals);
   slot_of_atom[a] = s < 0 ? 0x7fffffff : s;
}
/* rebuild: every partner of an owned atom must be present (owned or halo); flags[0] counts the missing ones */
__global__ void k_rows_check(GatherRows gr, const int *__restrict__ slot, int nown, int *flags)
{
   const int j = blockIdx.x * blockDim.x + threadIdx.x;
   int miss = 0;
   for (int pass = 0; pass < 2; pass++)
   {
      if (j >= (pass ? gr.nheavy : gr.nlight)) continue;
      const int o = pass ? gr.hatoms[j] : gr.latoms[j];
      if (slot[o] >= nown) continue;
      if (!pass)
      {
         for (int r = gr.boff[o]; r < gr.boff[o + 1]; r++) miss += slot[gr.brow[r].x] == 0x7fffffff;
         for (int r = gr.aoff[o]; r < gr.aoff[o + 1]; r++) miss += (slot[gr.arow[r].x] == 0x7fffffff) + (slot[gr.arow[r].y] == 0x7fffffff);
      }
      else
      {
         for (int r = gr.haoff[o]; r < gr.haoff[o + 1]; r++) miss += (slot[gr.harow[r].x] == 0x7fffffff) + (slot[gr.harow[r].y] == 0x7fffffff);
         for (int r = gr.toff[o]; r < gr.toff[o + 1]; r++) miss += (slot[gr.trow[r].x] == 0x7fffffff) + (slot[gr.trow[r].y] == 0x7fffffff) + (slot[gr.trow[r].z] == 0x7fffffff);
      }
   }
   if (miss) atomicAdd(&flags[0], miss);
}

template <class T>
static int up(ddcmi_ctx *ctx, dbuf<T> &buf, const T *src, size_t n)
{
   if (n == 0) return DDCMI_OK;
   ENSURE(ctx, buf, n);
   HIPCHK(ctx, hipMemcpyAsync(buf.p, src, n * sizeof(T), hipMemcpyHostToDevice, ctx->stream));
   HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
   return DDCMI_OK;
}

/* Rows of k_bonded_gather from term lists over atom numbers (caller-order indices, or -- decomposed runs --
 * positions in the sorted list of gids that occur in terms).  nbond = 0 when bonds are switched off. */
static int build_rows(ddcmi_ctx *ctx, int nbond, const int *bond_ij, const double *bond_kb, const double *bond_b0,
                      int nangle, const int *angle_ijk, const int *angle_func, const double *angle_k, const double *angle_t0,
                      int ntors, const int *tors_ijkl, const int *tors_func, const int *tors_n, const double *tors_k, const double *tors_delta)
{
   int rc;
   ctx->inc_nrow = 0; ctx->inc_light = 0; ctx->inc_lanes = 0; ctx->inc_hlanes = 0; ctx->inc_heavy = 0;
   ctx->nbond = nbond; ctx->nangle = nangle; ctx->ntors = ntors;
   if (nbond + nangle + ntors > 0)
   {
      for (int t = 0; t < nangle; t++)
         if (angle_func[t] != 1 && angle_func[t] != 2 && angle_func[t] != 10) SETERR(ctx, DDCMI_EINVAL, "angle %d: func %d is not 1, 2 or 10", t, angle_func[t]);
      for (int t = 0; t < ntors; t++)
         if (tors_func[t] != 1 && tors_func[t] != 2) SETERR(ctx, DDCMI_EINVAL, "dihedral %d: func %d is not 1 or 2", t, tors_func[t]);
      int amax = -1;
      for (int k = 0; k < 2 * nbond; k++) { if (bond_ij[k] < 0) SETERR(ctx, DDCMI_EINVAL, "negative atom index in a bond"); amax = std::max(amax, bond_ij[k]); }
      for (int k = 0; k < 3 * nangle; k++) { if (angle_ijk[k] < 0) SETERR(ctx, DDCMI_EINVAL, "negative atom index in an angle"); amax = std::max(amax, angle_ijk[k]); }
      for (int k = 0; k < 4 * ntors; k++) { if (tors_ijkl[k] < 0) SETERR(ctx, DDCMI_EINVAL, "negative atom index in a dihedral"); amax = std::max(amax, tors_ijkl[k]); }
      /* (indices are the caller's bead numbers of ddcmi_upload_state: checked against its count at the rebuild, when both are known; this bound keeps
       *  a wild index from sizing the row tables -- and it is the device's own, whose slot numbers are 29 bits in the tables of the decomposed path) */
      if (amax >= (1 << 29)) SETERR(ctx, DDCMI_EINVAL, "a bonded term names bead %d: more than 2^29 beads per device are not supported", amax);
      if (ctx->nloc > 0 && amax >= ctx->nloc && !ctx->bonded_gid) SETERR(ctx, DDCMI_EINVAL, "a bonded term names bead %d, the uploaded state holds %d", amax, ctx->nloc);
      for (int t = 0; t < nbond; t++)
         if (!std::isfinite(bond_kb[t]) || !std::isfinite(bond_b0[t])) SETERR(ctx, DDCMI_EINVAL, "bond %d: kb = %g, b0 = %g must be finite", t, bond_kb[t], bond_b0[t]);
      for (int t = 0; t < nangle; t++)
         if (!std::isfinite(angle_k[t]) || !std::isfinite(angle_t0[t])) SETERR(ctx, DDCMI_EINVAL, "angle %d: k = %g, theta0 = %g must be finite", t, angle_k[t], angle_t0[t]);
      for (int t = 0; t < ntors; t++)
         if (!std::isfinite(tors_k[t]) || !std::isfinite(tors_delta[t])) SETERR(ctx, DDCMI_EINVAL, "dihedral %d: k = %g, delta = %g must be finite", t, tors_k[t], tors_delta[t]);
      const int nrow = amax + 1;
      /* distinct parameter sets per kind (a force field has a handful) */
      std::map<std::array<double, 4>, int> ids[3];
      std::vector<double> par[3];
      auto pid = [&](int kind, double a, double b, double c, double d)
      {
         std::array<double, 4> key = {a, b, c, d};
         auto it = ids[kind].find(key);
         if (it != ids[kind].end()) return it->second;
         int id = (int)ids[kind].size();
         ids[kind][key] = id;
         if (kind == 0) { par[0].push_back(a); par[0].push_back(b); } else for (double v : key) par[kind].push_back(v);
         return id;
      };
      auto offsets = [&](const int *atoms, int nterm, int na)
      {
         std::vector<int> off((size_t)nrow + 1, 0);
         for (size_t k = 0; k < (size_t)nterm * na; k++) off[atoms[k] + 1]++;
         for (int a = 0; a < nrow; a++) off[a + 1] += off[a];
         return off;
      };
      /* func-1 angles go with the dihedrals into the second, heavier launch */
      std::vector<int> la, ha;
      for (int t = 0; t < nangle; t++) (angle_func[t] == 1 ? ha : la).push_back(t);
      auto offsets_sel = [&](const std::vector<int> &sel)
      {
         std::vector<int> off((size_t)nrow + 1, 0);
         for (int t : sel) for (int r = 0; r < 3; r++) off[angle_ijk[3 * t + r] + 1]++;
         for (int a = 0; a < nrow; a++) off[a + 1] += off[a];
         return off;
      };
      std::vector<int> boff = offsets(bond_ij, nbond, 2), aoff = offsets_sel(la), haoff = offsets_sel(ha), toff = offsets(tors_ijkl, ntors, 4);
      std::vector<int> brow(2 * (size_t)boff[nrow] + 2), arow(4 * (size_t)aoff[nrow] + 4), harow(4 * (size_t)haoff[nrow] + 4), trow(4 * (size_t)toff[nrow] + 4);
      {
         std::vector<int> fill(boff.begin(), boff.end() - 1);
         for (int t = 0; t < nbond; t++)
         {
            const int id = pid(0, bond_kb[t], bond_b0[t], 0, 0);
            for (int r = 0; r < 2; r++) { size_t w = fill[bond_ij[2 * t + r]]++; brow[2 * w] = bond_ij[2 * t + 1 - r]; brow[2 * w + 1] = (id << 2) | r; }
         }
      }
      for (int pass = 0; pass < 2; pass++)
      {
         const std::vector<int> &sel = pass ? ha : la;
         std::vector<int> &rows = pass ? harow : arow;
         const std::vector<int> &o = pass ? haoff : aoff;
         std::vector<int> fill(o.begin(), o.end() - 1);
         for (int t : sel)
         {
            const int id = pid(1, angle_k[t], angle_t0[t], (double)angle_func[t], 0);
            for (int r = 0; r < 3; r++)
            {
               size_t w = fill[angle_ijk[3 * t + r]]++;
               int q = 0;
               for (int a = 0; a < 3; a++) if (a != r) rows[4 * w + q++] = angle_ijk[3 * t + a];
               rows[4 * w + 2] = (id << 2) | r; rows[4 * w + 3] = 0;
            }
         }
      }
      std::vector<int> hatoms;
      for (int a = 0; a < nrow; a++) if (haoff[a + 1] > haoff[a] || toff[a + 1] > toff[a]) hatoms.push_back(a);
      ctx->inc_heavy = (int)hatoms.size();
      hatoms.push_back(0);
      std::vector<int> latoms;
      for (int a = 0; a < nrow; a++) if (boff[a + 1] > boff[a] || aoff[a + 1] > aoff[a]) latoms.push_back(a);
      ctx->inc_light = (int)latoms.size();
      /* molecules = connected components of the terms' graph: a run of a molecule's atoms that fits one workgroup does not straddle two */
      std::vector<int> root((size_t)nrow);
      for (int a = 0; a < nrow; a++) root[a] = a;
      auto find = [&](int a) { while (root[a] != a) { root[a] = root[root[a]]; a = root[a]; } return a; };
      auto join = [&](int a, int b) { a = find(a); b = find(b); if (a != b) root[std::max(a, b)] = std::min(a, b); };
      for (int t = 0; t < nbond; t++) join(bond_ij[2 * t], bond_ij[2 * t + 1]);
      for (int t = 0; t < nangle; t++) { join(angle_ijk[3 * t], angle_ijk[3 * t + 1]); join(angle_ijk[3 * t], angle_ijk[3 * t + 2]); }
      for (int t = 0; t < ntors; t++) for (int r = 1; r < 4; r++) join(tors_ijkl[4 * t], tors_ijkl[4 * t + r]);
      /* the lanes and row patterns of one launch (PatSet): an atom's rows with the partners as differences of atom numbers.  Rows of kind A are
       * wA ints wide with relA atom numbers in front, rows of kind B four ints with relB */
      struct HostPat { std::vector<int> desc, hdr, rowA, rowB; int nlanes = 0; };
      auto make_patterns = [&](const std::vector<int> &atoms, const std::vector<int> &offA, const std::vector<int> &rowsA, int wA, int relA,
                               const std::vector<int> &offB, const std::vector<int> &rowsB, int relB)
      {
         HostPat hp;
         std::unordered_map<std::string, int> seen;
         std::string key;
         std::vector<int> w;
         size_t run_end = 0;      /* index in atoms[] behind the current run of one molecule's atoms */
         for (size_t ai = 0; ai < atoms.size(); ai++)
         {
            const int a = atoms[ai];
            if (ai == run_end)
            {
               const int r0 = find(a);
               while (run_end < atoms.size() && find(atoms[run_end]) == r0) run_end++;
               const size_t len = run_end - ai, at = (hp.desc.size() / 2) % 256;
               if (len <= 256 && at + len > 256) for (size_t k = at; k < 256; k++) { hp.desc.push_back(nrow); hp.desc.push_back(0); }
            }
            const int nA = offA[a + 1] - offA[a], nB = offB[a + 1] - offB[a];
            w.clear();
            w.push_back(nA); w.push_back(nB);
            for (int r = offA[a]; r < offA[a + 1]; r++) for (int c = 0; c < wA; c++) w.push_back(rowsA[(size_t)wA * r + c] - (c < relA ? a : 0));
            for (int r = offB[a]; r < offB[a + 1]; r++) for (int c = 0; c < 4; c++) w.push_back(rowsB[4 * (size_t)r + c] - (c < relB ? a : 0));
            key.assign((const char *)w.data(), w.size() * sizeof(int));
            auto it = seen.find(key);
            int p;
            if (it != seen.end()) p = it->second;
            else
            {
               p = (int)seen.size();
               seen.emplace(key, p);
               hp.hdr.push_back((int)(hp.rowA.size() / wA)); hp.hdr.push_back(nA); hp.hdr.push_back((int)(hp.rowB.size() / 4)); hp.hdr.push_back(nB);
               hp.rowA.insert(hp.rowA.end(), w.begin() + 2, w.begin() + 2 + (size_t)wA * nA);
               hp.rowB.insert(hp.rowB.end(), w.begin() + 2 + (size_t)wA * nA, w.end());
            }
            hp.desc.push_back(a); hp.desc.push_back(p);
         }
         hp.nlanes = (int)(hp.desc.size() / 2);
         {
            /* bit 30 of a lane's pattern word: every partner of the atom is a lane of the same workgroup, at the distance of the atom numbers */
            std::vector<int> lane_of((size_t)nrow + 1, -1);
            for (int l = 0; l < hp.nlanes; l++) if (hp.desc[2 * (size_t)l] < nrow) lane_of[hp.desc[2 * (size_t)l]] = l;
            for (int l = 0; l < hp.nlanes; l++)
            {
               const int a = hp.desc[2 * (size_t)l];
               if (a >= nrow) { hp.desc[2 * (size_t)l + 1] |= 1 << 30; continue; }      /* (filler lanes walk no rows) */
               bool near = true;
               auto check = [&](int b) { const int lb = lane_of[b]; near = near && lb >= 0 && lb / 256 == l / 256 && lb - l == b - a; };
               for (int r = offA[a]; r < offA[a + 1]; r++) for (int c = 0; c < relA; c++) check(rowsA[(size_t)wA * r + c]);
               for (int r = offB[a]; r < offB[a + 1]; r++) for (int c = 0; c < relB; c++) check(rowsB[4 * (size_t)r + c]);
               if (near) hp.desc[2 * (size_t)l + 1] |= 1 << 30;
            }
         }
         hp.desc.push_back(nrow); hp.desc.push_back(1 << 30);
         hp.hdr.resize(hp.hdr.size() + 4, 0); hp.rowA.resize(hp.rowA.size() + 4, 0); hp.rowB.resize(hp.rowB.size() + 4, 0);      /* (the rows' read-ahead) */
         return hp;
      };
      latoms.push_back(0);
      {
         std::vector<int> fill(toff.begin(), toff.end() - 1);
         for (int t = 0; t < ntors; t++)
         {
            const int id = pid(2, tors_k[t], tors_delta[t], (double)tors_func[t], (double)tors_n[t]);
            for (int r = 0; r < 4; r++)
            {
               size_t w = fill[tors_ijkl[4 * t + r]]++;
               int q = 0;
               for (int a = 0; a < 4; a++) if (a != r) trow[4 * w + q++] = tors_ijkl[4 * t + a];
               trow[4 * w + 3] = (id << 2) | r;
            }
         }
      }
      for (int k = 0; k < 3; k++) if (ids[k].size() >= (1u << 29)) SETERR(ctx, DDCMI_EINVAL, "too many distinct bonded parameter sets");
      for (int k = 0; k < 3; k++) par[k].resize(par[k].size() + 4, 0.0);
      hatoms.pop_back(); latoms.pop_back();
      HostPat lpat = make_patterns(latoms, boff, brow, 2, 1, aoff, arow, 2), hpat = make_patterns(hatoms, haoff, harow, 4, 2, toff, trow, 3);
      hatoms.push_back(0); latoms.push_back(0);
      if (lpat.hdr.size() / 4 >= (1u << 30) || hpat.hdr.size() / 4 >= (1u << 30)) SETERR(ctx, DDCMI_EUNSUPPORTED, "more than 2^30 distinct bonded row patterns");
      ctx->inc_lanes = lpat.nlanes; ctx->inc_hlanes = hpat.nlanes;
      std::vector<int> tabs[2];
      for (int q = 0; q < 2; q++)
      {
         const HostPat &hp = q ? hpat : lpat;
         std::vector<int> &tb = tabs[q];
         auto put = [&](const void *src, size_t bytes) { const int at = (int)(tb.size() / 4); tb.resize(tb.size() + 4 * ((bytes + 15) / 16), 0); memcpy(&tb[4 * (size_t)at], src, bytes); return at; };
         put(hp.hdr.data(), hp.hdr.size() * sizeof(int));
         int *off = ctx->inc_tab_off[q];
         off[0] = put(hp.rowA.data(), hp.rowA.size() * sizeof(int));
         off[1] = put(hp.rowB.data(), hp.rowB.size() * sizeof(int));
         off[2] = put(par[q ? 1 : 0].data(), par[q ? 1 : 0].size() * sizeof(double));
         off[3] = put(par[q ? 2 : 1].data(), par[q ? 2 : 1].size() * sizeof(double));
         ctx->inc_tab_pieces[q] = (int)(tb.size() / 4);
      }
      if ((rc = up(ctx, ctx->inc_boff, boff.data(), boff.size())) || (rc = up(ctx, ctx->inc_aoff, aoff.data(), aoff.size())) || (rc = up(ctx, ctx->inc_toff, toff.data(), toff.size())) ||
          (rc = up(ctx, ctx->inc_hatoms, hatoms.data(), hatoms.size())) || (rc = up(ctx, ctx->inc_latoms, latoms.data(), latoms.size())) || (rc = up(ctx, ctx->inc_haoff, haoff.data(), haoff.size())) || (rc = up(ctx, ctx->inc_harow, harow.data(), harow.size())) ||
          (rc = up(ctx, ctx->inc_brow, brow.data(), brow.size())) || (rc = up(ctx, ctx->inc_arow, arow.data(), arow.size())) || (rc = up(ctx, ctx->inc_trow, trow.data(), trow.size())) ||
          (rc = up(ctx, ctx->inc_ldesc, lpat.desc.data(), lpat.desc.size())) || (rc = up(ctx, ctx->inc_hdesc, hpat.desc.data(), hpat.desc.size())) ||
          (rc = up(ctx, ctx->inc_tab, tabs[0].data(), tabs[0].size())) || (rc = up(ctx, ctx->inc_htab, tabs[1].data(), tabs[1].size())) ||
          (rc = up(ctx, ctx->inc_bpar, par[0].data(), par[0].size())) || (rc = up(ctx, ctx->inc_apar, par[1].data(), par[1].size())) || (rc = up(ctx, ctx->inc_tpar, par[2].data(), par[2].size()))) return rc;
      ctx->inc_nrow = nrow;
   }
   /* the per-kind sums are only written by kernels that run: clear stale ones */
   HIPCHK(ctx, hipMemsetAsync(ctx->d_results + R_SCR_BOND, 0, (R_RK - R_SCR_BOND) * sizeof(double), ctx->stream));
   HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
   ctx->forces_valid = false; ctx->list_valid = false;
   return DDCMI_OK;
}

extern "C" int ddcmi_set_bonded(ddcmi_ctx *ctx,
                                int nbond, const int *bond_ij, const double *bond_kb, const double *bond_b0,
                                int nangle, const int *angle_ijk, const int *angle_func, const double *angle_k, const double *angle_t0,
                                int ntors, const int *tors_ijkl, const int *tors_func, const int *tors_n, const double *tors_k, const double *tors_delta,
                                int excludePotentialTerm)
{
   ARGCHK(ctx, nbond < 0 || nangle < 0 || ntors < 0, "ddcmi_set_bonded: negative term count (%d bonds, %d angles, %d dihedrals)", nbond, nangle, ntors);
   ARGCHK(ctx, (nbond > 0 && (!bond_ij || !bond_kb || !bond_b0)) || (nangle > 0 && (!angle_ijk || !angle_func || !angle_k || !angle_t0)) ||
          (ntors > 0 && (!tors_ijkl || !tors_func || !tors_n || !tors_k || !tors_delta)), "ddcmi_set_bonded: an array of a term kind with terms is NULL");
   (void)hipSetDevice(ctx->device);
   ctx->excludePotentialTerm = excludePotentialTerm;
   ctx->bonded_gid = false; ctx->natom_g = 0;
   return build_rows(ctx, (excludePotentialTerm & 1) ? 0 : nbond, bond_ij, bond_kb, bond_b0, nangle, angle_ijk, angle_func, angle_k, angle_t0,
                     ntors, tors_ijkl, tors_func, tors_n, tors_k, tors_delta);
}

/* decomposed runs: every rank gets the whole system's terms with atoms named by gid.  The gids that occur
 * are sorted once; an atom's number is its place in that list, and the rows are built over these numbers. */
extern "C" int ddcmi_set_bonded_gid(ddcmi_ctx *ctx,
                                    int nbond, const uint64_t *bond_gid, const double *bond_kb, const double *bond_b0,
                                    int nangle, const uint64_t *angle_gid, const int *angle_func, const double *angle_k, const double *angle_t0,
                                    int ntors, const uint64_t *tors_gid, const int *tors_func, const int *tors_n, const double *tors_k, const double *tors_delta,
                                    int excludePotentialTerm)
{
   ARGCHK(ctx, nbond < 0 || nangle < 0 || ntors < 0, "ddcmi_set_bonded_gid: negative term count (%d bonds, %d angles, %d dihedrals)", nbond, nangle, ntors);
   ARGCHK(ctx, (nbond > 0 && (!bond_gid || !bond_kb || !bond_b0)) || (nangle > 0 && (!angle_gid || !angle_func || !angle_k || !angle_t0)) ||
          (ntors > 0 && (!tors_gid || !tors_func || !tors_n || !tors_k || !tors_delta)), "ddcmi_set_bonded_gid: an array of a term kind with terms is NULL");
   (void)hipSetDevice(ctx->device);
   ctx->excludePotentialTerm = excludePotentialTerm;
   ctx->bonded_gid = true;
   if (excludePotentialTerm & 1) nbond = 0;
   std::vector<uint64_t> ug;
   ug.reserve(2 * (size_t)nbond + 3 * (size_t)nangle + 4 * (size_t)ntors);
   ug.insert(ug.end(), bond_gid, bond_gid + 2 * (size_t)nbond);
   ug.insert(ug.end(), angle_gid, angle_gid + 3 * (size_t)nangle);
   ug.insert(ug.end(), tors_gid, tors_gid + 4 * (size_t)ntors);
   std::sort(ug.begin(), ug.end());
   ug.erase(std::unique(ug.begin(), ug.end()), ug.end());
   auto number = [&](const uint64_t *g, size_t n)
   {
      std::vector<int> out(n + 1);
      for (size_t k = 0; k < n; k++) out[k] = (int)(std::lower_bound(ug.begin(), ug.end(), g[k]) - ug.begin());
      return out;
   };
   std::vector<int> bi = number(bond_gid, 2 * (size_t)nbond), ai = number(angle_gid, 3 * (size_t)nangle), ti = number(tors_gid, 4 * (size_t)ntors);
   int rc;
   ctx->natom_g = (int)ug.size();
   if (ctx->natom_g > 0 && (rc = up(ctx, ctx->atom_gid, ug.data(), ug.size()))) return rc;
   return build_rows(ctx, nbond, bi.data(), bond_kb, bond_b0, nangle, ai.data(), angle_func, angle_k, angle_t0, ntors, ti.data(), tors_func, tors_n, tors_k, tors_delta);
}

/* ---- RESTRAINT potential (restraint.c:259-361) --------------------------------------- */
/* assignRestraintMap: the owned slot of each restrained gid (-1 when another rank owns it) */
__global__ void k_rest_locate(int nrest, const uint64_t *__restrict__ rgid, int nloc, unsigned mask, const unsigned long long *keys, const int *vals, int *slot)
{
   int r = blockIdx.x * blockDim.x + threadIdx.x;
   if (r >= nrest) return;
   int s = gid_find(rgid[r], mask, keys, vals);
   slot[r] = (s >= 0 && s < nloc) ? s : -1;
}
/* E = kb sum_c fc_c d_c^2, f_c = -2 kb fc_c d_c, virial += f (x) (fc d); d is the nearest image of
 * r - r0 (the reference applies nearestImage only when a restrained component exceeds half the
 * box: the same thing).  One workgroup, fixed summation order. */
__global__ __launch_bounds__(256) void k_restraint(int nrest, BoxArgs box, int origin, const int *__restrict__ slot, const int *__restrict__ fc,
                                                   const double *__restrict__ r0, const double *__restrict__ kb_, const double4 *__restrict__ pos,
                                                   double *fx, double *fy, double *fz, double4 *fb, double *out)
{
   double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
   for (int r = threadIdx.x; r < nrest; r += 256)
   {
      int i = slot[r];
      if (i < 0) continue;
      double4 p = pos[i];
      double d[3], c[3], pp[3] = {p.x, p.y, p.z};
      for (int a = 0; a < 3; a++)
      {
         double x0 = r0[3 * r + a] * box.L[a];
         if (origin == 0) x0 -= 0.5 * box.L[a];
         d[a] = pp[a] - x0;
         if (box.pbc >> a & 1) d[a] -= box.L[a] * rint(box.Linv[a] * d[a]);
         c[a] = fc[3 * r + a] * d[a];
      }
      double kb = kb_[r], kforce = -2 * kb;
      double f0 = kforce * c[0], f1 = kforce * c[1], f2 = kforce * c[2];
      if (fb) { atomicAdd(&fb[i].x, f0); atomicAdd(&fb[i].y, f1); atomicAdd(&fb[i].z, f2); }
      else { atomicAdd(&fx[i], f0); atomicAdd(&fy[i], f1); atomicAdd(&fz[i], f2); }
      acc[0] += kb * (c[0] * d[0] + c[1] * d[1] + c[2] * d[2]);
      acc[1] += f0 * c[0]; acc[2] += f1 * c[1]; acc[3] += f2 * c[2];
      acc[4] += f0 * c[1]; acc[5] += f0 * c[2]; acc[6] += f1 * c[2];
   }
   block_store<8>(acc, out);
}
extern "C" int ddcmi_set_restraints(ddcmi_ctx *ctx, int n, const uint64_t *gid, const int *fc, const double *r0, const double *kb, int origin)
{
   ARGCHK(ctx, n < 0 || (n > 0 && (!gid || !fc || !r0 || !kb)), "ddcmi_set_restraints: %d restraints%s", n, n < 0 ? "" : ", an array is NULL");
   for (int k = 0; k < n; k++)
      if (!std::isfinite(kb[k]) || !std::isfinite(r0[3 * k]) || !std::isfinite(r0[3 * k + 1]) || !std::isfinite(r0[3 * k + 2]))
         SETERR(ctx, DDCMI_EINVAL, "ddcmi_set_restraints: restraint %d has a constant or a reference position that is not finite", k);
   (void)hipSetDevice(ctx->device);
   int rc;
   ctx->nrest = n; ctx->rest_origin = origin;
   if (n > 0)
   {
      if ((rc = up(ctx, ctx->rest_gid, gid, (size_t)n)) || (rc = up(ctx, ctx->rest_fc, fc, 3 * (size_t)n)) ||
          (rc = up(ctx, ctx->rest_r0, r0, 3 * (size_t)n)) || (rc = up(ctx, ctx->rest_kb, kb, (size_t)n))) return rc;
      ENSURE(ctx, ctx->rest_slot, (size_t)n);
   }
   HIPCHK(ctx, hipMemsetAsync(ctx->d_results + R_SCR_REST, 0, 8 * sizeof(double), ctx->stream));
   HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
   ctx->forces_valid = false; ctx->list_valid = false;
   return DDCMI_OK;
}

static GatherRows gather_rows(const ddcmi_ctx *ctx)
{
   GatherRows gr = {ctx->inc_nrow, ctx->inc_boff.p, ctx->inc_aoff.p, ctx->inc_haoff.p, ctx->inc_toff.p, (const int2 *)ctx->inc_brow.p, (const int4 *)ctx->inc_arow.p,
                    (const int4 *)ctx->inc_harow.p, (const int4 *)ctx->inc_trow.p,
                    (const double2 *)ctx->inc_bpar.p, (const double4 *)ctx->inc_apar.p, (const double4 *)ctx->inc_tpar.p, ctx->inc_heavy, ctx->inc_hatoms.p, ctx->inc_light, ctx->inc_latoms.p,
                    {ctx->inc_lanes, (const int2 *)ctx->inc_ldesc.p, (const int4 *)ctx->inc_tab.p, ctx->inc_tab_pieces[0], ctx->inc_tab_off[0][0], ctx->inc_tab_off[0][1], ctx->inc_tab_off[0][2], ctx->inc_tab_off[0][3]},
                    {ctx->inc_hlanes, (const int2 *)ctx->inc_hdesc.p, (const int4 *)ctx->inc_htab.p, ctx->inc_tab_pieces[1], ctx->inc_tab_off[1][0], ctx->inc_tab_off[1][1], ctx->inc_tab_off[1][2], ctx->inc_tab_off[1][3]}};
   return gr;
}

/* decomposed runs, at every list rebuild: where are the atoms of the terms (and the restrained beads)?
 * One table gid -> lowest slot over the owned + halo beads, then slot_of_atom for the term atoms and a
 * check that no owned atom misses a partner. */
int ddcmi_bonded_localize(ddcmi_ctx *ctx)
{
   const bool terms = ctx->bonded_gid && ctx->inc_nrow > 0;
   const bool groups = (ctx->cons_gid && ctx->ncgroup > 0) || (ctx->mol_gid && ctx->nmol_multi > 0);
   if (!terms && ctx->nrest == 0 && !groups) return DDCMI_OK;      /* water without restraints: nothing to locate, no gid table */
   hipStream_t st = ctx->stream;
   const int nall = ctx->nloc + ctx->nhalo;
   unsigned cap = 1024;
   while (cap < 2u * (unsigned)std::max(nall, 1)) cap <<= 1;
   ctx->hmask = cap - 1;
   if (ctx->hkeys.ensure(cap) || ctx->hvals.ensure(cap)) SETERR(ctx, DDCMI_ENOMEM, "gid table of %u slots", cap);
   HIPCHK(ctx, hipMemsetAsync(ctx->hkeys.p, 0xff, (size_t)cap * sizeof(unsigned long long), st));
   HIPCHK(ctx, hipMemsetAsync(ctx->hvals.p, 0x7f, (size_t)cap * sizeof(int), st));
   HIPCHK(ctx, hipMemsetAsync(ctx->d_flags, 0, 8 * sizeof(int), st));
   if (nall > 0)
      hipLaunchKernelGGL(k_gid_insert, dim3(cdiv(nall, 256)), dim3(256), 0, st, nall, ctx->gid.p, ctx->hmask, ctx->hkeys.p, ctx->hvals.p);
   if (ctx->nrest > 0)
      hipLaunchKernelGGL(k_rest_locate, dim3(cdiv(ctx->nrest, 256)), dim3(256), 0, st, ctx->nrest, ctx->rest_gid.p, ctx->nloc, ctx->hmask, ctx->hkeys.p, ctx->hvals.p, ctx->rest_slot.p);
   if (groups) { int rcg = ddcmi_groups_localize(ctx); if (rcg) return rcg; }
   if (!terms) { HIPCHK(ctx, hipStreamSynchronize(st)); return DDCMI_OK; }
   ENSURE(ctx, ctx->slot_of_atom, (size_t)ctx->natom_g + 1);
   hipLaunchKernelGGL(k_atom_slots, dim3(cdiv(ctx->natom_g, 256)), dim3(256), 0, st, ctx->natom_g, ctx->atom_gid.p, ctx->hmask, ctx->hkeys.p, ctx->hvals.p, ctx->slot_of_atom.p);
   const int nl = std::max(ctx->inc_light, ctx->inc_heavy);
   if (nl > 0)
      hipLaunchKernelGGL(k_rows_check, dim3(cdiv(nl, 256)), dim3(256), 0, st, gather_rows(ctx), ctx->slot_of_atom.p, ctx->nloc, ctx->d_flags);
   HIPCHK(ctx, hipMemcpyAsync(ctx->h_flags, ctx->d_flags, sizeof(int), hipMemcpyDeviceToHost, st));
   HIPCHK(ctx, hipStreamSynchronize(st));
   if (ctx->h_flags[0] > 0)
      SETERR(ctx, DDCMI_EUNSUPPORTED, "%d bonded partners of owned beads lie beyond the halo (rmax+deltaR=%g): they are on no neighbouring domain's send list",
             ctx->h_flags[0], ctx->rmax + ctx->deltaR);
   return DDCMI_OK;
}

int ddcmi_launch_bonded(ddcmi_ctx *ctx, double4 *fb, int lean_slot)
{
   if (ctx->inc_nrow == 0 && ctx->nrest == 0) return DDCMI_OK;
   RoctxRange rng_cov("CHARMM_COVALENT");      /* charmmConvalent, bioCharmmCovalent.c:95-251 */
   hipStream_t st = ctx->stream;
   BoxArgs box;
   box.L[0] = ctx->h[0]; box.L[1] = ctx->h[4]; box.L[2] = ctx->h[8];
   for (int a = 0; a < 3; a++) box.Linv[a] = 1.0 / box.L[a];
   box.pbc = ctx->pbc;
   /* (restraints behind the bonded terms when the force goes to the record array: the light launch STORES its beads' records) */
   auto restraints = [&]()
   {
      if (ctx->nrest > 0)
         hipLaunchKernelGGL(k_restraint, dim3(1), dim3(256), 0, st, ctx->nrest, box, ctx->rest_origin, ctx->rest_slot.p, ctx->rest_fc.p, ctx->rest_r0.p, ctx->rest_kb.p,
                            ctx->pos.p, ctx->fx.p, ctx->fy.p, ctx->fz.p, fb, ctx->d_results + R_SCR_REST);
   };
   if (!fb || ctx->inc_nrow == 0) restraints();
   if (ctx->inc_nrow == 0) return DDCMI_OK;
   /* one lane per atom with terms; a decomposed run launches over the global atom lists and every rank
    * works on the atoms it owns */
   if (!ctx->bonded_gid) { int rcs = ddcmi_ensure_slots(ctx); if (rcs) return rcs; }
   const int *slot = ctx->bonded_gid ? ctx->slot_of_atom.p : ctx->slot_of_orig.p;
   const int nblk = cdiv(ctx->inc_lanes, 256), nblk2 = cdiv(ctx->inc_hlanes, 256);
   const int pstride = (nblk + nblk2 + 15) & ~15;
   ENSURE(ctx, ctx->bpartials, (size_t)pstride * GB_NV + 16);
   GatherRows gr = gather_rows(ctx);
   double *p1 = ctx->bpartials.p;
   if (lean_slot >= 0)
   {
      /* a lean step (launch_forces): the kernels' sums wait in the step's slot of a ring; one launch adds up every pending step's (ddcmi_lean_flush_bonded) */
      const size_t bstride = (size_t)pstride * GB_NV;
      ENSURE(ctx, ctx->lean_bpart, bstride * LEAN_W + 16);
      ctx->lean_bstride = bstride; ctx->lean_bpstride = pstride; ctx->lean_bnblk = nblk + nblk2;
      p1 = ctx->lean_bpart.p + bstride * (size_t)lean_slot;
   }
   double *p2 = p1 + nblk;      /* the heavy launch's workgroups follow the light one's in every row */
   const double *hrecv = ctx->halo_in_recv ? ctx->hrecv3.p : nullptr;      /* (received beads: their current positions are in the receive buffer) */
   static const bool no_lds_tab = getenv("DDCMI_NO_BONDED_LDS_TABLES") != nullptr;
   if (nblk > 0)
   {
      auto kl = (gr.lp.pieces <= GB_TAB_PIECES && !no_lds_tab) ? k_bonded_gather<false, true> : k_bonded_gather<false, false>;
      hipLaunchKernelGGL(kl, dim3(nblk), dim3(256), 0, st, gr, slot, ctx->nloc, ctx->nloc + ctx->nhalo, box, ctx->excludePotentialTerm,
                         ctx->pos.p, ctx->fx.p, ctx->fy.p, ctx->fz.p, fb, p1, pstride, hrecv, ctx->halo_src.p);
   }
   auto kh = (gr.hp.pieces <= GB_TAB_PIECES && !no_lds_tab) ? k_bonded_gather<true, true> : k_bonded_gather<true, false>;
   if (nblk2 > 0)
      hipLaunchKernelGGL(kh, dim3(nblk2), dim3(256), 0, st, gr, slot, ctx->nloc, ctx->nloc + ctx->nhalo, box, ctx->excludePotentialTerm,
                         ctx->pos.p, ctx->fx.p, ctx->fy.p, ctx->fz.p, fb, p2, pstride, hrecv, ctx->halo_src.p);
   if (fb) restraints();
   if (lean_slot < 0) hipLaunchKernelGGL(k_reduce_gather, dim3(GB_NV), dim3(RG_T), 0, st, ctx->bpartials.p, nblk + nblk2, pstride, ctx->d_results);
   return DDCMI_OK;
}

/* ------------------------------------------------------------------------------------------
 * nglfconstraint (one domain): velocity constraints, resMoveConsOld (nglfconstraint.c:180-264).
 * One lane per constraint group; the group's velocities, inverse masses and pair vectors live in
 * LDS ([item][lane], so a wavefront's lanes hit different banks) for the Gauss-Seidel sweeps:
 *   rvab = FRONT ((rab + dt vab)^2 - d^2) / (2 dt d^2) | BACK (rab . vab) / d^2
 *   gab = -rvab / (1/ma + 1/mb);  va += gab/ma rab;  vb -= gab/mb rab
 * until max |rvab dt| < 1e-12, at most 500 sweeps (the reference's tol and maxit).
 * status[0] = largest sweep count, status[1] = groups that hit maxit. */
#define CONS_T 64
template <int LOC>
__global__ __launch_bounds__(CONS_T) void k_constrain(int ngroups, const int *__restrict__ atom_off, const int *__restrict__ atoms, const int *__restrict__ pair_off,
                                                      const unsigned char *__restrict__ pa, const unsigned char *__restrict__ pb, const double *__restrict__ dist,
                                                      const int *__restrict__ slot, BoxArgs box, const double4 *__restrict__ pos, const int *__restrict__ species,
                                                      const double *__restrict__ invmass, double *vx, double *vy, double *vz, double dt, int maxA, int maxP, int *status, int nown)
{
   /* atoms == nullptr: slot[] is indexed by the position in the groups' atom lists (groups named by gid, decomposed
    * runs).  There every rank that owns an atom of a group solves the WHOLE group from the owned and halo copies of
    * its atoms (positions from the position halo, velocities from the velocity halo) and keeps the velocities of the
    * atoms it owns: slots below nown. */
   extern __shared__ double cons_sh[];
   const int t = threadIdx.x, g = blockIdx.x * CONS_T + t;
   double *v = cons_sh, *rm = cons_sh + 3 * maxA * CONS_T, *rab = rm + maxA * CONS_T;
   if (g >= ngroups) return;
   const int a0 = atom_off[g], na = atom_off[g + 1] - a0, p0 = pair_off[g], np = pair_off[g + 1] - p0;
#define CG_SLOT(k) (atoms ? slot[atoms[k]] : slot[k])
   {
      bool mine = false;
      for (int a = 0; a < na; a++) mine |= CG_SLOT(a0 + a) < nown;
      if (!mine) return;
   }
   for (int a = 0; a < na; a++)
   {
      int s = CG_SLOT(a0 + a);
      v[(3 * a + 0) * CONS_T + t] = vx[s]; v[(3 * a + 1) * CONS_T + t] = vy[s]; v[(3 * a + 2) * CONS_T + t] = vz[s];
      rm[a * CONS_T + t] = invmass[(int)((__double_as_longlong(pos[s].w) >> 16) & 0xffff)];      /* the record's tag names the species of owned and halo beads alike */
   }
   for (int ab = 0; ab < np; ab++)
   {
      int sa = CG_SLOT(a0 + pa[p0 + ab]), sb = CG_SLOT(a0 + pb[p0 + ab]);
      double x, y, z;
      bioVec(box, pos[sa], pos[sb], x, y, z);
      rab[(3 * ab + 0) * CONS_T + t] = x; rab[(3 * ab + 1) * CONS_T + t] = y; rab[(3 * ab + 2) * CONS_T + t] = z;
   }
   const double tol = 1.0e-12;
   const int maxit = 500;
   int it = 0;
   for (; it < maxit; it++)
   {
      double errMax = 0.0;
      for (int ab = 0; ab < np; ab++)
      {
         const int a = pa[p0 + ab], b = pb[p0 + ab];
         const double d = dist[p0 + ab], dist2 = d * d;
         const double rx = rab[(3 * ab + 0) * CONS_T + t], ry = rab[(3 * ab + 1) * CONS_T + t], rz = rab[(3 * ab + 2) * CONS_T + t];
         double *va = v + 3 * a * CONS_T + t, *vb = v + 3 * b * CONS_T + t;
         const double wx = va[0] - vb[0], wy = va[CONS_T] - vb[CONS_T], wz = va[2 * CONS_T] - vb[2 * CONS_T];
         const double rma = rm[a * CONS_T + t], rmb = rm[b * CONS_T + t];
         double rvab;
         if (LOC == 0)
         {
            double px = rx + dt * wx, py = ry + dt * wy, pz = rz + dt * wz;
            rvab = (px * px + py * py + pz * pz - dist2) / (2 * dt);
         }
         else rvab = rx * wx + ry * wy + rz * wz;
         rvab /= dist2;
         const double gab = -rvab / (rma + rmb);
         errMax = fmax(errMax, fabs(rvab * dt));
         const double ca = rma * gab, cb = rmb * gab;
         va[0] += ca * rx; va[CONS_T] += ca * ry; va[2 * CONS_T] += ca * rz;
         vb[0] -= cb * rx; vb[CONS_T] -= cb * ry; vb[2 * CONS_T] -= cb * rz;
      }
      if (errMax < tol) break;
   }
   for (int a = 0; a < na; a++)
   {
      int s = CG_SLOT(a0 + a);
      if (s >= nown) continue;
      vx[s] = v[(3 * a + 0) * CONS_T + t]; vy[s] = v[(3 * a + 1) * CONS_T + t]; vz[s] = v[(3 * a + 2) * CONS_T + t];
   }
#undef CG_SLOT
   atomicMax(status, it < maxit ? it + 1 : maxit);
   if (it == maxit) atomicAdd(status + 1, 1);
}

/* an offsets array of a caller's table: starts at 0, never decreases (tools/fuzz_abi.py: a table out of order sized a vector with a negative count) */
static int check_offsets(ddcmi_ctx *ctx, const char *who, const char *name, const int *off, int n)
{
   if (off[0] != 0) SETERR(ctx, DDCMI_EINVAL, "%s: %s[0] = %d, must be 0", who, name, off[0]);
   for (int k = 0; k < n; k++)
      if (off[k + 1] < off[k]) SETERR(ctx, DDCMI_EINVAL, "%s: %s decreases at %d (%d -> %d)", who, name, k, off[k], off[k + 1]);
   return DDCMI_OK;
}
template <class KEY>
static int set_constraints_impl(ddcmi_ctx *ctx, int ngroups, const int *pair_off, const KEY *pairI, const KEY *pairJ, const double *dist, bool by_gid)
{
   ARGCHK(ctx, ngroups < 0 || (ngroups > 0 && (!pair_off || !pairI || !pairJ || !dist)), "ddcmi_set_constraints: %d groups%s", ngroups, ngroups < 0 ? "" : ", an array is NULL");
   if (ngroups > 0) { int rco = check_offsets(ctx, "ddcmi_set_constraints", "pair_off", pair_off, ngroups); if (rco) return rco; }
   int amax = -1;
   if (ngroups > 0 && !by_gid)
      for (int k = 0; k < pair_off[ngroups]; k++)
      {
         if ((long long)pairI[k] < 0 || (long long)pairJ[k] < 0) SETERR(ctx, DDCMI_EINVAL, "ddcmi_set_constraints: pair %d names a negative bead index", k);
         amax = std::max(amax, (int)std::max(pairI[k], pairJ[k]));
      }
   if (amax >= 0 && ctx->nloc > 0 && amax >= ctx->nloc) SETERR(ctx, DDCMI_EINVAL, "ddcmi_set_constraints: a pair names bead %d, the uploaded state holds %d", amax, ctx->nloc);
   if (ngroups > 0 && !by_gid && (ctx->group_ || ctx->nranks > 1))
      SETERR(ctx, DDCMI_EINVAL, "with several domains constraint groups must be given by gid (ddcmi_set_constraints_gid): caller-order indices do not survive migration");
   (void)hipSetDevice(ctx->device);
   ctx->ncgroup = 0; ctx->ncpair = 0; ctx->cons_gid = false; ctx->cg_natom = 0; ctx->idx_amax_cons = -1;
   if (ngroups == 0) return DDCMI_OK;
   /* group-local atom lists (CONSTRAINT.atomIDList, bioMartini.c:405-425): pairs name positions in them */
   std::vector<int> aoff(ngroups + 1, 0);
   std::vector<KEY> alist;
   const int np_tot = pair_off[ngroups];
   std::vector<unsigned char> pa(np_tot), pb(np_tot);
   int maxA = 0, maxP = 0;
   for (int g = 0; g < ngroups; g++)
   {
      const int base = (int)alist.size();
      for (int k = pair_off[g]; k < pair_off[g + 1]; k++)
      {
         if (pairI[k] == pairJ[k] || !(dist[k] > 0.0) || !std::isfinite(dist[k])) SETERR(ctx, DDCMI_EINVAL, "constraint %d of group %d: atoms %lld %lld, distance %g", k - pair_off[g], g, (long long)pairI[k], (long long)pairJ[k], dist[k]);
         int loc[2];
         for (int e = 0; e < 2; e++)
         {
            const KEY at = e ? pairJ[k] : pairI[k];
            int f = -1;
            for (int q = base; q < (int)alist.size(); q++) if (alist[q] == at) { f = q - base; break; }
            if (f < 0) { f = (int)alist.size() - base; alist.push_back(at); }
            loc[e] = f;
         }
         if (loc[0] > 255 || loc[1] > 255) SETERR(ctx, DDCMI_EINVAL, "constraint group %d has more than 256 atoms", g);
         pa[k] = (unsigned char)loc[0]; pb[k] = (unsigned char)loc[1];
      }
      aoff[g + 1] = (int)alist.size();
      maxA = std::max(maxA, aoff[g + 1] - aoff[g]);
      maxP = std::max(maxP, pair_off[g + 1] - pair_off[g]);
   }
   if ((size_t)(4 * maxA + 3 * maxP) * CONS_T * sizeof(double) > 64 * 1024)
      SETERR(ctx, DDCMI_EINVAL, "constraint group of %d atoms / %d pairs exceeds the LDS budget of the solver (4 atoms + 3 pairs <= 128)", maxA, maxP);
   int rc;
   if (by_gid) { if ((rc = up(ctx, ctx->cg_atom_gid, (const uint64_t *)alist.data(), alist.size()))) return rc; ENSURE(ctx, ctx->cg_slot, alist.size() + 1); }
   else if ((rc = up(ctx, ctx->cg_atoms, (const int *)alist.data(), alist.size()))) return rc;
   ctx->cons_gid = by_gid; ctx->cg_natom = (int)alist.size(); ctx->list_valid = false;
   if ((rc = up(ctx, ctx->cg_atom_off, aoff.data(), (size_t)ngroups + 1)) ||
       (rc = up(ctx, ctx->cg_pair_off, pair_off, (size_t)ngroups + 1)) || (rc = up(ctx, ctx->cg_pa, pa.data(), (size_t)np_tot)) ||
       (rc = up(ctx, ctx->cg_pb, pb.data(), (size_t)np_tot)) || (rc = up(ctx, ctx->cg_dist, dist, (size_t)np_tot))) return rc;
   ENSURE(ctx, ctx->cons_status, 4);
   HIPCHK(ctx, hipMemsetAsync(ctx->cons_status.p, 0, 4 * sizeof(int), ctx->stream));
   HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
   ctx->ncgroup = ngroups; ctx->ncpair = np_tot; ctx->cons_maxA = maxA; ctx->cons_maxP = maxP; ctx->idx_amax_cons = amax;
   return DDCMI_OK;
}
extern "C" int ddcmi_set_constraints(ddcmi_ctx *ctx, int ngroups, const int *pair_off, const int *pairI, const int *pairJ, const double *dist)
{
   return set_constraints_impl<int>(ctx, ngroups, pair_off, pairI, pairJ, dist, false);
}
extern "C" int ddcmi_set_constraints_gid(ddcmi_ctx *ctx, int ngroups, const int *pair_off, const uint64_t *pairI, const uint64_t *pairJ, const double *dist)
{
   return set_constraints_impl<uint64_t>(ctx, ngroups, pair_off, pairI, pairJ, dist, true);
}

int ddcmi_launch_constraints(ddcmi_ctx *ctx, double dt, int location)
{
   if (ctx->ncgroup == 0) return DDCMI_OK;
   BoxArgs box;
   box.L[0] = ctx->h[0]; box.L[1] = ctx->h[4]; box.L[2] = ctx->h[8];
   for (int a = 0; a < 3; a++) box.Linv[a] = 1.0 / box.L[a];
   box.pbc = ctx->pbc;
   const size_t lds = (size_t)(4 * ctx->cons_maxA + 3 * ctx->cons_maxP) * CONS_T * sizeof(double);
   if (!ctx->cons_gid) { int rcs = ddcmi_ensure_slots(ctx); if (rcs) return rcs; }
   auto kern = location == 0 ? k_constrain<0> : k_constrain<1>;
   hipLaunchKernelGGL(kern, dim3(cdiv(ctx->ncgroup, CONS_T)), dim3(CONS_T), lds, ctx->stream, ctx->ncgroup, ctx->cg_atom_off.p,
                      ctx->cons_gid ? (const int *)nullptr : ctx->cg_atoms.p, ctx->cg_pair_off.p,
                      ctx->cg_pa.p, ctx->cg_pb.p, ctx->cg_dist.p, ctx->cons_gid ? ctx->cg_slot.p : ctx->slot_of_orig.p, box, ctx->pos.p, ctx->species.p, ctx->d_invmass.p,
                      ctx->vx.p, ctx->vy.p, ctx->vz.p, dt, ctx->cons_maxA, ctx->cons_maxP, ctx->cons_status.p, ctx->cons_gid ? ctx->nloc : 0x7fffffff);
   return DDCMI_OK;
}

extern "C" int ddcmi_constraint_stats(ddcmi_ctx *ctx, int *max_sweeps, int *unconverged, int reset)
{
   if (!ctx) return DDCMI_EINVAL;
   int h[2] = {0, 0};
   if (ctx->ncgroup > 0)
   {
      (void)hipSetDevice(ctx->device);
      HIPCHK(ctx, hipMemcpyAsync(h, ctx->cons_status.p, 2 * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
      if (reset) HIPCHK(ctx, hipMemsetAsync(ctx->cons_status.p, 0, 4 * sizeof(int), ctx->stream));
      HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
   }
   if (max_sweeps) *max_sweeps = h[0];
   if (unconverged) *unconverged = h[1];
   return DDCMI_OK;
}

/* molecularVirial (molecularPressure.c:23-56): sum over the beads of every molecule of (r - R) f on the
 * diagonal, R = the molecule's centre of mass; images are resolved relative to the molecule's first
 * listed atom (the reference uses its ownership species; any atom gives the same numbers while a
 * molecule is smaller than half the box).  One lane per molecule of two or more beads. */
__global__ __launch_bounds__(256) void k_mol_virial(int nmol, const int *__restrict__ mol_off, const int *__restrict__ atoms, const int *__restrict__ slot, BoxArgs box,
                                                    const double4 *__restrict__ pos, const int *__restrict__ species, const double *__restrict__ mass,
                                                    const double *__restrict__ fx, const double *__restrict__ fy, const double *__restrict__ fz, double *out)
{
   const int m = blockIdx.x * 256 + threadIdx.x;
   double acc[3] = {0, 0, 0};
   if (m < nmol)
   {
      const int a0 = mol_off[m], a1 = mol_off[m + 1];
      const double4 p0 = pos[slot[atoms[a0]]];
      double M = 0, Rx = 0, Ry = 0, Rz = 0;
      for (int a = a0; a < a1; a++)
      {
         int s = slot[atoms[a]];
         double x, y, z, w = mass[species[s]];
         bioVec(box, pos[s], p0, x, y, z);
         Rx += w * x; Ry += w * y; Rz += w * z; M += w;
      }
      Rx /= M; Ry /= M; Rz /= M;
      for (int a = a0; a < a1; a++)
      {
         int s = slot[atoms[a]];
         double x, y, z;
         bioVec(box, pos[s], p0, x, y, z);
         acc[0] += (x - Rx) * fx[s]; acc[1] += (y - Ry) * fy[s]; acc[2] += (z - Rz) * fz[s];
      }
   }
#pragma unroll
   for (int k = 0; k < 3; k++)
   {
      double w = wsum(acc[k]);
      if ((threadIdx.x & 63) == 0 && w != 0.0) atomicAdd(out + k, w);
   }
}

/* The same for molecules named by gid (decomposed runs).  A rank sums over the atoms it OWNS, positions taken as nearest
 * images about a reference point: a molecule all of whose atoms are here gives its whole term V - (P/M) o F (V = sum x o f,
 * P = sum m x, F = sum f; reference = its first atom); a SPLIT molecule (atoms on several ranks) gives V to the same sum
 * and leaves {P, F} in red[6k..] for the all-reduce -- the reference point is then the anchor all ranks agreed on at the
 * rebuild (ddcmi_mol_split_finish), and (P/M) o F is subtracted once the sums are complete (ddcmi_mol_split_term). */
__global__ __launch_bounds__(256) void k_mol_virial_gid(int nmol, int nown, const int *__restrict__ mol_off, const int *__restrict__ slot, const int *__restrict__ split,
                                                        const double *__restrict__ info, const double *__restrict__ mtot, BoxArgs box,
                                                        const double4 *__restrict__ pos, const double *__restrict__ mass,
                                                        const double *__restrict__ fx, const double *__restrict__ fy, const double *__restrict__ fz, double *red, double *out)
{
   const int m = blockIdx.x * 256 + threadIdx.x;
   double acc[3] = {0, 0, 0};
   if (m < nmol)
   {
      const int a0 = mol_off[m], a1 = mol_off[m + 1];
      const int k = split[m];
      double4 ref = make_double4(0, 0, 0, 0);
      bool any = false;
      for (int a = a0; a < a1; a++) any |= slot[a] < nown;
      if (any)
      {
         if (k >= 0) { ref.x = info[4 * m + 1]; ref.y = info[4 * m + 2]; ref.z = info[4 * m + 3]; }
         else ref = pos[slot[a0]];
         double P[3] = {0, 0, 0}, F[3] = {0, 0, 0}, V[3] = {0, 0, 0};
         for (int a = a0; a < a1; a++)
         {
            const int s = slot[a];
            if (s >= nown) continue;
            const double4 p = pos[s];
            double x, y, z;
            bioVec(box, p, ref, x, y, z);
            const double w = mass[(int)((__double_as_longlong(p.w) >> 16) & 0xffff)];
            P[0] += w * x; P[1] += w * y; P[2] += w * z;
            F[0] += fx[s]; F[1] += fy[s]; F[2] += fz[s];
            V[0] += x * fx[s]; V[1] += y * fy[s]; V[2] += z * fz[s];
         }
         if (k >= 0)
         {
            for (int c = 0; c < 3; c++) { red[6 * k + c] = P[c]; red[6 * k + 3 + c] = F[c]; acc[c] = V[c]; }
         }
         else
         {
            const double iM = 1.0 / mtot[m];
            for (int c = 0; c < 3; c++) acc[c] = V[c] - (P[c] * iM) * F[c];
         }
      }
   }
#pragma unroll
   for (int c = 0; c < 3; c++)
   {
      double w = wsum(acc[c]);
      if ((threadIdx.x & 63) == 0 && w != 0.0) atomicAdd(out + c, w);
   }
}
/* rebuild: mol_info[4m] = 1 if this rank owns an atom of molecule m, [4m+1..3] = position of its first listed atom if
 * this rank owns it; summed over the ranks that gives {owning ranks, anchor} */
__global__ void k_mol_info(int nmol, int nown, const int *__restrict__ mol_off, const int *__restrict__ slot, const double4 *__restrict__ pos, double *info)
{
   const int m = blockIdx.x * blockDim.x + threadIdx.x;
   if (m >= nmol) return;
   const int a0 = mol_off[m], a1 = mol_off[m + 1];
   bool any = false;
   for (int a = a0; a < a1; a++) any |= slot[a] < nown;
   double4 o = make_double4(any ? 1.0 : 0.0, 0, 0, 0);
   if (slot[a0] < nown) { const double4 p = pos[slot[a0]]; o.y = p.x; o.z = p.y; o.w = p.z; }
   info[4 * m] = o.x; info[4 * m + 1] = o.y; info[4 * m + 2] = o.z; info[4 * m + 3] = o.w;
}
__global__ void k_mol_split_flag(int nmol, const double *__restrict__ info, int *flag)
{
   const int m = blockIdx.x * blockDim.x + threadIdx.x;
   if (m < nmol) flag[m] = info[4 * m] > 1.5 ? 1 : 0;
}
__global__ void k_mol_split_index(int nmol, const double *__restrict__ info, int *split /* in: exclusive scan of the flags */)
{
   const int m = blockIdx.x * blockDim.x + threadIdx.x;
   if (m < nmol) split[m] = info[4 * m] > 1.5 ? split[m] : -1;
}
/* every atom of a molecule some of whose atoms are owned must be visible as far as THIS rank needs it: only the owned
 * ones are read, so nothing to check; constraint groups need all their atoms: flags[0] counts groups with an owned
 * and a missing atom */
__global__ void k_groups_check(int ngroups, int nown, const int *__restrict__ atom_off, const int *__restrict__ slot, int *flags)
{
   const int g = blockIdx.x * blockDim.x + threadIdx.x;
   if (g >= ngroups) return;
   bool mine = false, missing = false;
   for (int a = atom_off[g]; a < atom_off[g + 1]; a++) { mine |= slot[a] < nown; missing |= slot[a] == 0x7fffffff; }
   if (mine && missing) atomicAdd(&flags[0], 1);
}
/* sum over the split molecules of (P/M) o F from the all-reduced {P, F} */
__global__ __launch_bounds__(256) void k_mol_split_term(int nmol, const int *__restrict__ split, const double *__restrict__ mtot, const double *__restrict__ red, double *out)
{
   const int m = blockIdx.x * 256 + threadIdx.x;
   double acc[3] = {0, 0, 0};
   if (m < nmol && split[m] >= 0)
   {
      const int k = split[m];
      const double iM = 1.0 / mtot[m];
      for (int c = 0; c < 3; c++) acc[c] = (red[6 * k + c] * iM) * red[6 * k + 3 + c];
   }
#pragma unroll
   for (int c = 0; c < 3; c++)
   {
      double w = wsum(acc[c]);
      if ((threadIdx.x & 63) == 0 && w != 0.0) atomicAdd(out + c, w);
   }
}

extern "C" int ddcmi_set_molecule_lists_gid(ddcmi_ctx *ctx, long nmol_total, int nmulti, const int *mol_off, const uint64_t *mol_atom_gid, const double *mol_mass)
{
   ARGCHK(ctx, nmol_total < 0 || nmulti < 0 || (nmulti > 0 && (!mol_off || !mol_atom_gid || !mol_mass)), "ddcmi_set_molecule_lists_gid: %ld molecules, %d of several beads%s", nmol_total, nmulti, (nmol_total < 0 || nmulti < 0) ? "" : ", an array is NULL");
   if (nmulti > 0) { int rco = check_offsets(ctx, "ddcmi_set_molecule_lists_gid", "mol_off", mol_off, nmulti); if (rco) return rco; }
   for (int m = 0; m < nmulti; m++)
      if (!(mol_mass[m] > 0.0) || !std::isfinite(mol_mass[m])) SETERR(ctx, DDCMI_EINVAL, "ddcmi_set_molecule_lists_gid: molecule %d has mass %g", m, mol_mass[m]);
   (void)hipSetDevice(ctx->device);
   int rc;
   ctx->nmol_total = nmol_total; ctx->nmol_multi = 0; ctx->molv_valid = false; ctx->mol_gid = true; ctx->mol_natom = 0; ctx->nsplit = 0; ctx->list_valid = false;
   if (nmulti > 0)
   {
      const size_t na = (size_t)mol_off[nmulti];
      if ((rc = up(ctx, ctx->mol_off, mol_off, (size_t)nmulti + 1)) || (rc = up(ctx, ctx->mol_atom_gid, mol_atom_gid, na)) || (rc = up(ctx, ctx->mol_mtot, mol_mass, (size_t)nmulti))) return rc;
      ENSURE(ctx, ctx->mol_slot, na + 1); ENSURE(ctx, ctx->mol_split, (size_t)nmulti + 1); ENSURE(ctx, ctx->mol_info, 4 * (size_t)nmulti + 4);
      ctx->nmol_multi = nmulti; ctx->mol_natom = (int)na;
   }
   return DDCMI_OK;
}

/* rebuild (after the gid table of ddcmi_bonded_localize): where are the atoms of the constraint groups and of the molecules? */
int ddcmi_groups_localize(ddcmi_ctx *ctx)
{
   hipStream_t st = ctx->stream;
   if (ctx->cons_gid && ctx->ncgroup > 0)
   {
      hipLaunchKernelGGL(k_atom_slots, dim3(cdiv(ctx->cg_natom, 256)), dim3(256), 0, st, ctx->cg_natom, ctx->cg_atom_gid.p, ctx->hmask, ctx->hkeys.p, ctx->hvals.p, ctx->cg_slot.p);
      HIPCHK(ctx, hipMemsetAsync(ctx->d_flags, 0, sizeof(int), st));
      hipLaunchKernelGGL(k_groups_check, dim3(cdiv(ctx->ncgroup, 256)), dim3(256), 0, st, ctx->ncgroup, ctx->nloc, ctx->cg_atom_off.p, ctx->cg_slot.p, ctx->d_flags);
      HIPCHK(ctx, hipMemcpyAsync(ctx->h_flags, ctx->d_flags, sizeof(int), hipMemcpyDeviceToHost, st));
      HIPCHK(ctx, hipStreamSynchronize(st));
      if (ctx->h_flags[0] > 0)
         SETERR(ctx, DDCMI_EUNSUPPORTED, "%d constraint groups have an atom on this domain and a partner beyond its halo (rmax+deltaR=%g)", ctx->h_flags[0], ctx->rmax + ctx->deltaR);
   }
   if (ctx->mol_gid && ctx->nmol_multi > 0)
   {
      hipLaunchKernelGGL(k_atom_slots, dim3(cdiv(ctx->mol_natom, 256)), dim3(256), 0, st, ctx->mol_natom, ctx->mol_atom_gid.p, ctx->hmask, ctx->hkeys.p, ctx->hvals.p, ctx->mol_slot.p);
      hipLaunchKernelGGL(k_mol_info, dim3(cdiv(ctx->nmol_multi, 256)), dim3(256), 0, st, ctx->nmol_multi, ctx->nloc, ctx->mol_off.p, ctx->mol_slot.p, ctx->pos.p, ctx->mol_info.p);
   }
   return DDCMI_OK;
}
/* rebuild, after mol_info has been summed over the ranks: index the split molecules (the same numbers on every rank) */
int ddcmi_mol_split_finish(ddcmi_ctx *ctx)
{
   ctx->nsplit = 0;
   if (!ctx->mol_gid || ctx->nmol_multi == 0) return DDCMI_OK;
   hipStream_t st = ctx->stream;
   const int nm = ctx->nmol_multi;
   hipLaunchKernelGGL(k_mol_split_flag, dim3(cdiv(nm, 256)), dim3(256), 0, st, nm, ctx->mol_info.p, ctx->mol_split.p);
   int rc = ddcmi_scan_exclusive(ctx, ctx->mol_split.p, ctx->mol_split.p, nm, ctx->d_flags + 9);
   if (rc) return rc;
   hipLaunchKernelGGL(k_mol_split_index, dim3(cdiv(nm, 256)), dim3(256), 0, st, nm, ctx->mol_info.p, ctx->mol_split.p);
   HIPCHK(ctx, hipMemcpyAsync(ctx->h_flags + 9, ctx->d_flags + 9, sizeof(int), hipMemcpyDeviceToHost, st));
   HIPCHK(ctx, hipStreamSynchronize(st));
   ctx->nsplit = ctx->h_flags[9];
   ENSURE(ctx, ctx->mol_red, 6 * (size_t)ctx->nsplit + 8);
   return DDCMI_OK;
}
int ddcmi_mol_split_term(ddcmi_ctx *ctx, double out[3])
{
   out[0] = out[1] = out[2] = 0.0;
   if (ctx->nsplit == 0) return DDCMI_OK;
   double *d = ctx->d_results + R_SCR_MOLV + 4;
   HIPCHK(ctx, hipMemsetAsync(d, 0, 3 * sizeof(double), ctx->stream));
   hipLaunchKernelGGL(k_mol_split_term, dim3(cdiv(ctx->nmol_multi, 256)), dim3(256), 0, ctx->stream, ctx->nmol_multi, ctx->mol_split.p, ctx->mol_mtot.p, ctx->mol_red.p, d);
   HIPCHK(ctx, hipMemcpyAsync(out, d, 3 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
   HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
   return DDCMI_OK;
}

extern "C" int ddcmi_set_molecule_lists(ddcmi_ctx *ctx, long nmol_total, int nmulti, const int *mol_off, const int *mol_atoms)
{
   ARGCHK(ctx, nmol_total < 0 || nmulti < 0 || (nmulti > 0 && (!mol_off || !mol_atoms)), "ddcmi_set_molecule_lists: %ld molecules, %d of several beads%s", nmol_total, nmulti, (nmol_total < 0 || nmulti < 0) ? "" : ", an array is NULL");
   if (nmulti > 0) { int rco = check_offsets(ctx, "ddcmi_set_molecule_lists", "mol_off", mol_off, nmulti); if (rco) return rco; }
   int amax = -1;
   for (int k = 0; nmulti > 0 && k < mol_off[nmulti]; k++)
   {
      if (mol_atoms[k] < 0) SETERR(ctx, DDCMI_EINVAL, "ddcmi_set_molecule_lists: entry %d names a negative bead index", k);
      amax = std::max(amax, mol_atoms[k]);
   }
   if (amax >= 0 && ctx->nloc > 0 && amax >= ctx->nloc) SETERR(ctx, DDCMI_EINVAL, "ddcmi_set_molecule_lists: a molecule names bead %d, the uploaded state holds %d", amax, ctx->nloc);
   (void)hipSetDevice(ctx->device);
   int rc;
   ctx->idx_amax_mol = amax;
   ctx->nmol_total = nmol_total; ctx->nmol_multi = 0; ctx->molv_valid = false; ctx->mol_gid = false; ctx->nsplit = 0;
   if (nmulti > 0)
   {
      if ((rc = up(ctx, ctx->mol_off, mol_off, (size_t)nmulti + 1)) || (rc = up(ctx, ctx->mol_atoms, mol_atoms, (size_t)mol_off[nmulti]))) return rc;
      ctx->nmol_multi = nmulti;
   }
   return DDCMI_OK;
}

int ddcmi_launch_mol_virial(ddcmi_ctx *ctx)
{
   HIPCHK(ctx, hipMemsetAsync(ctx->d_results + R_SCR_MOLV, 0, 3 * sizeof(double), ctx->stream));
   if (ctx->nmol_multi == 0) return DDCMI_OK;
   BoxArgs box;
   box.L[0] = ctx->h[0]; box.L[1] = ctx->h[4]; box.L[2] = ctx->h[8];
   for (int a = 0; a < 3; a++) box.Linv[a] = 1.0 / box.L[a];
   box.pbc = ctx->pbc;
   if (ctx->mol_gid)
   {
      if (ctx->nsplit > 0) HIPCHK(ctx, hipMemsetAsync(ctx->mol_red.p, 0, 6 * (size_t)ctx->nsplit * sizeof(double), ctx->stream));
      hipLaunchKernelGGL(k_mol_virial_gid, dim3(cdiv(ctx->nmol_multi, 256)), dim3(256), 0, ctx->stream, ctx->nmol_multi, ctx->nloc, ctx->mol_off.p, ctx->mol_slot.p, ctx->mol_split.p,
                         ctx->mol_info.p, ctx->mol_mtot.p, box, ctx->pos.p, ctx->d_mass.p, ctx->fx.p, ctx->fy.p, ctx->fz.p, ctx->mol_red.p, ctx->d_results + R_SCR_MOLV);
      return DDCMI_OK;
   }
   { int rcs = ddcmi_ensure_slots(ctx); if (rcs) return rcs; }
   hipLaunchKernelGGL(k_mol_virial, dim3(cdiv(ctx->nmol_multi, 256)), dim3(256), 0, ctx->stream, ctx->nmol_multi, ctx->mol_off.p, ctx->mol_atoms.p, ctx->slot_of_orig.p, box,
                      ctx->pos.p, ctx->species.p, ctx->d_mass.p, ctx->fx.p, ctx->fy.p, ctx->fz.p, ctx->d_results + R_SCR_MOLV);
   return DDCMI_OK;
}
