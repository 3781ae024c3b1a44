/*
 * bonded.hip -- Martini bonded terms on the device: one thread per TERM
 * (the reference CUDA path uses one thread per atom looping its terms,
 * bondedGPU.cu:1267-2672).  Formulas follow the CPU reference:
 *   resBondSorted            bioCharmmCovalentEnergiesSorted.c:18-116
 *   resAngleSorted           :118-242   (func 1)
 *   resAngleCosineSorted     :244-363   (func 2)
 *   resAngleRestrainSorted   :365-487   (func 10)
 *   resTorsionSorted         :577-721   (func 1) via bioDihedralFast
 *   resImproperSorted        :723-848   (func 2)   (bioCharmmCovalentEnergies.c:266-351)
 * Term atoms are caller-order indices translated through slot_of_orig each
 * launch (atoms are re-sorted at every rebuild).  Separations use the rint-based
 * nearestImage (Preduce, preduce.c:282-338) like bioVec.  Forces go to the
 * member atoms with FP64 hardware atomics (a handful per atom); energies and
 * virial use per-block partials + a fixed-order second stage.
 */
#include "ddcmi_internal.h"
#include <math.h>

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
/* Where a term's atoms and parameters live.
 *   one domain: atoms[] holds caller-order indices, translated through slot[] each
 *     launch (atoms are re-sorted at every rebuild); term t uses parameter row t.  After a rebuild
 *     the lists are replaced by copies in the order of the first atom's device slot, atoms already
 *     translated (ddcmi_bonded_order): neighbouring threads then touch neighbouring beads.
 *   decomposed run: atoms[] holds device slots (owned or halo) of the terms this rank
 *     touches, rebuilt with the lists (ddcmi_bonded_localize); tmap[t] = parameter row.
 * A rank adds forces only to the atoms it owns (slot < nloc) and counts a term's
 * energy and virial with weight (atoms it owns)/(atoms of the term): every term is then
 * counted exactly once over all ranks, with no force return traffic. */
struct TermMap { const int *atoms; const int *slot; const int *tmap; int nloc; };
__device__ __forceinline__ int term_atom(const TermMap &m, int na, int t, int a)
{
   int i = m.atoms[na * t + a];
   return m.slot ? m.slot[i] : i;
}
__device__ __forceinline__ int term_row(const TermMap &m, int t) { return m.tmap ? m.tmap[t] : t; }
__device__ __forceinline__ void addf(const TermMap &m, double *fx, double *fy, double *fz, int i, double x, double y, double z)
{
   if (i < m.nloc) { atomicAdd(&fx[i], x); atomicAdd(&fy[i], y); atomicAdd(&fz[i], z); }
}
template <int NV>
__device__ __forceinline__ void weigh(double (&acc)[NV], const TermMap &m, int nown, int na)
{
   if (nown == na) return;
   double w = (double)nown / (double)na;
#pragma unroll
   for (int k = 0; k < NV; k++) acc[k] *= w;
}

__global__ __launch_bounds__(256) void k_bond(int nbond, BoxArgs box, TermMap tm, const double *__restrict__ kb_, const double *__restrict__ b0_,
                                              const double4 *__restrict__ pos,
                                              double *fx, double *fy, double *fz, double *partials)
{
   int t = blockIdx.x * blockDim.x + threadIdx.x;
   double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};   /* e, xx,yy,zz,xy,xz,yz */
   if (t < nbond)
   {
      int I = term_atom(tm, 2, t, 0), J = term_atom(tm, 2, t, 1);
      const int g = term_row(tm, t);
      const double kb = kb_[g], b0 = b0_[g];
      double x, y, z;
      bioVec(box, pos[I], pos[J], x, y, z);
      double b = sqrt(x * x + y * y + z * z);
      double bDelta = b - b0;
      acc[0] = kb * bDelta * bDelta;
      double ux = x / b, uy = y / b, uz = z / b;
      double kforce = -2 * kb * bDelta;
      double fxD = kforce * ux, fyD = kforce * uy, fzD = kforce * uz;
      addf(tm, fx, fy, fz, I, fxD, fyD, fzD);
      addf(tm, fx, fy, fz, J, -fxD, -fyD, -fzD);
      acc[1] = fxD * x; acc[2] = fyD * y; acc[3] = fzD * z;
      acc[4] = fxD * y; acc[5] = fxD * z; acc[6] = fyD * z;
      weigh(acc, tm, (I < tm.nloc) + (J < tm.nloc), 2);
   }
   block_store<8>(acc, partials + (size_t)blockIdx.x * 8);
}

__global__ __launch_bounds__(256) void k_angle(int nangle, BoxArgs box, TermMap tm, const int *__restrict__ func,
                                               const double *__restrict__ kt_, const double *__restrict__ t0_, int excl_mask,
                                               const double4 *__restrict__ pos,
                                               double *fx, double *fy, double *fz, double *partials)
{
   int t = blockIdx.x * blockDim.x + threadIdx.x;
   double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
   if (t < nangle)
   {
      const int g = term_row(tm, t);
      int f = func[g];
      bool skip = (f == 1 && (excl_mask & 2)) || (f == 2 && (excl_mask & 4)) || (f == 10 && (excl_mask & 256));
      if (!skip)
      {
         int I = term_atom(tm, 3, t, 0), J = term_atom(tm, 3, t, 1), K = term_atom(tm, 3, t, 2);
         double4 pj = pos[J];
         double ax, ay, az, cx, cy, cz;
         bioVec(box, pos[I], pj, ax, ay, az);
         bioVec(box, pos[K], pj, cx, cy, cz);
         double b_ij = sqrt(ax * ax + ay * ay + az * az), b_kj = sqrt(cx * cx + cy * cy + cz * cz);
         double uix = ax / b_ij, uiy = ay / b_ij, uiz = az / b_ij;
         double ukx = cx / b_kj, uky = cy / b_kj, ukz = cz / b_kj;
         double cosT = uix * ukx + uiy * uky + uiz * ukz;
         double kt = kt_[g], t0 = t0_[g];
         double coef_i, coef_k;
         if (f == 1)
         {
            double a = acos(cosT);
            double aDelta = a - t0;
            acc[0] = kt * aDelta * aDelta;
            double sinabs = sin(a);
            coef_i = 2 * kt * aDelta / (b_ij * sinabs);
            coef_k = 2 * kt * aDelta / (b_kj * sinabs);
         }
         else if (f == 2)
         {
            double aDelta = cosT - t0;
            acc[0] = kt * aDelta * aDelta;
            coef_i = -2 * kt * aDelta / b_ij;
            coef_k = -2 * kt * aDelta / b_kj;
         }
         else
         {
            double sinAsq = 1 - cosT * cosT;
            double aDelta = cosT - t0;
            acc[0] = kt * aDelta * aDelta / sinAsq;
            double coef_reb = -2 * kt * aDelta * (1 - cosT * t0) / (sinAsq * sinAsq);
            coef_i = coef_reb / b_ij;
            coef_k = coef_reb / b_kj;
         }
         double fxI = coef_i * (ukx - uix * cosT), fyI = coef_i * (uky - uiy * cosT), fzI = coef_i * (ukz - uiz * cosT);
         double fxK = coef_k * (uix - ukx * cosT), fyK = coef_k * (uiy - uky * cosT), fzK = coef_k * (uiz - ukz * cosT);
         addf(tm, fx, fy, fz, I, fxI, fyI, fzI);
         addf(tm, fx, fy, fz, K, fxK, fyK, fzK);
         addf(tm, fx, fy, fz, J, -(fxI + fxK), -(fyI + fyK), -(fzI + fzK));
         acc[1] = fxI * ax + fxK * cx; acc[2] = fyI * ay + fyK * cy; acc[3] = fzI * az + fzK * cz;
         acc[4] = fxI * ay + fxK * cy; acc[5] = fxI * az + fxK * cz; acc[6] = fyI * az + fyK * cz;
         weigh(acc, tm, (I < tm.nloc) + (J < tm.nloc) + (K < tm.nloc), 3);
      }
   }
   block_store<8>(acc, partials + (size_t)blockIdx.x * 8);
}

__global__ __launch_bounds__(256) void k_torsion(int ntors, BoxArgs box, TermMap tm, const int *__restrict__ func, const int *__restrict__ nn_,
                                                 const double *__restrict__ kk_, const double *__restrict__ delta_, int excl_mask,
                                                 const double4 *__restrict__ pos,
                                                 double *fx, double *fy, double *fz, double *partials)
{
   int t = blockIdx.x * blockDim.x + threadIdx.x;
   double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};   /* e_tors, e_impr, xx,yy,zz,xy,xz,yz */
   if (t < ntors)
   {
      const int g = term_row(tm, t);
      int f = func[g];
      bool skip = (f == 1 && (excl_mask & 16)) || (f == 2 && (excl_mask & 32));
      if (!skip)
      {
         int I = term_atom(tm, 4, t, 0), J = term_atom(tm, 4, t, 1), K = term_atom(tm, 4, t, 2), L = term_atom(tm, 4, t, 3);
         double4 pI = pos[I], pJ = pos[J], pK = pos[K], pL = pos[L];
         /* bioDihedralFast, bioCharmmCovalentEnergies.c:266-351 */
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
         double yy = 1.0 / sqrt(g1 * g2);
         double xx = yy * ff;
         double xab = yy * bc + xx / g1 * ab;
         double xbc = yy * ab + xx / g2 * bc;
         double xac = -yy * b2;
         double xaa = -0.5 * xx * b2 / g1;
         double xcc = -0.5 * xx * b2 / g2;
         double xbb = -yy * ac - 0.5 * xx * (a2 / g1 + c2 / g2);
         double cax = xab * bx + xac * cx + (2 * xaa) * ax, cay = xab * by + xac * cy + (2 * xaa) * ay, caz = xab * bz + xac * cz + (2 * xaa) * az;
         double cbx = xab * ax + xbc * cx + (2 * xbb) * bx, cby = xab * ay + xbc * cy + (2 * xbb) * by, cbz = xab * az + xbc * cz + (2 * xbb) * bz;
         double ccx = xbc * bx + xac * ax + (2 * xcc) * cx, ccy = xbc * by + xac * ay + (2 * xcc) * cy, ccz = xbc * bz + xac * az + (2 * xcc) * cz;
         double mx = ay * bz - az * by, my = az * bx - ax * bz, mz = ax * by - ay * bx;
         double nx = by * cz - bz * cy, ny = bz * cx - bx * cz, nz = bx * cy - by * cx;
         double qx = my * nz - mz * ny, qy = mz * nx - mx * nz, qz = mx * ny - my * nx;
         double signnum = bx * qx + by * qy + bz * qz;
         double sign = (signnum < 0.0) ? -1.0 : 1.0;
         xx = fmax(fmin(xx, 1.0), -1.0);
         double ang = sign * acos(xx);
         double sinX = sin(ang);
         double v0 = -(cax * ax + cbx * bx + ccx * cx), v3 = -(cax * ay + cbx * by + ccx * cy), v4 = -(cax * az + cbx * bz + ccx * cz);
         double v1 = -(cay * ay + cby * by + ccy * cy), v5 = -(cay * az + cby * bz + ccy * cz), v2 = -(caz * az + cbz * bz + ccz * cz);
         double kk;
         if (f == 1)
         {
            double kchi = kk_[g], delta = delta_[g];
            int n = nn_[g];
            acc[0] = kchi * (1 + cos(n * ang - delta));
            if (fabs(sinX) > FLOAT_EPS) kk = kchi * n * sin(n * ang - delta) / sinX;
            else
            {
               double nX = n * ang, nX2 = nX * nX, nX4 = nX2 * nX2, nX6 = nX4 * nX2, nX8 = nX4 * nX4, nX10 = nX8 * nX2;
               double X2 = ang * ang, X4 = X2 * X2, X6 = X4 * X2, X8 = X4 * X4, X10 = X8 * X2;
               double ratio = n * (1 - nX2 / 6 + nX4 / 120 - nX6 / 5040 + nX8 / 362880 - nX10 / 39916800) /
                              (1 - X2 / 6 + X4 / 120 - X6 / 5040 + X8 / 362880 - X10 / 39916800);
               if (delta < NEAR_ZERO_ANGLE) kk = kchi * n * ratio;
               else if (delta > NEAR_180_ANGLE) kk = -kchi * n * ratio;
               else kk = kchi * n * ratio;
            }
         }
         else
         {
            double kpsi = kk_[g], psi0 = delta_[g];
            double d = ang - psi0;
            if (d < -M_PI) d += 2 * M_PI; else if (d > M_PI) d -= 2 * M_PI;
            acc[1] = kpsi * d * d;
            double absX = sinX < 0 ? -sinX : sinX;
            if (absX > FLOAT_EPS) kk = -2 * kpsi * d / sinX;
            else
            {
               double i2 = ang * ang, i4 = i2 * i2, i6 = i4 * i2, i8 = i4 * i4, i10 = i8 * i2;
               kk = -2 * kpsi / (1 - i2 / 6 + i4 / 120 - i6 / 5040 + i8 / 362880 - i10 / 39916800);
            }
         }
         addf(tm, fx, fy, fz, I, -cax * kk, -cay * kk, -caz * kk);
         addf(tm, fx, fy, fz, J, -(cbx - cax) * kk, -(cby - cay) * kk, -(cbz - caz) * kk);
         addf(tm, fx, fy, fz, K, -(ccx - cbx) * kk, -(ccy - cby) * kk, -(ccz - cbz) * kk);
         addf(tm, fx, fy, fz, L, ccx * kk, ccy * kk, ccz * kk);
         acc[2] = v0 * kk; acc[3] = v1 * kk; acc[4] = v2 * kk; acc[5] = v3 * kk; acc[6] = v4 * kk; acc[7] = v5 * kk;
         weigh(acc, tm, (I < tm.nloc) + (J < tm.nloc) + (K < tm.nloc) + (L < tm.nloc), 4);
      }
   }
   block_store<8>(acc, partials + (size_t)blockIdx.x * 8);
}

/* one launch for the three term kinds: workgroup b sums the partials of kind b in a fixed order */
struct RedB { const double *partials[3]; int nblocks[3]; int nv[3]; double *out[3]; };
__global__ __launch_bounds__(256) void k_reduce_b(RedB rb)
{
   __shared__ double s[256];
   const double *partials = rb.partials[blockIdx.x];
   const int nblocks = rb.nblocks[blockIdx.x], nv = rb.nv[blockIdx.x];
   double *out = rb.out[blockIdx.x];
   if (nblocks <= 0) return;
   for (int k = 0; k < nv; k++)
   {
      double a = 0.0;
      for (int b = threadIdx.x; b < nblocks; b += 256) a += partials[(size_t)b * 8 + k];
      s[threadIdx.x] = a;
      __syncthreads();
      for (int off = 128; off > 0; off >>= 1)
      {
         if (threadIdx.x < off) s[threadIdx.x] += s[threadIdx.x + off];
         __syncthreads();
      }
      if (threadIdx.x == 0) out[k] = s[0];
      __syncthreads();
   }
}

/* ---- evaluation order of the terms of one domain: by the slot of the first atom ---------- */
__global__ void k_tkey_hist(int nterm, int na, const int *__restrict__ atoms, const int *__restrict__ slot, int *key, int *cnt)
{
   int t = blockIdx.x * blockDim.x + threadIdx.x;
   if (t >= nterm) return;
   int k = slot[atoms[na * t]];
   key[t] = k;
   atomicAdd(&cnt[k], 1);
}
__global__ void k_tkey_place(int nterm, const int *__restrict__ key, const int *__restrict__ start, int *fill, int *perm)
{
   int t = blockIdx.x * blockDim.x + threadIdx.x;
   if (t >= nterm) return;
   int k = key[t];
   perm[start[k] + atomicAdd(&fill[k], 1)] = t;
}
/* terms sharing a first atom: ascending term index, so the order does not depend on timing */
__global__ void k_tkey_fix(int nkey, const int *__restrict__ start, int *perm)
{
   int k = blockIdx.x * blockDim.x + threadIdx.x;
   if (k >= nkey) return;
   int a = start[k], b = start[k + 1];
   for (int i = a + 1; i < b; i++)
   {
      int v = perm[i], j = i - 1;
      while (j >= a && perm[j] > v) { perm[j + 1] = perm[j]; j--; }
      perm[j + 1] = v;
   }
}
static int sort_terms(ddcmi_ctx *ctx, int nterm, int na, const int *atoms, dbuf<int> &perm)
{
   if (nterm <= 0) return DDCMI_OK;
   hipStream_t st = ctx->stream;
   const int nkey = ctx->nloc;
   ENSURE(ctx, ctx->tk_key, (size_t)nterm);
   ENSURE(ctx, ctx->tk_cnt, (size_t)nkey + 2);
   ENSURE(ctx, ctx->tk_fill, (size_t)nkey + 2);
   ENSURE(ctx, perm, (size_t)nterm);
   HIPCHK(ctx, hipMemsetAsync(ctx->tk_cnt.p, 0, ((size_t)nkey + 2) * sizeof(int), st));
   HIPCHK(ctx, hipMemsetAsync(ctx->tk_fill.p, 0, ((size_t)nkey + 2) * sizeof(int), st));
   hipLaunchKernelGGL(k_tkey_hist, dim3(cdiv(nterm, 256)), dim3(256), 0, st, nterm, na, atoms, ctx->slot_of_orig.p, ctx->tk_key.p, ctx->tk_cnt.p);
   int rc = ddcmi_scan_exclusive(ctx, ctx->tk_cnt.p, nkey + 1, ctx->d_flags + 8);
   if (rc) return rc;
   hipLaunchKernelGGL(k_tkey_place, dim3(cdiv(nterm, 256)), dim3(256), 0, st, nterm, ctx->tk_key.p, ctx->tk_cnt.p, ctx->tk_fill.p, perm.p);
   hipLaunchKernelGGL(k_tkey_fix, dim3(cdiv(nkey, 256)), dim3(256), 0, st, nkey, ctx->tk_cnt.p, perm.p);
   return DDCMI_OK;
}
/* ... and the terms' atoms (as device slots) and parameters copied into that order, so the
 * kernels read everything but the bead records and forces with unit stride until the next rebuild */
template <int NA>
__global__ void k_term_slots(int nterm, const int *__restrict__ perm, const int *__restrict__ atoms, const int *__restrict__ slot, int *out)
{
   int t = blockIdx.x * blockDim.x + threadIdx.x;
   if (t >= nterm) return;
   int g = perm[t];
#pragma unroll
   for (int a = 0; a < NA; a++) out[NA * t + a] = slot[atoms[NA * g + a]];
}
template <class T>
__global__ void k_gather_perm(int n, const int *__restrict__ perm, const T *__restrict__ src, T *dst)
{
   int t = blockIdx.x * blockDim.x + threadIdx.x;
   if (t < n) dst[t] = src[perm[t]];
}
template <class T>
static int gather_perm(ddcmi_ctx *ctx, int n, const dbuf<int> &perm, const dbuf<T> &src, dbuf<T> &dst)
{
   ENSURE(ctx, dst, (size_t)n);
   hipLaunchKernelGGL(k_gather_perm<T>, dim3(cdiv(n, 256)), dim3(256), 0, ctx->stream, n, perm.p, src.p, dst.p);
   return DDCMI_OK;
}
/* one domain, at every rebuild: evaluation order of the caller-order term lists */
int ddcmi_bonded_order(ddcmi_ctx *ctx)
{
   if (ctx->bonded_gid || ctx->nranks > 1 || ctx->group_) return DDCMI_OK;
   hipStream_t st = ctx->stream;
   int rc;
   if ((rc = sort_terms(ctx, ctx->nbond, 2, ctx->bond_ij.p, ctx->o_bond)) || (rc = sort_terms(ctx, ctx->nangle, 3, ctx->angle_ijk.p, ctx->o_angle)) ||
       (rc = sort_terms(ctx, ctx->ntors, 4, ctx->tors_ijkl.p, ctx->o_tors))) return rc;
   if (ctx->nbond > 0)
   {
      ENSURE(ctx, ctx->s_bond_atoms, 2 * (size_t)ctx->nbond);
      hipLaunchKernelGGL(k_term_slots<2>, dim3(cdiv(ctx->nbond, 256)), dim3(256), 0, st, ctx->nbond, ctx->o_bond.p, ctx->bond_ij.p, ctx->slot_of_orig.p, ctx->s_bond_atoms.p);
      if ((rc = gather_perm(ctx, ctx->nbond, ctx->o_bond, ctx->bond_kb, ctx->s_bond_kb)) || (rc = gather_perm(ctx, ctx->nbond, ctx->o_bond, ctx->bond_b0, ctx->s_bond_b0))) return rc;
   }
   if (ctx->nangle > 0)
   {
      ENSURE(ctx, ctx->s_angle_atoms, 3 * (size_t)ctx->nangle);
      hipLaunchKernelGGL(k_term_slots<3>, dim3(cdiv(ctx->nangle, 256)), dim3(256), 0, st, ctx->nangle, ctx->o_angle.p, ctx->angle_ijk.p, ctx->slot_of_orig.p, ctx->s_angle_atoms.p);
      if ((rc = gather_perm(ctx, ctx->nangle, ctx->o_angle, ctx->angle_func, ctx->s_angle_func)) || (rc = gather_perm(ctx, ctx->nangle, ctx->o_angle, ctx->angle_k, ctx->s_angle_k)) ||
          (rc = gather_perm(ctx, ctx->nangle, ctx->o_angle, ctx->angle_t0, ctx->s_angle_t0))) return rc;
   }
   if (ctx->ntors > 0)
   {
      ENSURE(ctx, ctx->s_tors_atoms, 4 * (size_t)ctx->ntors);
      hipLaunchKernelGGL(k_term_slots<4>, dim3(cdiv(ctx->ntors, 256)), dim3(256), 0, st, ctx->ntors, ctx->o_tors.p, ctx->tors_ijkl.p, ctx->slot_of_orig.p, ctx->s_tors_atoms.p);
      if ((rc = gather_perm(ctx, ctx->ntors, ctx->o_tors, ctx->tors_func, ctx->s_tors_func)) || (rc = gather_perm(ctx, ctx->ntors, ctx->o_tors, ctx->tors_n, ctx->s_tors_n)) ||
          (rc = gather_perm(ctx, ctx->ntors, ctx->o_tors, ctx->tors_k, ctx->s_tors_k)) || (rc = gather_perm(ctx, ctx->ntors, ctx->o_tors, ctx->tors_delta, ctx->s_tors_delta))) return rc;
   }
   ctx->bonded_ordered = true;
   return DDCMI_OK;
}

/* ---- decomposed runs: terms are given by gid and located among the owned + halo beads - */
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
/* pass 1: which of the global terms touch an owned bead; their slots.  flags[0] counts
 * terms with an owned atom whose partner is neither owned nor in the halo. */
template <int NA>
__global__ void k_term_locate(int nterm, const uint64_t *__restrict__ tgid, int nloc, unsigned mask, const unsigned long long *keys, const int *vals,
                              int *sel, int *slots, int *flags)
{
   int t = blockIdx.x * blockDim.x + threadIdx.x;
   if (t >= nterm) return;
   int s[NA], nown = 0, nmiss = 0;
#pragma unroll
   for (int a = 0; a < NA; a++)
   {
      s[a] = gid_find(tgid[(size_t)NA * t + a], mask, keys, vals);
      nown += (s[a] >= 0 && s[a] < nloc);
      nmiss += (s[a] < 0);
   }
   int take = nown > 0;
   if (take && nmiss) { atomicAdd(&flags[0], 1); take = 0; }
   sel[t] = take;
#pragma unroll
   for (int a = 0; a < NA; a++) slots[(size_t)NA * t + a] = s[a];
}
/* pass 2: stable compaction (sel has been turned into an exclusive scan) */
template <int NA>
__global__ void k_term_compact(int nterm, const int *__restrict__ pre, const int *__restrict__ slots, int total, int *tmap, int *latoms)
{
   int t = blockIdx.x * blockDim.x + threadIdx.x;
   if (t >= nterm) return;
   int o = pre[t], nxt = (t + 1 < nterm) ? pre[t + 1] : total;
   if (nxt == o) return;
   tmap[o] = t;
#pragma unroll
   for (int a = 0; a < NA; a++) latoms[(size_t)NA * o + a] = slots[(size_t)NA * t + a];
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

extern "C" int ddcmi_set_bonded(ddcmi_ctx *ctx,
                                int nbond, const int *bond_ij, const double *bond_kb, const double *bond_b0,
                                int nangle, const int *angle_ijk, const int *angle_func, const double *angle_k, const double *angle_t0,
                                int ntors, const int *tors_ijkl, const int *tors_func, const int *tors_n, const double *tors_k, const double *tors_delta,
                                int excludePotentialTerm)
{
   if (!ctx || nbond < 0 || nangle < 0 || ntors < 0) return DDCMI_EINVAL;
   (void)hipSetDevice(ctx->device);
   ctx->excludePotentialTerm = excludePotentialTerm;
   ctx->bonded_gid = false; ctx->bonded_ordered = false; ctx->list_valid = false;
   ctx->nbond = (excludePotentialTerm & 1) ? 0 : nbond;
   ctx->nangle = nangle; ctx->ntors = ntors;
   int rc;
   if (ctx->nbond > 0)
   {
      if (!bond_ij || !bond_kb || !bond_b0) return DDCMI_EINVAL;
      if ((rc = up(ctx, ctx->bond_ij, bond_ij, 2 * (size_t)nbond)) || (rc = up(ctx, ctx->bond_kb, bond_kb, nbond)) || (rc = up(ctx, ctx->bond_b0, bond_b0, nbond))) return rc;
   }
   if (nangle > 0)
   {
      if (!angle_ijk || !angle_func || !angle_k || !angle_t0) return DDCMI_EINVAL;
      for (int t = 0; t < nangle; t++)
         if (angle_func[t] != 1 && angle_func[t] != 2 && angle_func[t] != 10) SETERR(ctx, DDCMI_EINVAL, "angle %d: func %d is not 1, 2 or 10", t, angle_func[t]);
      if ((rc = up(ctx, ctx->angle_ijk, angle_ijk, 3 * (size_t)nangle)) || (rc = up(ctx, ctx->angle_func, angle_func, nangle)) ||
          (rc = up(ctx, ctx->angle_k, angle_k, nangle)) || (rc = up(ctx, ctx->angle_t0, angle_t0, nangle))) return rc;
   }
   if (ntors > 0)
   {
      if (!tors_ijkl || !tors_func || !tors_n || !tors_k || !tors_delta) return DDCMI_EINVAL;
      for (int t = 0; t < ntors; t++)
         if (tors_func[t] != 1 && tors_func[t] != 2) SETERR(ctx, DDCMI_EINVAL, "dihedral %d: func %d is not 1 or 2", t, tors_func[t]);
      if ((rc = up(ctx, ctx->tors_ijkl, tors_ijkl, 4 * (size_t)ntors)) || (rc = up(ctx, ctx->tors_func, tors_func, ntors)) || (rc = up(ctx, ctx->tors_n, tors_n, ntors)) ||
          (rc = up(ctx, ctx->tors_k, tors_k, ntors)) || (rc = up(ctx, ctx->tors_delta, tors_delta, ntors))) return rc;
   }
   /* the per-kind sums are only written by kernels that run: clear stale ones */
   HIPCHK(ctx, hipMemsetAsync(ctx->d_results + R_SCR_BOND, 0, (R_RK - R_SCR_BOND) * sizeof(double), ctx->stream));
   HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
   ctx->forces_valid = false;
   return DDCMI_OK;
}

extern "C" int ddcmi_set_bonded_gid(ddcmi_ctx *ctx,
                                    int nbond, const uint64_t *bond_gid, const double *bond_kb, const double *bond_b0,
                                    int nangle, const uint64_t *angle_gid, const int *angle_func, const double *angle_k, const double *angle_t0,
                                    int ntors, const uint64_t *tors_gid, const int *tors_func, const int *tors_n, const double *tors_k, const double *tors_delta,
                                    int excludePotentialTerm)
{
   if (!ctx || nbond < 0 || nangle < 0 || ntors < 0) return DDCMI_EINVAL;
   (void)hipSetDevice(ctx->device);
   ctx->excludePotentialTerm = excludePotentialTerm;
   ctx->bonded_gid = true;
   ctx->g_nbond = (excludePotentialTerm & 1) ? 0 : nbond;
   ctx->g_nangle = nangle; ctx->g_ntors = ntors;
   ctx->nbond = ctx->nangle = ctx->ntors = 0;           /* local counts: set by ddcmi_bonded_localize at every rebuild */
   int rc;
   if (ctx->g_nbond > 0)
   {
      if (!bond_gid || !bond_kb || !bond_b0) return DDCMI_EINVAL;
      if ((rc = up(ctx, ctx->gbond_gid, bond_gid, 2 * (size_t)nbond)) || (rc = up(ctx, ctx->bond_kb, bond_kb, nbond)) || (rc = up(ctx, ctx->bond_b0, bond_b0, nbond))) return rc;
   }
   if (nangle > 0)
   {
      if (!angle_gid || !angle_func || !angle_k || !angle_t0) return DDCMI_EINVAL;
      for (int t = 0; t < nangle; t++)
         if (angle_func[t] != 1 && angle_func[t] != 2 && angle_func[t] != 10) SETERR(ctx, DDCMI_EINVAL, "angle %d: func %d is not 1, 2 or 10", t, angle_func[t]);
      if ((rc = up(ctx, ctx->gangle_gid, angle_gid, 3 * (size_t)nangle)) || (rc = up(ctx, ctx->angle_func, angle_func, nangle)) ||
          (rc = up(ctx, ctx->angle_k, angle_k, nangle)) || (rc = up(ctx, ctx->angle_t0, angle_t0, nangle))) return rc;
   }
   if (ntors > 0)
   {
      if (!tors_gid || !tors_func || !tors_n || !tors_k || !tors_delta) return DDCMI_EINVAL;
      for (int t = 0; t < ntors; t++)
         if (tors_func[t] != 1 && tors_func[t] != 2) SETERR(ctx, DDCMI_EINVAL, "dihedral %d: func %d is not 1 or 2", t, tors_func[t]);
      if ((rc = up(ctx, ctx->gtors_gid, tors_gid, 4 * (size_t)ntors)) || (rc = up(ctx, ctx->tors_func, tors_func, ntors)) || (rc = up(ctx, ctx->tors_n, tors_n, ntors)) ||
          (rc = up(ctx, ctx->tors_k, tors_k, ntors)) || (rc = up(ctx, ctx->tors_delta, tors_delta, ntors))) return rc;
   }
   HIPCHK(ctx, hipMemsetAsync(ctx->d_results + R_SCR_BOND, 0, (R_RK - R_SCR_BOND) * sizeof(double), ctx->stream));
   HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
   ctx->forces_valid = false;
   ctx->list_valid = false;               /* the local term lists are made with the neighbour list */
   return DDCMI_OK;
}

template <int NA>
static int localize_kind(ddcmi_ctx *ctx, int nterm, const uint64_t *tgid, dbuf<int> &tmap, dbuf<int> &latoms, int *nlocal, int *d_total)
{
   *nlocal = 0;
   if (nterm <= 0) return DDCMI_OK;
   hipStream_t st = ctx->stream;
   ENSURE(ctx, ctx->term_sel, (size_t)nterm + 1);
   ENSURE(ctx, ctx->term_slots, (size_t)NA * nterm);
   hipLaunchKernelGGL(k_term_locate<NA>, dim3(cdiv(nterm, 256)), dim3(256), 0, st, nterm, tgid, ctx->nloc, ctx->hmask, ctx->hkeys.p, ctx->hvals.p,
                      ctx->term_sel.p, ctx->term_slots.p, ctx->d_flags);
   int rc = ddcmi_scan_exclusive(ctx, ctx->term_sel.p, nterm, d_total);
   if (rc) return rc;
   int total = 0;
   HIPCHK(ctx, hipMemcpyAsync(&total, d_total, sizeof(int), hipMemcpyDeviceToHost, st));
   HIPCHK(ctx, hipStreamSynchronize(st));
   if (total > 0)
   {
      ENSURE(ctx, tmap, (size_t)total);
      ENSURE(ctx, latoms, (size_t)NA * total);
      hipLaunchKernelGGL(k_term_compact<NA>, dim3(cdiv(nterm, 256)), dim3(256), 0, st, nterm, ctx->term_sel.p, ctx->term_slots.p, total, tmap.p, latoms.p);
   }
   *nlocal = total;
   return DDCMI_OK;
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
                                                   double *fx, double *fy, double *fz, double *out)
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
      atomicAdd(&fx[i], f0); atomicAdd(&fy[i], f1); atomicAdd(&fz[i], f2);
      acc[0] += kb * (c[0] * d[0] + c[1] * d[1] + c[2] * d[2]);
      acc[1] += f0 * c[0]; acc[2] += f1 * c[1]; acc[3] += f2 * c[2];
      acc[4] += f0 * c[1]; acc[5] += f0 * c[2]; acc[6] += f1 * c[2];
   }
   block_store<8>(acc, out);
}
extern "C" int ddcmi_set_restraints(ddcmi_ctx *ctx, int n, const uint64_t *gid, const int *fc, const double *r0, const double *kb, int origin)
{
   if (!ctx || n < 0 || (n > 0 && (!gid || !fc || !r0 || !kb))) return DDCMI_EINVAL;
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

/* decomposed runs, at every list rebuild: find the terms that touch an owned bead and
 * the device slots (owned or halo) of their atoms */
int ddcmi_bonded_localize(ddcmi_ctx *ctx)
{
   if (!ctx->bonded_gid && ctx->nrest == 0) return DDCMI_OK;
   if (ctx->bonded_gid && ctx->g_nbond + ctx->g_nangle + ctx->g_ntors == 0 && ctx->nrest == 0)
   { ctx->nbond = ctx->nangle = ctx->ntors = 0; return DDCMI_OK; }      /* water: nothing to locate, no gid table */
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
   int rc;
   int *d_total = ctx->d_flags + 8;
   if (ctx->nrest > 0)
      hipLaunchKernelGGL(k_rest_locate, dim3(cdiv(ctx->nrest, 256)), dim3(256), 0, st, ctx->nrest, ctx->rest_gid.p, ctx->nloc, ctx->hmask, ctx->hkeys.p, ctx->hvals.p, ctx->rest_slot.p);
   if (!ctx->bonded_gid) { HIPCHK(ctx, hipStreamSynchronize(st)); return DDCMI_OK; }
   if ((rc = localize_kind<2>(ctx, ctx->g_nbond, ctx->gbond_gid.p, ctx->l_bond_map, ctx->l_bond_atoms, &ctx->nbond, d_total))) return rc;
   if ((rc = localize_kind<3>(ctx, ctx->g_nangle, ctx->gangle_gid.p, ctx->l_angle_map, ctx->l_angle_atoms, &ctx->nangle, d_total))) return rc;
   if ((rc = localize_kind<4>(ctx, ctx->g_ntors, ctx->gtors_gid.p, ctx->l_tors_map, ctx->l_tors_atoms, &ctx->ntors, d_total))) return rc;
   HIPCHK(ctx, hipMemcpyAsync(ctx->h_flags, ctx->d_flags, sizeof(int), hipMemcpyDeviceToHost, st));
   HIPCHK(ctx, hipMemsetAsync(ctx->d_results + R_SCR_BOND, 0, (R_RK - R_SCR_BOND) * sizeof(double), st));
   HIPCHK(ctx, hipStreamSynchronize(st));
   if (ctx->h_flags[0] > 0)
      SETERR(ctx, DDCMI_EUNSUPPORTED, "%d bonded terms reach beyond the halo (rmax+deltaR=%g): a partner of an owned bead is on no neighbouring domain's send list",
             ctx->h_flags[0], ctx->rmax + ctx->deltaR);
   return DDCMI_OK;
}

int ddcmi_launch_bonded(ddcmi_ctx *ctx)
{
   if (ctx->nbond + ctx->nangle + ctx->ntors + ctx->nrest == 0) return DDCMI_OK;
   hipStream_t st = ctx->stream;
   BoxArgs box;
   box.L[0] = ctx->h[0]; box.L[1] = ctx->h[4]; box.L[2] = ctx->h[8];
   for (int a = 0; a < 3; a++) box.Linv[a] = 1.0 / box.L[a];
   box.pbc = ctx->pbc;
   if (ctx->nrest > 0)
      hipLaunchKernelGGL(k_restraint, dim3(1), dim3(256), 0, st, ctx->nrest, box, ctx->rest_origin, ctx->rest_slot.p, ctx->rest_fc.p, ctx->rest_r0.p, ctx->rest_kb.p,
                         ctx->pos.p, ctx->fx.p, ctx->fy.p, ctx->fz.p, ctx->d_results + R_SCR_REST);
   if (ctx->nbond + ctx->nangle + ctx->ntors == 0) return DDCMI_OK;
   const bool gidmode = ctx->bonded_gid;
   int nbb = cdiv(ctx->nbond, 256), nab = cdiv(ctx->nangle, 256), ntb = cdiv(ctx->ntors, 256);
   ENSURE(ctx, ctx->bpartials, (size_t)(nbb + nab + ntb + 3) * 8);
   double *pb = ctx->bpartials.p, *pa = pb + (size_t)nbb * 8, *pt = pa + (size_t)nab * 8;
   if (ctx->nbond > 0)
   {
      const bool ord = !gidmode && ctx->bonded_ordered;
      TermMap tm = gidmode ? TermMap{ctx->l_bond_atoms.p, nullptr, ctx->l_bond_map.p, ctx->nloc}
                 : ord ? TermMap{ctx->s_bond_atoms.p, nullptr, nullptr, ctx->nloc} : TermMap{ctx->bond_ij.p, ctx->slot_of_orig.p, nullptr, ctx->nloc};
      hipLaunchKernelGGL(k_bond, dim3(nbb), dim3(256), 0, st, ctx->nbond, box, tm, ord ? ctx->s_bond_kb.p : ctx->bond_kb.p, ord ? ctx->s_bond_b0.p : ctx->bond_b0.p, ctx->pos.p,
                         ctx->fx.p, ctx->fy.p, ctx->fz.p, pb);
   }
   if (ctx->nangle > 0)
   {
      const bool ord = !gidmode && ctx->bonded_ordered;
      TermMap tm = gidmode ? TermMap{ctx->l_angle_atoms.p, nullptr, ctx->l_angle_map.p, ctx->nloc}
                 : ord ? TermMap{ctx->s_angle_atoms.p, nullptr, nullptr, ctx->nloc} : TermMap{ctx->angle_ijk.p, ctx->slot_of_orig.p, nullptr, ctx->nloc};
      hipLaunchKernelGGL(k_angle, dim3(nab), dim3(256), 0, st, ctx->nangle, box, tm, ord ? ctx->s_angle_func.p : ctx->angle_func.p, ord ? ctx->s_angle_k.p : ctx->angle_k.p, ord ? ctx->s_angle_t0.p : ctx->angle_t0.p,
                         ctx->excludePotentialTerm, ctx->pos.p, ctx->fx.p, ctx->fy.p, ctx->fz.p, pa);
   }
   if (ctx->ntors > 0)
   {
      const bool ord = !gidmode && ctx->bonded_ordered;
      TermMap tm = gidmode ? TermMap{ctx->l_tors_atoms.p, nullptr, ctx->l_tors_map.p, ctx->nloc}
                 : ord ? TermMap{ctx->s_tors_atoms.p, nullptr, nullptr, ctx->nloc} : TermMap{ctx->tors_ijkl.p, ctx->slot_of_orig.p, nullptr, ctx->nloc};
      hipLaunchKernelGGL(k_torsion, dim3(ntb), dim3(256), 0, st, ctx->ntors, box, tm, ord ? ctx->s_tors_func.p : ctx->tors_func.p, ord ? ctx->s_tors_n.p : ctx->tors_n.p,
                         ord ? ctx->s_tors_k.p : ctx->tors_k.p, ord ? ctx->s_tors_delta.p : ctx->tors_delta.p,
                         ctx->excludePotentialTerm, ctx->pos.p, ctx->fx.p, ctx->fy.p, ctx->fz.p, pt);
   }
   RedB rb;
   rb.partials[0] = pb; rb.nblocks[0] = (ctx->nbond > 0) ? nbb : 0; rb.nv[0] = 7; rb.out[0] = ctx->d_results + R_SCR_BOND;
   rb.partials[1] = pa; rb.nblocks[1] = (ctx->nangle > 0) ? nab : 0; rb.nv[1] = 7; rb.out[1] = ctx->d_results + R_SCR_ANGLE;
   rb.partials[2] = pt; rb.nblocks[2] = (ctx->ntors > 0) ? ntb : 0; rb.nv[2] = 8; rb.out[2] = ctx->d_results + R_SCR_TORS;
   hipLaunchKernelGGL(k_reduce_b, dim3(3), dim3(256), 0, st, rb);
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
                                                      const double *__restrict__ invmass, double *vx, double *vy, double *vz, double dt, int maxA, int maxP, int *status)
{
   extern __shared__ double cons_sh[];
   const int t = threadIdx.x, g = blockIdx.x * CONS_T + t;
   double *v = cons_sh, *rm = cons_sh + 3 * maxA * CONS_T, *rab = rm + maxA * CONS_T;
   if (g >= ngroups) return;
   const int a0 = atom_off[g], na = atom_off[g + 1] - a0, p0 = pair_off[g], np = pair_off[g + 1] - p0;
   for (int a = 0; a < na; a++)
   {
      int s = slot[atoms[a0 + a]];
      v[(3 * a + 0) * CONS_T + t] = vx[s]; v[(3 * a + 1) * CONS_T + t] = vy[s]; v[(3 * a + 2) * CONS_T + t] = vz[s];
      rm[a * CONS_T + t] = invmass[species[s]];
   }
   for (int ab = 0; ab < np; ab++)
   {
      int sa = slot[atoms[a0 + pa[p0 + ab]]], sb = slot[atoms[a0 + pb[p0 + ab]]];
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
      int s = slot[atoms[a0 + a]];
      vx[s] = v[(3 * a + 0) * CONS_T + t]; vy[s] = v[(3 * a + 1) * CONS_T + t]; vz[s] = v[(3 * a + 2) * CONS_T + t];
   }
   atomicMax(status, it < maxit ? it + 1 : maxit);
   if (it == maxit) atomicAdd(status + 1, 1);
}

extern "C" int ddcmi_set_constraints(ddcmi_ctx *ctx, int ngroups, const int *pair_off, const int *pairI, const int *pairJ, const double *dist)
{
   if (!ctx || ngroups < 0 || (ngroups > 0 && (!pair_off || !pairI || !pairJ || !dist))) return DDCMI_EINVAL;
   if (ngroups > 0 && (ctx->group_ || ctx->nranks > 1)) SETERR(ctx, DDCMI_EINVAL, "velocity constraints are implemented for one domain only");
   (void)hipSetDevice(ctx->device);
   ctx->ncgroup = 0; ctx->ncpair = 0;
   if (ngroups == 0) return DDCMI_OK;
   /* group-local atom lists (CONSTRAINT.atomIDList, bioMartini.c:405-425): pairs name positions in them */
   std::vector<int> aoff(ngroups + 1, 0), alist;
   const int np_tot = pair_off[ngroups];
   std::vector<unsigned char> pa(np_tot), pb(np_tot);
   int maxA = 0, maxP = 0;
   for (int g = 0; g < ngroups; g++)
   {
      const int base = (int)alist.size();
      for (int k = pair_off[g]; k < pair_off[g + 1]; k++)
      {
         if (pairI[k] == pairJ[k] || !(dist[k] > 0.0)) SETERR(ctx, DDCMI_EINVAL, "constraint %d of group %d: atoms %d %d, distance %g", k - pair_off[g], g, pairI[k], pairJ[k], dist[k]);
         int loc[2];
         for (int e = 0; e < 2; e++)
         {
            const int at = e ? pairJ[k] : pairI[k];
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
   if ((rc = up(ctx, ctx->cg_atom_off, aoff.data(), (size_t)ngroups + 1)) || (rc = up(ctx, ctx->cg_atoms, alist.data(), alist.size())) ||
       (rc = up(ctx, ctx->cg_pair_off, pair_off, (size_t)ngroups + 1)) || (rc = up(ctx, ctx->cg_pa, pa.data(), (size_t)np_tot)) ||
       (rc = up(ctx, ctx->cg_pb, pb.data(), (size_t)np_tot)) || (rc = up(ctx, ctx->cg_dist, dist, (size_t)np_tot))) return rc;
   ENSURE(ctx, ctx->cons_status, 4);
   HIPCHK(ctx, hipMemsetAsync(ctx->cons_status.p, 0, 4 * sizeof(int), ctx->stream));
   HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
   ctx->ncgroup = ngroups; ctx->ncpair = np_tot; ctx->cons_maxA = maxA; ctx->cons_maxP = maxP;
   return DDCMI_OK;
}

int ddcmi_launch_constraints(ddcmi_ctx *ctx, double dt, int location)
{
   if (ctx->ncgroup == 0) return DDCMI_OK;
   BoxArgs box;
   box.L[0] = ctx->h[0]; box.L[1] = ctx->h[4]; box.L[2] = ctx->h[8];
   for (int a = 0; a < 3; a++) box.Linv[a] = 1.0 / box.L[a];
   box.pbc = ctx->pbc;
   const size_t lds = (size_t)(4 * ctx->cons_maxA + 3 * ctx->cons_maxP) * CONS_T * sizeof(double);
   auto kern = location == 0 ? k_constrain<0> : k_constrain<1>;
   hipLaunchKernelGGL(kern, dim3(cdiv(ctx->ncgroup, CONS_T)), dim3(CONS_T), lds, ctx->stream, ctx->ncgroup, ctx->cg_atom_off.p, ctx->cg_atoms.p, ctx->cg_pair_off.p,
                      ctx->cg_pa.p, ctx->cg_pb.p, ctx->cg_dist.p, ctx->slot_of_orig.p, box, ctx->pos.p, ctx->species.p, ctx->d_invmass.p,
                      ctx->vx.p, ctx->vy.p, ctx->vz.p, dt, ctx->cons_maxA, ctx->cons_maxP, ctx->cons_status.p);
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

extern "C" int ddcmi_set_molecule_lists(ddcmi_ctx *ctx, long nmol_total, int nmulti, const int *mol_off, const int *mol_atoms)
{
   if (!ctx || nmol_total < 0 || nmulti < 0 || (nmulti > 0 && (!mol_off || !mol_atoms))) return DDCMI_EINVAL;
   (void)hipSetDevice(ctx->device);
   int rc;
   ctx->nmol_total = nmol_total; ctx->nmol_multi = 0; ctx->molv_valid = false;
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
   hipLaunchKernelGGL(k_mol_virial, dim3(cdiv(ctx->nmol_multi, 256)), dim3(256), 0, ctx->stream, ctx->nmol_multi, ctx->mol_off.p, ctx->mol_atoms.p, ctx->slot_of_orig.p, box,
                      ctx->pos.p, ctx->species.p, ctx->d_mass.p, ctx->fx.p, ctx->fy.p, ctx->fz.p, ctx->d_results + R_SCR_MOLV);
   return DDCMI_OK;
}
