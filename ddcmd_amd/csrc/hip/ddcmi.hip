/*
 * ddcmi.hip -- MI355X (gfx950 / CDNA4) device path of ddcMD's Martini MD inner
 * loop, written from scratch for wave64.  This file holds the context, the
 * bin-sort / image-atom / neighbour-list build, the nonbonded kernel, the NGLF
 * integrator kernels and the reductions.  C-ABI: include/ddcmi.h.
 *
 * Design (DESIGN.md has the full story):
 *  - owned atoms are kept cell-sorted (tile-major cells) in one 32-byte record
 *    {x,y,z,(ljtype|species)} so a neighbour gather is two 16-byte loads from one
 *    64-byte line, and 256 consecutive atoms are a compact blob whose neighbours
 *    live in one XCD's L2 (the nonbonded launch maps contiguous tile ranges to XCDs);
 *  - periodic boundaries are handled by image atoms appended after the owned
 *    ones (the same slots hold RCCL halo atoms in multi-GPU runs), refreshed each
 *    step, so the inner loop has no minimum-image arithmetic;
 *  - a FULL neighbour list (ELL, slot-major => coalesced) is rebuilt every
 *    updateRate steps at rmax+deltaR; each owned atom accumulates its own force,
 *    no atomics, no force return message; energies and virial count 1/2 per visit;
 *  - all sums are FP64 and use fixed-order two-stage reductions (bitwise
 *    reproducible run to run).
 *
 * Reference semantics followed: martiniNonBond / martiniIntraMoleReaction /
 * reOrgPairs (bioMartini.c:989-1208,1392-1485), pairlist1 (pairlist.c:205-314),
 * nglf (nglf.c:67-112), free/berendsen kicks (free.c:13-28, berendsen.c:30-89),
 * kinetic_terms (energy.c:48-163).
 */
#include "ddcmi_internal.h"
#include <math.h>
#include <algorithm>

static std::string g_create_err;

/* ------------------------------------------------------------------------- */
/* small device helpers                                                       */
__device__ __forceinline__ int cell_linear(const GridParams &gp, int cx, int cy, int cz)
{
   int tx = cx >> 2, ty = cy >> 2, tz = cz >> 2;
   return ((((tz * gp.T[1] + ty) * gp.T[0]) + tx) << 6) | ((cz & 3) << 4) | ((cy & 3) << 2) | (cx & 3);
}
__device__ __forceinline__ void cell_coords(const GridParams &gp, double x, double y, double z, bool owned, int &cx, int &cy, int &cz)
{
   double r[3] = {x, y, z};
   int c[3];
#pragma unroll
   for (int a = 0; a < 3; a++)
   {
      int ic = (int)floor((r[a] - gp.lo[a]) * gp.cinv[a]);
      if (owned) ic = min(max(ic, 0), gp.n[a] - 1);
      ic += gp.m[a];
      c[a] = min(max(ic, 0), gp.g[a] - 1);
   }
   cx = c[0]; cy = c[1]; cz = c[2];
}
__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
   for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
   return v;
}
/* block (256 threads) reduction of NV values per thread into out[NV] by thread 0 */
template <int NV>
__device__ __forceinline__ void block_reduce_store(double (&v)[NV], double *out)
{
   __shared__ double s_red[4][NV];
   int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
   for (int k = 0; k < NV; k++)
   {
      double s = wave_sum(v[k]);
      if (lane == 0) s_red[w][k] = s;
   }
   __syncthreads();
   if (threadIdx.x < NV)
   {
      int k = threadIdx.x;
      out[k] = ((s_red[0][k] + s_red[1][k]) + (s_red[2][k] + s_red[3][k]));
   }
}

/* ------------------------------------------------------------------------- */
/* sort: wrap + cell id + in-cell rank                                        */
__global__ void k_wrap_cell(GridParams gp, int nloc, double4 *pos, int *cid, int *rank, int *cell_cnt)
{
   int i = blockIdx.x * blockDim.x + threadIdx.x;
   if (i >= nloc) return;
   double4 p = pos[i];
   /* backInBox_fast: PreduceOrthorhombicB7_OneLatticeReduction (preduce.c:147-160) */
   if (gp.pbc & 1) { if (p.x > 0.5 * gp.L[0]) p.x -= gp.L[0]; if (p.x < -0.5 * gp.L[0]) p.x += gp.L[0]; }
   if (gp.pbc & 2) { if (p.y > 0.5 * gp.L[1]) p.y -= gp.L[1]; if (p.y < -0.5 * gp.L[1]) p.y += gp.L[1]; }
   if (gp.pbc & 4) { if (p.z > 0.5 * gp.L[2]) p.z -= gp.L[2]; if (p.z < -0.5 * gp.L[2]) p.z += gp.L[2]; }
   pos[i] = p;
   int cx, cy, cz;
   cell_coords(gp, p.x, p.y, p.z, true, cx, cy, cz);
   int c = cell_linear(gp, cx, cy, cz);
   cid[i] = c;
   rank[i] = atomicAdd(&cell_cnt[c], 1);
}
__global__ void k_scatter_order(int n, const int *cid, const int *rank, const int *cell_start, int *order)
{
   int i = blockIdx.x * blockDim.x + threadIdx.x;
   if (i >= n) return;
   order[cell_start[cid[i]] + rank[i]] = i;
}
/* make the in-cell order deterministic: ascending previous index */
__global__ void k_sort_cells(int ncell, const int *cell_start, const int *cell_cnt, int *order)
{
   int c = blockIdx.x * blockDim.x + threadIdx.x;
   if (c >= ncell) return;
   int s = cell_start[c], n = cell_cnt[c];
   for (int a = 1; a < n; a++)
   {
      int v = order[s + a];
      int b = a - 1;
      while (b >= 0 && order[s + b] > v) { order[s + b + 1] = order[s + b]; b--; }
      order[s + b + 1] = v;
   }
}
__global__ void k_gather_state(int nloc, const int *order,
                               const double4 *pos, const double *vx, const double *vy, const double *vz,
                               const int *species, const int *group, const uint64_t *gid, const int *orig,
                               double4 *pos2, double *vx2, double *vy2, double *vz2,
                               int *species2, int *group2, uint64_t *gid2, int *orig2, int *slot_of_orig)
{
   int k = blockIdx.x * blockDim.x + threadIdx.x;
   if (k >= nloc) return;
   int i = order[k];
   pos2[k] = pos[i];
   vx2[k] = vx[i]; vy2[k] = vy[i]; vz2[k] = vz[i];
   species2[k] = species[i]; group2[k] = group[i]; gid2[k] = gid[i];
   int o = orig[i];
   orig2[k] = o;
   slot_of_orig[o] = k;
}

/* ------------------------------------------------------------------------- */
/* periodic image atoms                                                       */
__device__ __forceinline__ void image_dirs(const GridParams &gp, const double4 &p, int d[3])
{
   double r[3] = {p.x, p.y, p.z};
#pragma unroll
   for (int a = 0; a < 3; a++)
   {
      d[a] = 0;
      if (gp.m[a] == 0) continue;               /* no image margin on this axis */
      if (r[a] < gp.lo[a] + gp.rlist) d[a] = +1;                                /* image at r+L */
      else if (r[a] >= gp.lo[a] + gp.n[a] / gp.cinv[a] - gp.rlist) d[a] = -1;   /* image at r-L */
   }
}
__global__ void k_count_images(GridParams gp, int nloc, const double4 *pos, int *nimg)
{
   int i = blockIdx.x * blockDim.x + threadIdx.x;
   if (i >= nloc) return;
   int d[3];
   image_dirs(gp, pos[i], d);
   nimg[i] = (1 + (d[0] != 0)) * (1 + (d[1] != 0)) * (1 + (d[2] != 0)) - 1;
}
__global__ void k_fill_images(GridParams gp, int nloc, const double4 *pos, const int *img_off,
                              int *hsrc, int *hshift, int *hcid, int *hrank, int *cell_cnt_h)
{
   int i = blockIdx.x * blockDim.x + threadIdx.x;
   if (i >= nloc) return;
   double4 p = pos[i];
   int d[3];
   image_dirs(gp, p, d);
   int k = img_off[i];
   for (int sz = 0; sz <= (d[2] != 0); sz++)
      for (int sy = 0; sy <= (d[1] != 0); sy++)
         for (int sx = 0; sx <= (d[0] != 0); sx++)
         {
            if (!(sx | sy | sz)) continue;
            int ix = sx * d[0], iy = sy * d[1], iz = sz * d[2];
            double x = p.x + ix * gp.L[0], y = p.y + iy * gp.L[1], z = p.z + iz * gp.L[2];
            int cx, cy, cz;
            cell_coords(gp, x, y, z, false, cx, cy, cz);
            int c = cell_linear(gp, cx, cy, cz);
            hsrc[k] = i;
            hshift[k] = (ix + 1) + 3 * (iy + 1) + 9 * (iz + 1);
            hcid[k] = c;
            hrank[k] = atomicAdd(&cell_cnt_h[c], 1);
            k++;
         }
}
__global__ void k_gather_halo(int nhalo, const int *horder, const int *hsrc_t, const int *hshift_t, int *halo_src, int *halo_shift)
{
   int h = blockIdx.x * blockDim.x + threadIdx.x;
   if (h >= nhalo) return;
   int k = horder[h];
   halo_src[h] = hsrc_t[k];
   halo_shift[h] = hshift_t[k];
}
/* refresh image atoms from their sources: every step (replaces ddcUpdate's
 * position halo for the self-image case, ddcUpdate.c:40-85) */
__global__ void k_halo_update(int nloc, int nhalo, const int *halo_src, const int *halo_shift, double L0, double L1, double L2,
                              double4 *pos, uint64_t *gid, bool with_gid)
{
   int h = blockIdx.x * blockDim.x + threadIdx.x;
   if (h >= nhalo) return;
   int s = halo_src[h], code = halo_shift[h];
   double4 p = pos[s];
   p.x += (double)(code % 3 - 1) * L0;
   p.y += (double)((code / 3) % 3 - 1) * L1;
   p.z += (double)(code / 9 - 1) * L2;
   pos[nloc + h] = p;
   if (with_gid) gid[nloc + h] = gid[s];
}
__global__ void k_merge_cells(int ncell, int nloc, const int *cnt_o, const int *start_o, const int *cnt_h, const int *start_h, int *cell_start, int *cell_cnt)
{
   int c = blockIdx.x * blockDim.x + threadIdx.x;
   if (c >= ncell) return;
   int co = cnt_o[c], ch = cnt_h[c];
   cell_start[c] = co > 0 ? start_o[c] : nloc + start_h[c];
   cell_cnt[c] = co + ch;
}

/* ------------------------------------------------------------------------- */
/* neighbour list: pairlist1 semantics (pairlist.c:205-314) as a FULL list --
 * every j != i with |r_ij| < rmax+deltaR -- plus the reOrgPairs split
 * (bioMartini.c:1392-1485) done at build time. */
__global__ __launch_bounds__(DDCMI_BLOCK) void k_build_list(GridParams gp, int nloc, int npad, const double4 *pos, const uint64_t *gid, const int *species,
                                                            const int *cell_start, const int *cell_cnt,
                                                            int nmoltype, const int *moltype_sp, const int *mol_nspecies, const int *bpair_off,
                                                            const int *bpairI, const int *bpairJ,
                                                            int maxnbr, int *nbr, int *nbr_cnt, int maxexcl, int *excl, int *excl_cnt, int *flags)
{
   int i = blockIdx.x * blockDim.x + threadIdx.x;
   if (i >= nloc) return;
   double4 pi = pos[i];
   int cx, cy, cz;
   cell_coords(gp, pi.x, pi.y, pi.z, true, cx, cy, cz);
   double rl2 = gp.rlist * gp.rlist;
   uint64_t gi = 0;
   int mt = 0, mns = 1;
   if (nmoltype > 0) { gi = gid[i]; mt = moltype_sp[species[i]]; mns = mol_nspecies[mt]; }
   int cnt = 0, ecnt = 0;
   for (int dz = -2; dz <= 2; dz++)
   {
      int jz = cz + dz;
      if (jz < 0 || jz >= gp.g[2]) continue;
      for (int dy = -2; dy <= 2; dy++)
      {
         int jy = cy + dy;
         if (jy < 0 || jy >= gp.g[1]) continue;
         for (int dx = -2; dx <= 2; dx++)
         {
            int jx = cx + dx;
            if (jx < 0 || jx >= gp.g[0]) continue;
            int c = cell_linear(gp, jx, jy, jz);
            int s = cell_start[c], n = cell_cnt[c];
            for (int j = s; j < s + n; j++)
            {
               if (j == i) continue;
               double4 pj = pos[j];
               double x = pi.x - pj.x, y = pi.y - pj.y, z = pi.z - pj.z;
               double r2 = x * x + y * y + z * z;
               if (r2 < rl2)
               {
                  bool pruned = false;
                  if (nmoltype > 0)
                  {
                     uint64_t gj = gid[j];
                     if ((gi >> 32) == (gj >> 32))
                     {
                        if (mns > 1)
                        {
                           unsigned aI = (unsigned)(gi & 65535ull), aJ = (unsigned)(gj & 65535ull);
                           for (int k = bpair_off[mt]; k < bpair_off[mt + 1]; k++)
                           {
                              unsigned eI = (unsigned)bpairI[k], eJ = (unsigned)bpairJ[k];
                              if ((aI == eI && aJ == eJ) || (aJ == eI && aI == eJ)) { pruned = true; break; }
                           }
                        }
                        else pruned = true;
                     }
                  }
                  if (!pruned)
                  {
                     if (cnt < maxnbr) nbr[(size_t)cnt * npad + i] = j;
                     cnt++;
                  }
                  else
                  {
                     if (ecnt < maxexcl) excl[(size_t)ecnt * npad + i] = j;
                     ecnt++;
                  }
               }
            }
         }
      }
   }
   nbr_cnt[i] = min(cnt, maxnbr);
   excl_cnt[i] = min(ecnt, maxexcl);
   if (cnt > maxnbr) atomicMax(&flags[0], cnt);
   if (ecnt > maxexcl) atomicMax(&flags[1], ecnt);
   atomicMax(&flags[2], cnt);
   atomicMax(&flags[3], ecnt);
}
__global__ void k_sum_counts(int n, const int *a, const int *b, unsigned long long *out)
{
   int i = blockIdx.x * blockDim.x + threadIdx.x;
   unsigned long long va = (i < n) ? (unsigned long long)a[i] : 0ull, vb = (i < n) ? (unsigned long long)b[i] : 0ull;
   for (int off = 32; off > 0; off >>= 1) { va += __shfl_down(va, off, 64); vb += __shfl_down(vb, off, 64); }
   if ((threadIdx.x & 63) == 0) { atomicAdd(&out[0], va); atomicAdd(&out[1], vb); }
}

/* ------------------------------------------------------------------------- */
/* THE hot kernel: martiniNonBond (bioMartini.c:989-1122) + martiniIntraMoleReaction
 * (:1124-1208) over the full list.  One lane per owned atom; the ELL list is
 * slot-major so a wave reads 256 contiguous bytes per slot; neighbour records
 * are gathered from L1/L2 (cell-sorted => local).  The LJ table sits in LDS.
 *
 *   ir   = 1/sqrt(r2)           (v_rsq_f32 seed + 2 Newton steps: full FP64)
 *   s2   = sigma^2 ir^2 ; s6 = s2^3 ; s12 = s6^2
 *   vLJ += 4eps(s12-s6)+shift ; dvdr = 24eps(s6-2s12) ir^2
 *   vEle+= kqij(ir + krf r2 - crf) ; dvdr += kqij(2krf - ir^3)
 *   f_i -= dvdr d ;  virial += f (x) d
 */
__device__ __forceinline__ double rsqrt_f64(double x)
{
   float xf = (float)x;
   double y = (double)__frsqrt_rn(xf);
   double h = 0.5 * x;
   y = y * (1.5 - h * y * y);
   y = y * (1.5 - h * y * y);
   /* third step is free of charge accuracy-wise only when the seed is poor; two
    * steps from a 23-bit seed give < 2 ulp, checked against sqrt(1/x) in tests */
   return y;
}

template <bool HAS_Q>
__global__ __launch_bounds__(DDCMI_BLOCK) void k_nonbond(int nloc, int npad, int nblocks_logical,
                                                         const double4 *__restrict__ pos, const double *__restrict__ qatom,
                                                         const int *__restrict__ nbr, const int *__restrict__ nbr_cnt,
                                                         const int *__restrict__ excl, const int *__restrict__ excl_cnt,
                                                         const double4 *__restrict__ ljtab, int nlj,
                                                         double rc2, double krf, double crf, double keR,
                                                         double *__restrict__ fx, double *__restrict__ fy, double *__restrict__ fz,
                                                         double *__restrict__ partials)
{
   extern __shared__ double4 s_lj[];
   /* XCD-aware mapping: hardware deals blocks round-robin over the 8 XCDs, so
    * give XCD x the contiguous tile range [x*per, (x+1)*per): neighbouring tiles
    * then share one L2 (speed only, never correctness). */
   int per = (nblocks_logical + 7) >> 3;
   int lb = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
   for (int t = threadIdx.x; t < nlj * nlj; t += blockDim.x) s_lj[t] = ljtab[t];
   __syncthreads();
   double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};   /* vLJ, vEle, xx,yy,zz,xy,xz,yz */
   if (lb < nblocks_logical)
   {
      int i = lb * DDCMI_BLOCK + threadIdx.x;
      bool active = i < nloc;
      int ii = active ? i : nloc - 1;
      double4 pi = pos[ii];
      int ti = (int)(__double_as_longlong(pi.w) & 0xffffffffll);
      double kqi = 0.0;
      if (HAS_Q) kqi = keR * qatom[ii];
      int cnt = active ? nbr_cnt[i] : 0;
      double fxi = 0, fyi = 0, fzi = 0;
      const int *col = nbr + ii;
      for (int k = 0; k < cnt; k++)
      {
         int j = col[(size_t)k * npad];
         double4 pj = pos[j];
         double x = pi.x - pj.x, y = pi.y - pj.y, z = pi.z - pj.z;
         double r2 = x * x + y * y + z * z;
         if (r2 < rc2)
         {
            int tj = (int)(__double_as_longlong(pj.w) & 0xffffffffll);
            double4 lj = s_lj[ti * nlj + tj];          /* {sigma^2, 4eps, shift, 24eps} */
            double ir = rsqrt_f64(r2);
            double ir2 = ir * ir;
            double s2 = lj.x * ir2;
            double s4 = s2 * s2;
            double s6 = s4 * s2;
            double s12 = s6 * s6;
            acc[0] += lj.y * (s12 - s6) + lj.z;
            double dvdr = lj.w * (s6 - 2.0 * s12) * ir2;
            if (HAS_Q)
            {
               double kqij = kqi * qatom[j];
               acc[1] += kqij * (ir + krf * r2 - crf);
               dvdr += kqij * (2.0 * krf - ir2 * ir);
            }
            double fxij = -dvdr * x, fyij = -dvdr * y, fzij = -dvdr * z;
            fxi += fxij; fyi += fyij; fzi += fzij;
            acc[2] += fxij * x; acc[3] += fyij * y; acc[4] += fzij * z;
            acc[5] += fxij * y; acc[6] += fxij * z; acc[7] += fyij * z;
         }
      }
      if (HAS_Q)
      {
         /* excluded (same-molecule bonded) pairs: reaction-field correction only */
         int ecnt = active ? excl_cnt[i] : 0;
         const int *ecol = excl + ii;
         for (int k = 0; k < ecnt; k++)
         {
            int j = ecol[(size_t)k * npad];
            double4 pj = pos[j];
            double x = pi.x - pj.x, y = pi.y - pj.y, z = pi.z - pj.z;
            double r2 = x * x + y * y + z * z;
            if (r2 < rc2)
            {
               double kqij = kqi * qatom[j];
               acc[1] += kqij * (krf * r2 - crf);
               double dvdr = kqij * (2.0 * krf);
               double fxij = -dvdr * x, fyij = -dvdr * y, fzij = -dvdr * z;
               fxi += fxij; fyi += fyij; fzi += fzij;
               acc[2] += fxij * x; acc[3] += fyij * y; acc[4] += fzij * z;
               acc[5] += fxij * y; acc[6] += fxij * z; acc[7] += fyij * z;
            }
         }
      }
      if (active) { fx[i] = fxi; fy[i] = fyi; fz[i] = fzi; }
   }
   if (lb < nblocks_logical) block_reduce_store<8>(acc, partials + (size_t)lb * 8);
}

/* zero forces (nonbonded excluded via excludePotentialTerm) */
__global__ void k_zero3(int n, double *a, double *b, double *c)
{
   int i = blockIdx.x * blockDim.x + threadIdx.x;
   if (i < n) { a[i] = 0; b[i] = 0; c[i] = 0; }
}

/* fixed-order second stage: out[k] = sum_b partials[b*stride+k] */
__global__ __launch_bounds__(DDCMI_BLOCK) void k_reduce_partials(const double *partials, int nblocks, int stride, int nv, double *out)
{
   __shared__ double s[DDCMI_BLOCK];
   for (int k = 0; k < nv; k++)
   {
      double a = 0.0;
      for (int b = threadIdx.x; b < nblocks; b += DDCMI_BLOCK) a += partials[(size_t)b * stride + k];
      s[threadIdx.x] = a;
      __syncthreads();
      for (int off = DDCMI_BLOCK / 2; off > 0; off >>= 1)
      {
         if (threadIdx.x < off) s[threadIdx.x] += s[threadIdx.x + off];
         __syncthreads();
      }
      if (threadIdx.x == 0) out[k] = s[0];
      __syncthreads();
   }
}
/* final energies / virial: full list counts every pair twice */
__global__ void k_finish_energy(double *r, double self_ele)
{
   if (threadIdx.x != 0 || blockIdx.x != 0) return;
   double lj = 0.5 * r[R_NB_LJ];
   double ele = 0.5 * r[R_NB_ELE] + self_ele;
   r[R_E + DDCMI_E_LJ] = lj;
   r[R_E + DDCMI_E_ELE] = ele;
   /* bonded scratch: bond {e,vir6} angle {e,vir6} tors {e_tors,e_impr,vir6} */
   double eb[4] = {r[R_SCR_BOND], r[R_SCR_ANGLE], r[R_SCR_TORS], r[R_SCR_TORS + 1]};
   double etot = lj + ele;
   for (int k = 0; k < 4; k++) { r[R_E + DDCMI_E_BOND + k] = eb[k]; etot += eb[k]; }
   r[R_E + DDCMI_E_TOTAL] = etot;
   for (int k = 0; k < 6; k++)
      r[R_VIR + k] = 0.5 * r[R_NB_VIR + k] + ((r[R_SCR_BOND + 1 + k] + r[R_SCR_ANGLE + 1 + k]) + r[R_SCR_TORS + 2 + k]);
}

/* ------------------------------------------------------------------------- */
/* NGLF integrator kernels (nglf.c:67-112)                                    */
/* FRONT half kick (free.c:13-28 / berendsen.c:64-89) fused with the drift
 * (nglf.c:80-87).  The wrap of nglf.c:90 is applied at rebuild/download time
 * instead (positions stay continuous between rebuilds so image atoms and the
 * list remain valid); the downloaded coordinates are identical up to rounding. */
struct GroupLambda { double v[32]; };
__global__ void k_kick_drift(int nloc, double dt, const double *__restrict__ invmass, const int *__restrict__ species,
                             const int *__restrict__ group, GroupLambda glambda,
                             const double *__restrict__ fx, const double *__restrict__ fy, const double *__restrict__ fz,
                             double *__restrict__ vx, double *__restrict__ vy, double *__restrict__ vz, double4 *__restrict__ pos)
{
   int i = blockIdx.x * blockDim.x + threadIdx.x;
   if (i >= nloc) return;
   double a = (0.5 * dt) * invmass[species[i]];
   double lam = glambda.v[group[i] & 31];
   double x = vx[i], y = vy[i], z = vz[i];
   if (lam != 1.0) { x *= lam; y *= lam; z *= lam; }
   x += a * fx[i]; y += a * fy[i]; z += a * fz[i];
   vx[i] = x; vy[i] = y; vz[i] = z;
   double4 p = pos[i];
   p.x += dt * x; p.y += dt * y; p.z += dt * z;
   pos[i] = p;
}
/* BACK half kick (nglf.c:100-104) fused with kinetic_terms (energy.c:48-163):
 * rk = sum 1/2 m v^2, tion = sum m v (x) v */
__global__ __launch_bounds__(DDCMI_BLOCK) void k_kick_ke(int nloc, double dt, const double *__restrict__ invmass, const double *__restrict__ massv,
                                                         const int *__restrict__ species,
                                                         const double *__restrict__ fx, const double *__restrict__ fy, const double *__restrict__ fz,
                                                         double *__restrict__ vx, double *__restrict__ vy, double *__restrict__ vz,
                                                         double *__restrict__ partials, int do_kick)
{
   int i = blockIdx.x * blockDim.x + threadIdx.x;
   double acc[7] = {0, 0, 0, 0, 0, 0, 0};
   if (i < nloc)
   {
      int sp = species[i];
      double x = vx[i], y = vy[i], z = vz[i];
      if (do_kick)
      {
         double a = (0.5 * dt) * invmass[sp];
         x += a * fx[i]; y += a * fy[i]; z += a * fz[i];
         vx[i] = x; vy[i] = y; vz[i] = z;
      }
      double m = massv[sp];
      double vxx = x * x, vyy = y * y, vzz = z * z;
      acc[0] = 0.5 * m * (vxx + vyy + vzz);
      acc[1] = m * vxx; acc[2] = m * vyy; acc[3] = m * vzz;
      acc[4] = m * (x * y); acc[5] = m * (x * z); acc[6] = m * (y * z);
   }
   block_reduce_store<7>(acc, partials + (size_t)blockIdx.x * 8);
}
/* per-group kinetic energy and member count (energy.c:124-133) */
__global__ __launch_bounds__(DDCMI_BLOCK) void k_group_ke(int nloc, int ngroup, const double *__restrict__ massv, const int *__restrict__ species,
                                                          const int *__restrict__ group,
                                                          const double *__restrict__ vx, const double *__restrict__ vy, const double *__restrict__ vz,
                                                          double *out /* [2*ngroup], zeroed */)
{
   int i = blockIdx.x * blockDim.x + threadIdx.x;
   for (int g = 0; g < ngroup; g++)
   {
      double k = 0.0, c = 0.0;
      if (i < nloc && group[i] == g)
      {
         double m = massv[species[i]];
         k = 0.5 * m * (vx[i] * vx[i] + vy[i] * vy[i] + vz[i] * vz[i]);
         c = 1.0;
      }
      k = wave_sum(k); c = wave_sum(c);
      if ((threadIdx.x & 63) == 0 && c > 0.0) { atomicAdd(&out[2 * g], k); atomicAdd(&out[2 * g + 1], c); }
   }
}

/* download helpers: caller order + wrap */
__global__ void k_export_pos(GridParams gp, int nloc, const double4 *pos, const int *orig, double *ox, double *oy, double *oz)
{
   int i = blockIdx.x * blockDim.x + threadIdx.x;
   if (i >= nloc) return;
   double4 p = pos[i];
   if (gp.pbc & 1) { if (p.x > 0.5 * gp.L[0]) p.x -= gp.L[0]; if (p.x < -0.5 * gp.L[0]) p.x += gp.L[0]; }
   if (gp.pbc & 2) { if (p.y > 0.5 * gp.L[1]) p.y -= gp.L[1]; if (p.y < -0.5 * gp.L[1]) p.y += gp.L[1]; }
   if (gp.pbc & 4) { if (p.z > 0.5 * gp.L[2]) p.z -= gp.L[2]; if (p.z < -0.5 * gp.L[2]) p.z += gp.L[2]; }
   int o = orig[i];
   ox[o] = p.x; oy[o] = p.y; oz[o] = p.z;
}
__global__ void k_export3(int nloc, const double *a, const double *b, const double *c, const int *orig, double *oa, double *ob, double *oc)
{
   int i = blockIdx.x * blockDim.x + threadIdx.x;
   if (i >= nloc) return;
   int o = orig[i];
   oa[o] = a[i]; ob[o] = b[i]; oc[o] = c[i];
}
__global__ void k_init_state(int n, const double *rx, const double *ry, const double *rz, const int *species, const int *ljtype_sp,
                             const double *charge_sp, double4 *pos, double *qatom, int *orig, int *slot)
{
   int i = blockIdx.x * blockDim.x + threadIdx.x;
   if (i >= n) return;
   int sp = species[i];
   long long w = ((long long)sp << 32) | (long long)(unsigned)ljtype_sp[sp];
   pos[i] = make_double4(rx[i], ry[i], rz[i], __longlong_as_double(w));
   qatom[i] = charge_sp[sp];
   orig[i] = i;
   slot[i] = i;
}
__global__ void k_fill_q(int n, const double4 *pos, const double *charge_sp, double *qatom)
{
   int i = blockIdx.x * blockDim.x + threadIdx.x;
   if (i >= n) return;
   int sp = (int)(__double_as_longlong(pos[i].w) >> 32);
   qatom[i] = charge_sp[sp];
}
__global__ void k_list_to_csr(int nloc, int npad, const int *lst, const int *cnt, const int *orig, const int *halo_src, const int *start, int *jout)
{
   int i = blockIdx.x * blockDim.x + threadIdx.x;
   if (i >= nloc) return;
   int o = orig[i];
   int s = start[o];
   for (int k = 0; k < cnt[i]; k++)
   {
      int j = lst[(size_t)k * npad + i];
      if (j >= nloc) j = halo_src[j - nloc];
      jout[s + k] = orig[j];
   }
}
__global__ void k_counts_by_orig(int nloc, const int *cnt, const int *orig, int *out)
{
   int i = blockIdx.x * blockDim.x + threadIdx.x;
   if (i < nloc) out[orig[i]] = cnt[i];
}

/* ------------------------------------------------------------------------- */
/* context                                                                    */
extern "C" const char *ddcmi_version(void) { return "ddcmi 0.1 (gfx950)"; }

extern "C" int ddcmi_device_count(void)
{
   int n = 0;
   if (hipGetDeviceCount(&n) != hipSuccess) return 0;
   return n;
}

extern "C" const char *ddcmi_last_error(const ddcmi_ctx *ctx) { return ctx ? ctx->err.c_str() : g_create_err.c_str(); }

extern "C" int ddcmi_create(ddcmi_ctx **out, int device)
{
   if (!out) return DDCMI_EINVAL;
   *out = nullptr;
   int n = 0;
   hipError_t e = hipGetDeviceCount(&n);
   if (e != hipSuccess || n <= 0)
   {
      g_create_err = std::string("no HIP device available: ") + hipGetErrorString(e);
      return DDCMI_ENODEVICE;
   }
   if (device < 0 || device >= n) { g_create_err = "device ordinal out of range"; return DDCMI_EINVAL; }
   if ((e = hipSetDevice(device)) != hipSuccess) { g_create_err = hipGetErrorString(e); return DDCMI_ENODEVICE; }
   ddcmi_ctx *ctx = new ddcmi_ctx();
   ctx->device = device;
   if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess ||
       hipMalloc((void **)&ctx->d_results, R_SIZE * sizeof(double)) != hipSuccess ||
       hipHostMalloc((void **)&ctx->h_results, R_SIZE * sizeof(double), hipHostMallocDefault) != hipSuccess ||
       hipMalloc((void **)&ctx->d_flags, 16 * sizeof(int)) != hipSuccess ||
       hipHostMalloc((void **)&ctx->h_flags, 16 * sizeof(int), hipHostMallocDefault) != hipSuccess)
   {
      g_create_err = "context allocation failed";
      delete ctx;
      return DDCMI_ENOMEM;
   }
   (void)hipMemset(ctx->d_results, 0, R_SIZE * sizeof(double));
   memset(ctx->h_results, 0, R_SIZE * sizeof(double));
   ctx->gtype.assign(1, DDCMI_FREE); ctx->ginterval.assign(1, 1); ctx->gTeq.assign(1, 0); ctx->gtau.assign(1, 0);
   ctx->glambda.assign(1, 1.0); ctx->gTsum.assign(1, 0); ctx->gT.assign(1, 0); ctx->gnT.assign(1, 0); ctx->gdoScaling.assign(1, 0);
   *out = ctx;
   return DDCMI_OK;
}

extern "C" void ddcmi_destroy(ddcmi_ctx *ctx)
{
   if (!ctx) return;
   (void)hipSetDevice(ctx->device);
   (void)hipStreamSynchronize(ctx->stream);
   ddcmi_comm_destroy(ctx);
   dbuf<double> *db[] = {&ctx->d_invmass, &ctx->d_mass, &ctx->d_charge_sp, &ctx->bpartials, &ctx->vx, &ctx->vy, &ctx->vz, &ctx->vx2, &ctx->vy2, &ctx->vz2,
                         &ctx->fx, &ctx->fy, &ctx->fz, &ctx->qatom, &ctx->bond_kb, &ctx->bond_b0, &ctx->angle_k, &ctx->angle_t0, &ctx->tors_k, &ctx->tors_delta, &ctx->partials};
   for (auto b : db) b->release();
   dbuf<int> *ib[] = {&ctx->d_ljtype_sp, &ctx->d_moltype_sp, &ctx->d_mol_nspecies, &ctx->d_bpair_off, &ctx->d_bpairI, &ctx->d_bpairJ, &ctx->species, &ctx->species2,
                      &ctx->group, &ctx->group2, &ctx->orig, &ctx->orig2, &ctx->slot_of_orig, &ctx->cid, &ctx->crank, &ctx->order, &ctx->cell_cnt_o, &ctx->cell_start_o,
                      &ctx->cell_cnt_h, &ctx->cell_start_h, &ctx->cell_start, &ctx->cell_cnt, &ctx->nimg, &ctx->img_off, &ctx->hsrc_t, &ctx->hshift_t, &ctx->hcid, &ctx->hrank,
                      &ctx->horder, &ctx->halo_src, &ctx->halo_shift, &ctx->scan_tmp, &ctx->nbr, &ctx->nbr_cnt, &ctx->excl, &ctx->excl_cnt,
                      &ctx->bond_ij, &ctx->angle_ijk, &ctx->angle_func, &ctx->tors_ijkl, &ctx->tors_func, &ctx->tors_n};
   for (auto b : ib) b->release();
   ctx->pos.release(); ctx->pos2.release(); ctx->d_ljtab.release(); ctx->gid.release(); ctx->gid2.release();
   for (auto &e : ctx->ev) (void)hipEventDestroy(e);
   if (ctx->d_results) (void)hipFree(ctx->d_results);
   if (ctx->h_results) (void)hipHostFree(ctx->h_results);
   if (ctx->d_flags) (void)hipFree(ctx->d_flags);
   if (ctx->h_flags) (void)hipHostFree(ctx->h_flags);
   (void)hipStreamDestroy(ctx->stream);
   delete ctx;
}

template <class T>
static int upload_vec(ddcmi_ctx *ctx, dbuf<T> &buf, const T *src, size_t n)
{
   if (n == 0) return DDCMI_OK;
   ENSURE(ctx, buf, n);
   HIPCHK(ctx, hipMemcpyAsync(buf.p, src, n * sizeof(T), hipMemcpyHostToDevice, ctx->stream));
   HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
   return DDCMI_OK;
}

extern "C" int ddcmi_set_box(ddcmi_ctx *ctx, const double h[9], int pbc)
{
   if (!ctx || !h) return DDCMI_EINVAL;
   const int off[6] = {1, 2, 3, 5, 6, 7};
   for (int k = 0; k < 6; k++)
      if (fabs(h[off[k]]) > 1e-10) SETERR(ctx, DDCMI_EUNSUPPORTED, "only orthorhombic boxes are supported (h[%d]=%g)", off[k], h[off[k]]);
   if (!(h[0] > 0 && h[4] > 0 && h[8] > 0)) SETERR(ctx, DDCMI_EINVAL, "box lengths must be positive");
   memcpy(ctx->h, h, sizeof(double) * 9);
   ctx->pbc = pbc;
   ctx->have_box = true;
   ctx->list_valid = false;
   return DDCMI_OK;
}

extern "C" int ddcmi_set_species(ddcmi_ctx *ctx, int nspecies, const double *mass, const double *charge, const int *ljtype, const int *moltype)
{
   if (!ctx || nspecies <= 0 || !mass || !ljtype) return DDCMI_EINVAL;
   (void)hipSetDevice(ctx->device);
   ctx->nspecies = nspecies;
   ctx->mass.assign(mass, mass + nspecies);
   ctx->charge.assign(nspecies, 0.0);
   if (charge) ctx->charge.assign(charge, charge + nspecies);
   ctx->ljtype.assign(ljtype, ljtype + nspecies);
   ctx->moltype.assign(nspecies, 0);
   if (moltype) ctx->moltype.assign(moltype, moltype + nspecies);
   ctx->has_charge = false;
   std::vector<double> inv(nspecies);
   for (int s = 0; s < nspecies; s++)
   {
      if (!(mass[s] > 0)) SETERR(ctx, DDCMI_EINVAL, "species %d has non-positive mass", s);
      inv[s] = 1.0 / mass[s];
      if (ctx->charge[s] != 0.0) ctx->has_charge = true;
   }
   int rc;
   if ((rc = upload_vec(ctx, ctx->d_invmass, inv.data(), nspecies))) return rc;
   if ((rc = upload_vec(ctx, ctx->d_mass, ctx->mass.data(), nspecies))) return rc;
   if ((rc = upload_vec(ctx, ctx->d_charge_sp, ctx->charge.data(), nspecies))) return rc;
   if ((rc = upload_vec(ctx, ctx->d_ljtype_sp, ctx->ljtype.data(), nspecies))) return rc;
   if ((rc = upload_vec(ctx, ctx->d_moltype_sp, ctx->moltype.data(), nspecies))) return rc;
   return DDCMI_OK;
}

extern "C" int ddcmi_set_nonbonded(ddcmi_ctx *ctx, int nlj, const double *sigma, const double *eps, const double *shift,
                                   double rmax, double keR, double krf, double crf)
{
   if (!ctx || nlj <= 0 || !sigma || !eps || !shift || !(rmax > 0)) return DDCMI_EINVAL;
   (void)hipSetDevice(ctx->device);
   ctx->nlj = nlj;
   ctx->sigma.assign(sigma, sigma + nlj * nlj); ctx->eps.assign(eps, eps + nlj * nlj); ctx->shift.assign(shift, shift + nlj * nlj);
   ctx->rmax = rmax; ctx->keR = keR; ctx->krf = krf; ctx->crf = crf;
   std::vector<double4> tab(nlj * nlj);
   /* table index = ti*nlj + tj; the reference indexes sj + nspecies*si (bioMartini.c:1052) on a symmetric table */
   for (int k = 0; k < nlj * nlj; k++) tab[k] = make_double4(sigma[k] * sigma[k], 4.0 * eps[k], shift[k], 24.0 * eps[k]);
   ctx->list_valid = false;
   return upload_vec(ctx, ctx->d_ljtab, tab.data(), tab.size());
}

extern "C" int ddcmi_set_molecules(ddcmi_ctx *ctx, int nmoltype, const int *mol_nspecies, const int *bpair_off, const int *bpairI, const int *bpairJ)
{
   if (!ctx || nmoltype < 0) return DDCMI_EINVAL;
   (void)hipSetDevice(ctx->device);
   ctx->nmoltype = nmoltype;
   ctx->list_valid = false;
   if (nmoltype == 0) return DDCMI_OK;
   if (!mol_nspecies || !bpair_off) return DDCMI_EINVAL;
   ctx->mol_nspecies.assign(mol_nspecies, mol_nspecies + nmoltype);
   ctx->bpair_off.assign(bpair_off, bpair_off + nmoltype + 1);
   int nb = bpair_off[nmoltype];
   ctx->bpairI.assign(nb + 1, 0); ctx->bpairJ.assign(nb + 1, 0);
   for (int k = 0; k < nb; k++) { ctx->bpairI[k] = bpairI[k]; ctx->bpairJ[k] = bpairJ[k]; }
   int rc;
   if ((rc = upload_vec(ctx, ctx->d_mol_nspecies, ctx->mol_nspecies.data(), nmoltype))) return rc;
   if ((rc = upload_vec(ctx, ctx->d_bpair_off, ctx->bpair_off.data(), nmoltype + 1))) return rc;
   if ((rc = upload_vec(ctx, ctx->d_bpairI, ctx->bpairI.data(), nb + 1))) return rc;
   if ((rc = upload_vec(ctx, ctx->d_bpairJ, ctx->bpairJ.data(), nb + 1))) return rc;
   return DDCMI_OK;
}

extern "C" int ddcmi_set_neighbor(ddcmi_ctx *ctx, double deltaR, int updateRate)
{
   if (!ctx || deltaR < 0) return DDCMI_EINVAL;
   if (updateRate <= 0) SETERR(ctx, DDCMI_EUNSUPPORTED, "updateRate must be > 0 (displacement-triggered rebuilds, ddcUpdateAll.c:56, are not implemented)");
   ctx->deltaR = deltaR; ctx->updateRate = updateRate; ctx->list_valid = false;
   return DDCMI_OK;
}

extern "C" int ddcmi_set_groups(ddcmi_ctx *ctx, int ngroup, const int *type, const double *Teq, const double *tau, const int *interval)
{
   if (!ctx || ngroup <= 0 || ngroup > 32 || !type) return DDCMI_EINVAL;
   ctx->ngroup = ngroup;
   ctx->gtype.assign(type, type + ngroup);
   ctx->gTeq.assign(ngroup, 0.0); ctx->gtau.assign(ngroup, 0.0); ctx->ginterval.assign(ngroup, 1);
   for (int g = 0; g < ngroup; g++)
   {
      if (type[g] != DDCMI_FREE && type[g] != DDCMI_BERENDSEN) SETERR(ctx, DDCMI_EUNSUPPORTED, "group %d: only FREE and BERENDSEN groups are supported", g);
      if (Teq) ctx->gTeq[g] = Teq[g];
      if (tau) ctx->gtau[g] = tau[g];
      if (interval && interval[g] > 0) ctx->ginterval[g] = interval[g];
   }
   ctx->glambda.assign(ngroup, 1.0); ctx->gTsum.assign(ngroup, 0.0); ctx->gT.assign(ngroup, 0.0);
   ctx->gnT.assign(ngroup, 0); ctx->gdoScaling.assign(ngroup, 0);
   return DDCMI_OK;
}

extern "C" int ddcmi_set_clock(ddcmi_ctx *ctx, int64_t loop, double time)
{
   if (!ctx) return DDCMI_EINVAL;
   ctx->loop = loop; ctx->time = time;
   return DDCMI_OK;
}
extern "C" int ddcmi_get_clock(const ddcmi_ctx *ctx, int64_t *loop, double *time)
{
   if (!ctx) return DDCMI_EINVAL;
   if (loop) *loop = ctx->loop;
   if (time) *time = ctx->time;
   return DDCMI_OK;
}
extern "C" int ddcmi_nlocal(const ddcmi_ctx *ctx) { return ctx ? ctx->nloc : 0; }
extern "C" void *ddcmi_stream(ddcmi_ctx *ctx) { return ctx ? (void *)ctx->stream : nullptr; }
extern "C" int ddcmi_sync(ddcmi_ctx *ctx)
{
   if (!ctx) return DDCMI_EINVAL;
   HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
   return DDCMI_OK;
}

/* ------------------------------------------------------------------------- */
extern "C" int ddcmi_upload_state(ddcmi_ctx *ctx, int nlocal, const double *rx, const double *ry, const double *rz,
                                  const double *vx, const double *vy, const double *vz,
                                  const uint64_t *gid, const int *species, const int *group)
{
   if (!ctx || nlocal <= 0 || !rx || !ry || !rz || !species) return DDCMI_EINVAL;
   if (ctx->nspecies <= 0) SETERR(ctx, DDCMI_EINVAL, "ddcmi_set_species must be called before ddcmi_upload_state");
   (void)hipSetDevice(ctx->device);
   for (int i = 0; i < nlocal; i++)
      if (species[i] < 0 || species[i] >= ctx->nspecies) SETERR(ctx, DDCMI_EINVAL, "particle %d has species %d outside [0,%d)", i, species[i], ctx->nspecies);
   if (group)
      for (int i = 0; i < nlocal; i++)
         if (group[i] < 0 || group[i] >= ctx->ngroup) SETERR(ctx, DDCMI_EINVAL, "particle %d has group %d outside [0,%d)", i, group[i], ctx->ngroup);
   int n = nlocal;
   size_t cap = (size_t)n + n / 4 + 1024;     /* room for image atoms; grown on demand */
   ENSURE(ctx, ctx->pos, cap); ENSURE(ctx, ctx->pos2, cap); ENSURE(ctx, ctx->qatom, cap);
   ENSURE(ctx, ctx->gid, cap); ENSURE(ctx, ctx->gid2, cap);
   dbuf<double> *d3[] = {&ctx->vx, &ctx->vy, &ctx->vz, &ctx->vx2, &ctx->vy2, &ctx->vz2, &ctx->fx, &ctx->fy, &ctx->fz};
   for (auto b : d3) ENSURE(ctx, *b, n);
   dbuf<int> *i1[] = {&ctx->species, &ctx->species2, &ctx->group, &ctx->group2, &ctx->orig, &ctx->orig2, &ctx->slot_of_orig, &ctx->cid, &ctx->crank, &ctx->order, &ctx->nimg, &ctx->img_off};
   for (auto b : i1) ENSURE(ctx, *b, n + 1);
   /* stage through vx2/vy2/vz2 as scratch for the positions */
   HIPCHK(ctx, hipMemcpyAsync(ctx->vx2.p, rx, n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
   HIPCHK(ctx, hipMemcpyAsync(ctx->vy2.p, ry, n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
   HIPCHK(ctx, hipMemcpyAsync(ctx->vz2.p, rz, n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
   HIPCHK(ctx, hipMemcpyAsync(ctx->species.p, species, n * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
   if (group) HIPCHK(ctx, hipMemcpyAsync(ctx->group.p, group, n * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
   else HIPCHK(ctx, hipMemsetAsync(ctx->group.p, 0, n * sizeof(int), ctx->stream));
   if (gid) HIPCHK(ctx, hipMemcpyAsync(ctx->gid.p, gid, n * sizeof(uint64_t), hipMemcpyHostToDevice, ctx->stream));
   else
   {
      std::vector<uint64_t> g(n);
      for (int i = 0; i < n; i++) g[i] = (uint64_t)i << 32;
      HIPCHK(ctx, hipMemcpyAsync(ctx->gid.p, g.data(), n * sizeof(uint64_t), hipMemcpyHostToDevice, ctx->stream));
      HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
   }
   hipLaunchKernelGGL(k_init_state, dim3(cdiv(n, 256)), dim3(256), 0, ctx->stream, n, ctx->vx2.p, ctx->vy2.p, ctx->vz2.p, ctx->species.p,
                      ctx->d_ljtype_sp.p, ctx->d_charge_sp.p, ctx->pos.p, ctx->qatom.p, ctx->orig.p, ctx->slot_of_orig.p);
   HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
   if (vx && vy && vz)
   {
      HIPCHK(ctx, hipMemcpyAsync(ctx->vx.p, vx, n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
      HIPCHK(ctx, hipMemcpyAsync(ctx->vy.p, vy, n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
      HIPCHK(ctx, hipMemcpyAsync(ctx->vz.p, vz, n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
   }
   else
   {
      HIPCHK(ctx, hipMemsetAsync(ctx->vx.p, 0, n * sizeof(double), ctx->stream));
      HIPCHK(ctx, hipMemsetAsync(ctx->vy.p, 0, n * sizeof(double), ctx->stream));
      HIPCHK(ctx, hipMemsetAsync(ctx->vz.p, 0, n * sizeof(double), ctx->stream));
   }
   HIPCHK(ctx, hipMemsetAsync(ctx->fx.p, 0, n * sizeof(double), ctx->stream));
   HIPCHK(ctx, hipMemsetAsync(ctx->fy.p, 0, n * sizeof(double), ctx->stream));
   HIPCHK(ctx, hipMemsetAsync(ctx->fz.p, 0, n * sizeof(double), ctx->stream));
   HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
   ctx->nloc = n; ctx->nhalo = 0;
   ctx->npad = cdiv(n, DDCMI_BLOCK) * DDCMI_BLOCK;
   /* self electrostatic term -1/2 sum q_i^2 keR crf over local atoms (bioMartini.c:1030-1035) */
   double q2 = 0.0;
   for (int i = 0; i < n; i++) { double q = ctx->charge[species[i]]; q2 += q * q; }
   ctx->self_ele = -0.5 * q2 * ctx->keR * ctx->crf;
   ctx->list_valid = false; ctx->forces_valid = false;
   return DDCMI_OK;
}

extern "C" int ddcmi_download_state(ddcmi_ctx *ctx, int mask, double *rx, double *ry, double *rz, double *vx, double *vy, double *vz,
                                    double *fx, double *fy, double *fz)
{
   if (!ctx || ctx->nloc <= 0) return DDCMI_EINVAL;
   (void)hipSetDevice(ctx->device);
   int n = ctx->nloc, nb = cdiv(n, 256);
   /* scratch: vx2,vy2,vz2 are free between rebuilds */
   if ((mask & DDCMI_POS) && rx && ry && rz)
   {
      GridParams gp = ctx->gp;
      if (!ctx->list_valid) { gp.pbc = ctx->pbc; gp.L[0] = ctx->h[0]; gp.L[1] = ctx->h[4]; gp.L[2] = ctx->h[8]; }
      hipLaunchKernelGGL(k_export_pos, dim3(nb), dim3(256), 0, ctx->stream, gp, n, ctx->pos.p, ctx->orig.p, ctx->vx2.p, ctx->vy2.p, ctx->vz2.p);
      HIPCHK(ctx, hipMemcpyAsync(rx, ctx->vx2.p, n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
      HIPCHK(ctx, hipMemcpyAsync(ry, ctx->vy2.p, n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
      HIPCHK(ctx, hipMemcpyAsync(rz, ctx->vz2.p, n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
      HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
   }
   if ((mask & DDCMI_VEL) && vx && vy && vz)
   {
      hipLaunchKernelGGL(k_export3, dim3(nb), dim3(256), 0, ctx->stream, n, ctx->vx.p, ctx->vy.p, ctx->vz.p, ctx->orig.p, ctx->vx2.p, ctx->vy2.p, ctx->vz2.p);
      HIPCHK(ctx, hipMemcpyAsync(vx, ctx->vx2.p, n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
      HIPCHK(ctx, hipMemcpyAsync(vy, ctx->vy2.p, n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
      HIPCHK(ctx, hipMemcpyAsync(vz, ctx->vz2.p, n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
      HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
   }
   if ((mask & DDCMI_FORCE) && fx && fy && fz)
   {
      hipLaunchKernelGGL(k_export3, dim3(nb), dim3(256), 0, ctx->stream, n, ctx->fx.p, ctx->fy.p, ctx->fz.p, ctx->orig.p, ctx->vx2.p, ctx->vy2.p, ctx->vz2.p);
      HIPCHK(ctx, hipMemcpyAsync(fx, ctx->vx2.p, n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
      HIPCHK(ctx, hipMemcpyAsync(fy, ctx->vy2.p, n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
      HIPCHK(ctx, hipMemcpyAsync(fz, ctx->vz2.p, n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
      HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
   }
   return DDCMI_OK;
}

/* ------------------------------------------------------------------------- */
static int setup_grid(ddcmi_ctx *ctx)
{
   GridParams &gp = ctx->gp;
   double rlist = ctx->rmax + ctx->deltaR;
   gp.rlist = rlist;
   gp.pbc = ctx->pbc;
   double L[3] = {ctx->h[0], ctx->h[4], ctx->h[8]};
   long ncell = 1;
   for (int a = 0; a < 3; a++)
   {
      gp.L[a] = L[a];
      bool periodic = (ctx->pbc >> a) & 1;
      if (periodic && L[a] < 2.0 * rlist)
         SETERR(ctx, DDCMI_EUNSUPPORTED, "box length %g on axis %d is shorter than 2*(rmax+deltaR)=%g: the nearest-image convention the reference relies on breaks down", L[a], a, 2.0 * rlist);
      gp.lo[a] = -0.5 * L[a];
      double cmin = 0.5 * rlist;
      int n = (int)floor(L[a] / cmin);
      if (n < 1) n = 1;
      gp.n[a] = n;
      gp.cinv[a] = (double)n / L[a];
      gp.m[a] = periodic ? 2 : 0;
      gp.g[a] = n + 2 * gp.m[a];
      gp.T[a] = (gp.g[a] + 3) / 4;
      ncell *= gp.T[a] * 4;
   }
   if (ncell > 2000000000L) SETERR(ctx, DDCMI_EUNSUPPORTED, "cell grid too large");
   gp.ncell = (int)ncell;
   return DDCMI_OK;
}

extern "C" int ddcmi_build_list(ddcmi_ctx *ctx)
{
   if (!ctx) return DDCMI_EINVAL;
   if (ctx->nloc <= 0 || !ctx->have_box || ctx->nlj <= 0 || ctx->updateRate <= 0)
      SETERR(ctx, DDCMI_EINVAL, "ddcmi_build_list needs box, nonbonded parameters, neighbor settings and an uploaded state");
   (void)hipSetDevice(ctx->device);
   int rc = setup_grid(ctx);
   if (rc) return rc;
   GridParams &gp = ctx->gp;
   hipStream_t st = ctx->stream;
   int n = ctx->nloc, nb = cdiv(n, 256), ncell = gp.ncell, ncb = cdiv(ncell, 256);
   dbuf<int> *cb[] = {&ctx->cell_cnt_o, &ctx->cell_start_o, &ctx->cell_cnt_h, &ctx->cell_start_h, &ctx->cell_start, &ctx->cell_cnt};
   for (auto b : cb) ENSURE(ctx, *b, ncell + 1);
   /* 1. wrap + cell ids + counting sort of the owned atoms */
   HIPCHK(ctx, hipMemsetAsync(ctx->cell_cnt_o.p, 0, ncell * sizeof(int), st));
   HIPCHK(ctx, hipMemsetAsync(ctx->cell_cnt_h.p, 0, ncell * sizeof(int), st));
   hipLaunchKernelGGL(k_wrap_cell, dim3(nb), dim3(256), 0, st, gp, n, ctx->pos.p, ctx->cid.p, ctx->crank.p, ctx->cell_cnt_o.p);
   HIPCHK(ctx, hipMemcpyAsync(ctx->cell_start_o.p, ctx->cell_cnt_o.p, ncell * sizeof(int), hipMemcpyDeviceToDevice, st));
   if ((rc = ddcmi_scan_exclusive(ctx, ctx->cell_start_o.p, ncell, nullptr))) return rc;
   hipLaunchKernelGGL(k_scatter_order, dim3(nb), dim3(256), 0, st, n, ctx->cid.p, ctx->crank.p, ctx->cell_start_o.p, ctx->order.p);
   hipLaunchKernelGGL(k_sort_cells, dim3(ncb), dim3(256), 0, st, ncell, ctx->cell_start_o.p, ctx->cell_cnt_o.p, ctx->order.p);
   hipLaunchKernelGGL(k_gather_state, dim3(nb), dim3(256), 0, st, n, ctx->order.p, ctx->pos.p, ctx->vx.p, ctx->vy.p, ctx->vz.p,
                      ctx->species.p, ctx->group.p, ctx->gid.p, ctx->orig.p,
                      ctx->pos2.p, ctx->vx2.p, ctx->vy2.p, ctx->vz2.p, ctx->species2.p, ctx->group2.p, ctx->gid2.p, ctx->orig2.p, ctx->slot_of_orig.p);
   std::swap(ctx->pos, ctx->pos2); std::swap(ctx->vx, ctx->vx2); std::swap(ctx->vy, ctx->vy2); std::swap(ctx->vz, ctx->vz2);
   std::swap(ctx->species, ctx->species2); std::swap(ctx->group, ctx->group2); std::swap(ctx->gid, ctx->gid2); std::swap(ctx->orig, ctx->orig2);
   /* 2. periodic image atoms */
   hipLaunchKernelGGL(k_count_images, dim3(nb), dim3(256), 0, st, gp, n, ctx->pos.p, ctx->nimg.p);
   HIPCHK(ctx, hipMemcpyAsync(ctx->img_off.p, ctx->nimg.p, n * sizeof(int), hipMemcpyDeviceToDevice, st));
   if ((rc = ddcmi_scan_exclusive(ctx, ctx->img_off.p, n, ctx->d_flags + 8))) return rc;
   HIPCHK(ctx, hipMemcpyAsync(ctx->h_flags + 8, ctx->d_flags + 8, sizeof(int), hipMemcpyDeviceToHost, st));
   HIPCHK(ctx, hipStreamSynchronize(st));
   int nh = ctx->h_flags[8];
   ctx->nhalo = nh;
   if (nh > 0)
   {
      dbuf<int> *hb[] = {&ctx->hsrc_t, &ctx->hshift_t, &ctx->hcid, &ctx->hrank, &ctx->horder, &ctx->halo_src, &ctx->halo_shift};
      for (auto b : hb) ENSURE(ctx, *b, nh + 1);
      if ((size_t)(n + nh) > ctx->pos.cap)
      {
         if (ctx->pos.ensure(n + nh, true, st) || ctx->pos2.ensure(n + nh) || ctx->qatom.ensure(n + nh) || ctx->gid.ensure(n + nh, true, st) || ctx->gid2.ensure(n + nh))
            SETERR(ctx, DDCMI_ENOMEM, "growing particle arrays for %d image atoms failed", nh);
      }
      int nhb = cdiv(nh, 256);
      hipLaunchKernelGGL(k_fill_images, dim3(nb), dim3(256), 0, st, gp, n, ctx->pos.p, ctx->img_off.p, ctx->hsrc_t.p, ctx->hshift_t.p, ctx->hcid.p, ctx->hrank.p, ctx->cell_cnt_h.p);
      HIPCHK(ctx, hipMemcpyAsync(ctx->cell_start_h.p, ctx->cell_cnt_h.p, ncell * sizeof(int), hipMemcpyDeviceToDevice, st));
      if ((rc = ddcmi_scan_exclusive(ctx, ctx->cell_start_h.p, ncell, nullptr))) return rc;
      hipLaunchKernelGGL(k_scatter_order, dim3(nhb), dim3(256), 0, st, nh, ctx->hcid.p, ctx->hrank.p, ctx->cell_start_h.p, ctx->horder.p);
      hipLaunchKernelGGL(k_sort_cells, dim3(ncb), dim3(256), 0, st, ncell, ctx->cell_start_h.p, ctx->cell_cnt_h.p, ctx->horder.p);
      hipLaunchKernelGGL(k_gather_halo, dim3(nhb), dim3(256), 0, st, nh, ctx->horder.p, ctx->hsrc_t.p, ctx->hshift_t.p, ctx->halo_src.p, ctx->halo_shift.p);
      hipLaunchKernelGGL(k_halo_update, dim3(nhb), dim3(256), 0, st, n, nh, ctx->halo_src.p, ctx->halo_shift.p, gp.L[0], gp.L[1], gp.L[2], ctx->pos.p, ctx->gid.p, true);
   }
   else HIPCHK(ctx, hipMemsetAsync(ctx->cell_start_h.p, 0, ncell * sizeof(int), st));
   hipLaunchKernelGGL(k_merge_cells, dim3(ncb), dim3(256), 0, st, ncell, n, ctx->cell_cnt_o.p, ctx->cell_start_o.p, ctx->cell_cnt_h.p, ctx->cell_start_h.p, ctx->cell_start.p, ctx->cell_cnt.p);
   hipLaunchKernelGGL(k_fill_q, dim3(cdiv(n + nh, 256)), dim3(256), 0, st, n + nh, ctx->pos.p, ctx->d_charge_sp.p, ctx->qatom.p);
   /* 3. full neighbour list (ELL, slot-major) */
   ctx->npad = cdiv(n, DDCMI_BLOCK) * DDCMI_BLOCK;
   if (ctx->maxnbr == 0)
   {
      double vol = gp.L[0] * gp.L[1] * gp.L[2];
      double expect = 4.0 / 3.0 * M_PI * gp.rlist * gp.rlist * gp.rlist * (double)n / vol;
      ctx->maxnbr = ((int)(expect * 1.25) + 24 + 7) & ~7;
      ctx->maxexcl = ctx->nmoltype > 0 ? 8 : 1;
      bool multi = false;
      for (int m = 0; m < ctx->nmoltype; m++) if (ctx->mol_nspecies[m] > 1) multi = true;
      if (multi) ctx->maxexcl = 16;
   }
   ENSURE(ctx, ctx->nbr_cnt, ctx->npad); ENSURE(ctx, ctx->excl_cnt, ctx->npad);
   for (int attempt = 0; attempt < 8; attempt++)
   {
      ENSURE(ctx, ctx->nbr, (size_t)ctx->maxnbr * ctx->npad);
      ENSURE(ctx, ctx->excl, (size_t)ctx->maxexcl * ctx->npad);
      HIPCHK(ctx, hipMemsetAsync(ctx->d_flags, 0, 8 * sizeof(int), st));
      hipLaunchKernelGGL(k_build_list, dim3(nb), dim3(DDCMI_BLOCK), 0, st, gp, n, ctx->npad, ctx->pos.p, ctx->gid.p, ctx->species.p,
                         ctx->cell_start.p, ctx->cell_cnt.p, ctx->nmoltype, ctx->d_moltype_sp.p, ctx->d_mol_nspecies.p, ctx->d_bpair_off.p,
                         ctx->d_bpairI.p, ctx->d_bpairJ.p, ctx->maxnbr, ctx->nbr.p, ctx->nbr_cnt.p, ctx->maxexcl, ctx->excl.p, ctx->excl_cnt.p, ctx->d_flags);
      HIPCHK(ctx, hipMemcpyAsync(ctx->h_flags, ctx->d_flags, 8 * sizeof(int), hipMemcpyDeviceToHost, st));
      HIPCHK(ctx, hipStreamSynchronize(st));
      if (ctx->h_flags[0] == 0 && ctx->h_flags[1] == 0) break;
      if (ctx->h_flags[0]) ctx->maxnbr = ((int)(ctx->h_flags[0] * 1.15) + 8 + 7) & ~7;
      if (ctx->h_flags[1]) ctx->maxexcl = ctx->h_flags[1] + 4;
      if (attempt == 7) SETERR(ctx, DDCMI_ENOMEM, "neighbour list capacity could not be settled");
   }
   /* statistics */
   {
      unsigned long long *d_tot = (unsigned long long *)(ctx->d_results + R_FLAGS);
      HIPCHK(ctx, hipMemsetAsync(d_tot, 0, 2 * sizeof(unsigned long long), st));
      hipLaunchKernelGGL(k_sum_counts, dim3(nb), dim3(256), 0, st, n, ctx->nbr_cnt.p, ctx->excl_cnt.p, d_tot);
      unsigned long long tot[2];
      HIPCHK(ctx, hipMemcpyAsync(tot, d_tot, sizeof(tot), hipMemcpyDeviceToHost, st));
      HIPCHK(ctx, hipStreamSynchronize(st));
      ctx->list_entries = (int64_t)tot[0]; ctx->excl_entries = (int64_t)tot[1];
   }
   int nblk = cdiv(n, DDCMI_BLOCK);
   ENSURE(ctx, ctx->partials, (size_t)(nblk + 8) * 8);
   ctx->list_valid = true;
   ctx->nrebuild++;
   return DDCMI_OK;
}

/* ------------------------------------------------------------------------- */
static int launch_forces(ddcmi_ctx *ctx)
{
   hipStream_t st = ctx->stream;
   int n = ctx->nloc, nh = ctx->nhalo;
   int nblk = cdiv(n, DDCMI_BLOCK);
   if (nh > 0)
      hipLaunchKernelGGL(k_halo_update, dim3(cdiv(nh, 256)), dim3(256), 0, st, n, nh, ctx->halo_src.p, ctx->halo_shift.p,
                         ctx->gp.L[0], ctx->gp.L[1], ctx->gp.L[2], ctx->pos.p, ctx->gid.p, false);
   HIPCHK(ctx, hipMemsetAsync(ctx->d_results, 0, R_RK * sizeof(double), st));
   if ((ctx->excludePotentialTerm & 128) == 0)
   {
      int grid = ((nblk + 7) / 8) * 8;
      size_t lds = (size_t)ctx->nlj * ctx->nlj * sizeof(double4);
      hipEvent_t e0 = nullptr, e1 = nullptr;
      if (ctx->timing)
      {
         if (ctx->ev_used + 2 > ctx->ev.size())
         {
            size_t old = ctx->ev.size();
            ctx->ev.resize(old + 256);
            for (size_t k = old; k < ctx->ev.size(); k++) HIPCHK(ctx, hipEventCreate(&ctx->ev[k]));
         }
         e0 = ctx->ev[ctx->ev_used++]; e1 = ctx->ev[ctx->ev_used++];
         HIPCHK(ctx, hipEventRecord(e0, st));
      }
      bool useq = ctx->has_charge;
      if (useq)
         hipLaunchKernelGGL(k_nonbond<true>, dim3(grid), dim3(DDCMI_BLOCK), lds, st, n, ctx->npad, nblk, ctx->pos.p, ctx->qatom.p, ctx->nbr.p, ctx->nbr_cnt.p,
                            ctx->excl.p, ctx->excl_cnt.p, ctx->d_ljtab.p, ctx->nlj, ctx->rmax * ctx->rmax, ctx->krf, ctx->crf, ctx->keR,
                            ctx->fx.p, ctx->fy.p, ctx->fz.p, ctx->partials.p);
      else
         hipLaunchKernelGGL(k_nonbond<false>, dim3(grid), dim3(DDCMI_BLOCK), lds, st, n, ctx->npad, nblk, ctx->pos.p, ctx->qatom.p, ctx->nbr.p, ctx->nbr_cnt.p,
                            ctx->excl.p, ctx->excl_cnt.p, ctx->d_ljtab.p, ctx->nlj, ctx->rmax * ctx->rmax, ctx->krf, ctx->crf, ctx->keR,
                            ctx->fx.p, ctx->fy.p, ctx->fz.p, ctx->partials.p);
      if (ctx->timing) { HIPCHK(ctx, hipEventRecord(e1, st)); ctx->t_launches++; }
      hipLaunchKernelGGL(k_reduce_partials, dim3(1), dim3(DDCMI_BLOCK), 0, st, ctx->partials.p, nblk, 8, 8, ctx->d_results + R_NB_LJ);
   }
   else
      hipLaunchKernelGGL(k_zero3, dim3(cdiv(n, 256)), dim3(256), 0, st, n, ctx->fx.p, ctx->fy.p, ctx->fz.p);
   int rc = ddcmi_launch_bonded(ctx);
   if (rc) return rc;
   double self = ((ctx->excludePotentialTerm & 128) == 0) ? ctx->self_ele : 0.0;
   hipLaunchKernelGGL(k_finish_energy, dim3(1), dim3(64), 0, st, ctx->d_results, self);
   ctx->forces_valid = true;
   return DDCMI_OK;
}

static int fetch_results(ddcmi_ctx *ctx)
{
   HIPCHK(ctx, hipMemcpyAsync(ctx->h_results, ctx->d_results, R_SIZE * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
   HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
   return DDCMI_OK;
}

extern "C" int ddcmi_eval_forces(ddcmi_ctx *ctx, double *energies, double *virial)
{
   if (!ctx) return DDCMI_EINVAL;
   (void)hipSetDevice(ctx->device);
   int rc;
   if (!ctx->list_valid && (rc = ddcmi_build_list(ctx))) return rc;
   if ((rc = launch_forces(ctx))) return rc;
   if ((rc = fetch_results(ctx))) return rc;
   if (energies) for (int k = 0; k < DDCMI_NE; k++) energies[k] = ctx->h_results[R_E + k];
   if (virial) for (int k = 0; k < 6; k++) virial[k] = ctx->h_results[R_VIR + k];
   return DDCMI_OK;
}

static int launch_kinetic(ddcmi_ctx *ctx, double dt, int do_kick)
{
   int n = ctx->nloc, nblk = cdiv(n, DDCMI_BLOCK);
   ENSURE(ctx, ctx->partials, (size_t)(nblk + 8) * 8);
   hipLaunchKernelGGL(k_kick_ke, dim3(nblk), dim3(DDCMI_BLOCK), 0, ctx->stream, n, dt, ctx->d_invmass.p, ctx->d_mass.p, ctx->species.p,
                      ctx->fx.p, ctx->fy.p, ctx->fz.p, ctx->vx.p, ctx->vy.p, ctx->vz.p, ctx->partials.p, do_kick);
   hipLaunchKernelGGL(k_reduce_partials, dim3(1), dim3(DDCMI_BLOCK), 0, ctx->stream, ctx->partials.p, nblk, 8, 7, ctx->d_results + R_RK);
   return DDCMI_OK;
}

/* berendsen_Update FRONT_TIMESTEP (berendsen.c:30-62), host scalar logic */
static void berendsen_update(ddcmi_ctx *ctx, double dt_half)
{
   for (int g = 0; g < ctx->ngroup; g++)
   {
      if (ctx->gtype[g] != DDCMI_BERENDSEN) continue;
      ctx->gTsum[g] += ctx->gT[g];
      ctx->gnT[g] += 1;
      double Tave = ctx->gTsum[g] / ctx->gnT[g];
      double ratio = (Tave == 0) ? 0 : ctx->gTeq[g] / Tave;
      if (ctx->gtau[g] != 0) ctx->glambda[g] = sqrt(1 + (2.0 * dt_half / ctx->gtau[g]) * (ratio - 1));
      else ctx->glambda[g] = sqrt(ratio);
      ctx->gdoScaling[g] = 0;
      if (ctx->loop % ctx->ginterval[g] == 0) { ctx->gTsum[g] = 0; ctx->gnT[g] = 0; ctx->gdoScaling[g] = 1; }
   }
}

extern "C" int ddcmi_step_nglf(ddcmi_ctx *ctx, double dt, int nsteps)
{
   if (!ctx || nsteps < 0) return DDCMI_EINVAL;
   if (!ctx->forces_valid) SETERR(ctx, DDCMI_EINVAL, "ddcmi_step_nglf needs forces: call ddcmi_eval_forces first (firstEnergyCall, masters.c:579)");
   (void)hipSetDevice(ctx->device);
   hipStream_t st = ctx->stream;
   int n = ctx->nloc, nb = cdiv(n, 256), rc;
   GroupLambda lam;
   for (int g = 0; g < 32; g++) lam.v[g] = 1.0;
   for (int s = 0; s < nsteps; s++)
   {
      /* lambda applies at the FRONT kick when doScaling is set (berendsen.c:74-80) */
      for (int g = 0; g < ctx->ngroup; g++) lam.v[g] = (ctx->gtype[g] == DDCMI_BERENDSEN && ctx->gdoScaling[g]) ? ctx->glambda[g] : 1.0;
      hipLaunchKernelGGL(k_kick_drift, dim3(nb), dim3(256), 0, st, n, dt, ctx->d_invmass.p, ctx->species.p, ctx->group.p, lam,
                         ctx->fx.p, ctx->fy.p, ctx->fz.p, ctx->vx.p, ctx->vy.p, ctx->vz.p, ctx->pos.p);
      ctx->time += dt;
      ctx->loop += 1;
      /* ddcUpdateAll.c:64-71: rebuild when loop % updateRate == 0 */
      if (ctx->loop % ctx->updateRate == 0 || !ctx->list_valid)
      {
         if ((rc = ddcmi_build_list(ctx))) return rc;
         n = ctx->nloc; nb = cdiv(n, 256);
      }
      if ((rc = launch_forces(ctx))) return rc;
      if ((rc = launch_kinetic(ctx, dt, 1))) return rc;
      berendsen_update(ctx, 0.5 * dt);
   }
   return DDCMI_OK;
}

extern "C" int ddcmi_get_energies(ddcmi_ctx *ctx, double *energies, double *virial, double *rk, double *tion)
{
   if (!ctx) return DDCMI_EINVAL;
   (void)hipSetDevice(ctx->device);
   int rc = fetch_results(ctx);
   if (rc) return rc;
   if (energies) for (int k = 0; k < DDCMI_NE; k++) energies[k] = ctx->h_results[R_E + k];
   if (virial) for (int k = 0; k < 6; k++) virial[k] = ctx->h_results[R_VIR + k];
   if (rk) *rk = ctx->h_results[R_RK];
   if (tion) for (int k = 0; k < 6; k++) tion[k] = ctx->h_results[R_TION + k];
   return DDCMI_OK;
}

extern "C" int ddcmi_kinetic(ddcmi_ctx *ctx, double *rk, double *tion)
{
   if (!ctx || ctx->nloc <= 0) return DDCMI_EINVAL;
   (void)hipSetDevice(ctx->device);
   int rc = launch_kinetic(ctx, 0.0, 0);
   if (rc) return rc;
   return ddcmi_get_energies(ctx, nullptr, nullptr, rk, tion);
}

extern "C" int ddcmi_group_temperatures(ddcmi_ctx *ctx, double *Tgroup)
{
   if (!ctx || ctx->nloc <= 0) return DDCMI_EINVAL;
   (void)hipSetDevice(ctx->device);
   hipStream_t st = ctx->stream;
   int n = ctx->nloc, ng = ctx->ngroup;
   HIPCHK(ctx, hipMemsetAsync(ctx->d_results + R_GROUP, 0, 2 * ng * sizeof(double), st));
   hipLaunchKernelGGL(k_group_ke, dim3(cdiv(n, DDCMI_BLOCK)), dim3(DDCMI_BLOCK), 0, st, n, ng, ctx->d_mass.p, ctx->species.p, ctx->group.p,
                      ctx->vx.p, ctx->vy.p, ctx->vz.p, ctx->d_results + R_GROUP);
   int rc = fetch_results(ctx);
   if (rc) return rc;
   for (int g = 0; g < ng; g++)
   {
      double rk = ctx->h_results[R_GROUP + 2 * g], num = ctx->h_results[R_GROUP + 2 * g + 1];
      if (num > 0.0) ctx->gT[g] = 2.0 * rk / (3.0 * num);     /* energyInfo.c:139 */
      if (Tgroup) Tgroup[g] = ctx->gT[g];
   }
   return DDCMI_OK;
}

/* ------------------------------------------------------------------------- */
extern "C" int ddcmi_list_stats(const ddcmi_ctx *ctx, int64_t stats[8])
{
   if (!ctx || !stats) return DDCMI_EINVAL;
   stats[0] = ctx->list_entries; stats[1] = ctx->excl_entries; stats[2] = ctx->maxnbr; stats[3] = ctx->nhalo;
   stats[4] = ctx->gp.ncell; stats[5] = ctx->nrebuild; stats[6] = ctx->npad; stats[7] = 0;
   return DDCMI_OK;
}

extern "C" int ddcmi_get_list(ddcmi_ctx *ctx, int which, int *start, int *j, int64_t *nentries)
{
   if (!ctx || !ctx->list_valid || which < 0 || which > 1) return DDCMI_EINVAL;
   (void)hipSetDevice(ctx->device);
   hipStream_t st = ctx->stream;
   int n = ctx->nloc, nb = cdiv(n, 256);
   const int *lst = which ? ctx->excl.p : ctx->nbr.p;
   const int *cnt = which ? ctx->excl_cnt.p : ctx->nbr_cnt.p;
   int64_t tot = which ? ctx->excl_entries : ctx->list_entries;
   if (nentries) *nentries = tot;
   if (!start) return DDCMI_OK;
   std::vector<int> c(n);
   dbuf<int> d_start, d_j;
   if (d_start.ensure(n + 1)) SETERR(ctx, DDCMI_ENOMEM, "get_list alloc");
   hipLaunchKernelGGL(k_counts_by_orig, dim3(nb), dim3(256), 0, st, n, cnt, ctx->orig.p, d_start.p);
   HIPCHK(ctx, hipMemcpyAsync(c.data(), d_start.p, n * sizeof(int), hipMemcpyDeviceToHost, st));
   HIPCHK(ctx, hipStreamSynchronize(st));
   start[0] = 0;
   for (int i = 0; i < n; i++) start[i + 1] = start[i] + c[i];
   if (j && tot > 0)
   {
      if (d_j.ensure(tot)) { d_start.release(); SETERR(ctx, DDCMI_ENOMEM, "get_list alloc"); }
      HIPCHK(ctx, hipMemcpyAsync(d_start.p, start, (n + 1) * sizeof(int), hipMemcpyHostToDevice, st));
      hipLaunchKernelGGL(k_list_to_csr, dim3(nb), dim3(256), 0, st, n, ctx->npad, lst, cnt, ctx->orig.p, ctx->halo_src.p, d_start.p, d_j.p);
      HIPCHK(ctx, hipMemcpyAsync(j, d_j.p, tot * sizeof(int), hipMemcpyDeviceToHost, st));
      HIPCHK(ctx, hipStreamSynchronize(st));
   }
   d_start.release(); d_j.release();
   return DDCMI_OK;
}

extern "C" int ddcmi_timing_enable(ddcmi_ctx *ctx, int on)
{
   if (!ctx) return DDCMI_EINVAL;
   ctx->timing = on != 0;
   return DDCMI_OK;
}
extern "C" int ddcmi_timing_read(ddcmi_ctx *ctx, int64_t *launches, double *total_ms, int reset)
{
   if (!ctx) return DDCMI_EINVAL;
   (void)hipSetDevice(ctx->device);
   HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
   for (size_t k = 0; k + 1 < ctx->ev_used; k += 2)
   {
      float ms = 0;
      HIPCHK(ctx, hipEventElapsedTime(&ms, ctx->ev[k], ctx->ev[k + 1]));
      ctx->t_ms += ms;
   }
   ctx->ev_used = 0;
   if (launches) *launches = ctx->t_launches;
   if (total_ms) *total_ms = ctx->t_ms;
   if (reset) { ctx->t_launches = 0; ctx->t_ms = 0; }
   return DDCMI_OK;
}
