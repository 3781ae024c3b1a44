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
#include <functional>
#include <rccl/rccl.h>

static std::string g_create_err;

/* hipFuncSetAttribute(MaxDynamicSharedMemorySize) costs tens of microseconds of host time a call: the limit of a kernel is
 * raised once per (device, kernel) and only ever upwards */
#include <map>
#include <mutex>
static hipError_t dyn_lds_limit(int device, const void *fn, int bytes)
{
   static std::mutex mu;
   static std::map<std::pair<int, const void *>, int> have;
   std::lock_guard<std::mutex> lk(mu);
   int &h = have[std::make_pair(device, fn)];
   if (bytes <= h) return hipSuccess;
   hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
   if (e == hipSuccess) h = bytes;
   return e;
}

/* k_nonbond addresses its staged beads by raw LDS byte offsets from LDS address 0 (a list entry's slot bits ARE the gather's
 * address): that holds while the kernel has no static LDS, so that its dynamic region starts at 0.  Asked of the runtime once per
 * (device, kernel) -- a compiler or runtime update that adds static LDS makes every launch return an error with this message
 * (the kernel used to check on the device and __builtin_trap(): a process abort with no message). */
static bool lds_starts_at_zero(int device, const void *fn, size_t *static_bytes)
{
   static std::mutex mu;
   static std::map<std::pair<int, const void *>, size_t> seen;
   std::lock_guard<std::mutex> lk(mu);
   auto it = seen.find(std::make_pair(device, fn));
   if (it == seen.end())
   {
      hipFuncAttributes at;
      memset(&at, 0, sizeof(at));
      const size_t sb = hipFuncGetAttributes(&at, fn) == hipSuccess ? at.sharedSizeBytes : 0;
      it = seen.emplace(std::make_pair(device, fn), sb).first;
   }
   *static_bytes = it->second;
   return it->second == 0;
}

/* ------------------------------------------------------------------------- */
/* small device helpers                                                       */
__device__ __forceinline__ int cell_linear(const GridParams &gp, int cx, int cy, int cz)
{
   int tx = cx / TCX, ty = cy / TCY, tz = cz / TCZ;
   int lx = cx - tx * TCX, ly = cy - ty * TCY, lz = cz - tz * TCZ;
   return (((tz * gp.T[1] + ty) * gp.T[0]) + tx) * TCELLS + (lz * TCY + ly) * TCX + lx;
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
/* cell of an image/halo bead.  Computed from the position, then forced onto the correct
 * side per axis (side[a]: +1 = beyond the hi face, -1 = below the lo face, 0 = inside the
 * interior range): a bead sitting exactly on a face must never land in an interior cell,
 * where the owned beads' ranges live. */
__device__ __forceinline__ int halo_cell(const GridParams &gp, double x, double y, double z, const int side[3])
{
   double r[3] = {x, y, z};
   int c[3];
#pragma unroll
   for (int a = 0; a < 3; a++)
   {
      int ic = (int)floor((r[a] - gp.lo[a]) * gp.cinv[a]);
      if (side[a] > 0) ic = max(ic, gp.n[a]);
      else if (side[a] < 0) ic = min(ic, -1);
      else ic = min(max(ic, 0), gp.n[a] - 1);
      ic += gp.m[a];
      c[a] = min(max(ic, 0), gp.g[a] - 1);
   }
   return cell_linear(gp, c[0], c[1], c[2]);
}
__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
   for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
   return v;
}
/* the same sum without the LDS crossbar (__shfl_* is ds_bpermute: two per step and double, in the pair kernel they queue behind
 * the gathers of every other wave of the CU): butterflies inside the rows of 16 lanes by DPP -- quad_perm [1,0,3,2], [2,3,0,1],
 * row_half_mirror, row_mirror: after the four every lane holds its row's sum -- then the four rows by v_readlane.  Fixed order. */
template <int CTRL>
__device__ __forceinline__ double dpp_move(double v)
{
   const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, false);
   const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, false);
   return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_sum_dpp(double v)
{
   v += dpp_move<0xB1>(v);
   v += dpp_move<0x4E>(v);
   v += dpp_move<0x141>(v);
   v += dpp_move<0x140>(v);
   double r[4];
#pragma unroll
   for (int q = 0; q < 4; q++)
      r[q] = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 16 * q), __builtin_amdgcn_readlane(__double2loint(v), 16 * q));
   return (r[0] + r[1]) + (r[2] + r[3]);
}
/* the first half of it: every lane gets the sum over its ROW of 16 lanes (four DPP butterflies, no readlane) */
__device__ __forceinline__ double row_sum_dpp(double v)
{
   v += dpp_move<0xB1>(v);
   v += dpp_move<0x4E>(v);
   v += dpp_move<0x141>(v);
   v += dpp_move<0x140>(v);
   return v;
}
__device__ __forceinline__ int wave_max_dpp(int v)      /* every lane gets the maximum over the wave (v >= 0) */
{
   v = max(v, __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xf, 0xf, false));
   v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xf, 0xf, false));
   v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x141, 0xf, 0xf, false));
   v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x140, 0xf, 0xf, false));
   return max(max(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)), max(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48)));
}
__device__ __forceinline__ int wave_scan_inclusive_dpp(int v)      /* prefix sum over the 64 lanes: row_shr 1, 2, 4, 8 inside the rows (a lane without a source adds 0), then the totals of the rows before */
{
   v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);
   v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);
   v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);
   v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);
   const int t0 = __builtin_amdgcn_readlane(v, 15), t1 = __builtin_amdgcn_readlane(v, 31), t2 = __builtin_amdgcn_readlane(v, 47);
   const int row = (int)(threadIdx.x & 63) >> 4;
   return v + (row >= 1 ? t0 : 0) + (row >= 2 ? t1 : 0) + (row >= 3 ? t2 : 0);
}
/* block (NW waves) reduction of NV values per thread into out[NV], fixed order */
template <int NV, int NW = 4>
__device__ __forceinline__ void block_reduce_store(double (&v)[NV], double *out)
{
   __shared__ double s_red[NW][NV];
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
      double a = s_red[0][k];
#pragma unroll
      for (int q = 1; q < NW; q++) a += s_red[q][k];
      out[k] = a;
   }
}

/* backInBox_fast: PreduceOrthorhombicB7_OneLatticeReduction (preduce.c:147-160) */
__device__ __forceinline__ void back_in_box(const GridParams &gp, double4 &p)
{
   if (gp.pbc & 1) { if (p.x > 0.5 * gp.L[0]) p.x -= gp.L[0]; if (p.x < -0.5 * gp.L[0]) p.x += gp.L[0]; }
   if (gp.pbc & 2) { if (p.y > 0.5 * gp.L[1]) p.y -= gp.L[1]; if (p.y < -0.5 * gp.L[1]) p.y += gp.L[1]; }
   if (gp.pbc & 4) { if (p.z > 0.5 * gp.L[2]) p.z -= gp.L[2]; if (p.z < -0.5 * gp.L[2]) p.z += gp.L[2]; }
}
/* ------------------------------------------------------------------------- */
/* sort: wrap + cell id + in-cell rank                                        */
__global__ void k_wrap_cell(GridParams gp, int nloc, const double4 *pos, int *cid, int *rank, int *cell_cnt, int *runaway, int *runaway2, int *renumber = nullptr, int *zero28 = nullptr)
{
   int i = blockIdx.x * blockDim.x + threadIdx.x;
   int c = -1;
   if (zero28 && i < 28) zero28[i] = 0;      /* decomposed runs: the direction counters of the halo selection that follows the sort (the migration round has read them) */
   if (i < nloc)
   {
      if (renumber) renumber[i] = i;      /* decomposed runs: orig = the bead's place in front of this sort */
      double4 p = pos[i];
      /* a bead that is not a number, or more than a box length outside the box: the run has blown up (the
       * reference would abort in its domain assignment); reported at this rebuild instead of as a full cell */
      if (!(((gp.pbc & 1) ? fabs(p.x) < 1.5 * gp.L[0] : fabs(p.x) < 1e300) && ((gp.pbc & 2) ? fabs(p.y) < 1.5 * gp.L[1] : fabs(p.y) < 1e300) &&
            ((gp.pbc & 4) ? fabs(p.z) < 1.5 * gp.L[2] : fabs(p.z) < 1e300))) { atomicAdd(runaway, 1); if (runaway2) atomicAdd(runaway2, 1); }
      /* backInBox_fast (nglf.c:90).  The wrapped record is not written back here: k_gather_state, which moves every record
       * anyway, wraps it again the same way (128 MB less to write at 4 M beads) */
      back_in_box(gp, p);
      int cx, cy, cz;
      cell_coords(gp, p.x, p.y, p.z, true, cx, cy, cz);
      c = cell_linear(gp, cx, cy, cz);
      cid[i] = c;
   }
   /* the beads arrive in the order of the previous sort, so a wave holds runs of beads of one cell: one
    * atomic per run instead of one per bead (the final in-cell order is fixed by k_sort_cells) */
   const int lane = threadIdx.x & 63;
   const int cprev = __shfl_up(c, 1, 64);
   const bool head = lane == 0 || cprev != c;
   const unsigned long long hb = __ballot(head);
   const unsigned long long upto = lane == 63 ? ~0ull : ((2ull << lane) - 1ull);
   const int hl = 63 - __clzll((long long)(hb & upto));
   const unsigned long long after = hl == 63 ? 0ull : (hb & ~((2ull << hl) - 1ull));
   const int len = (after ? __ffsll((long long)after) - 1 : 64) - hl;
   int base = 0;
   if (head && c >= 0) base = atomicAdd(&cell_cnt[c], len);
   base = __shfl(base, hl, 64);
   if (c >= 0) rank[i] = base + (lane - hl);
}
__global__ void k_scatter_order(int n, const int *cid, const int *rank, const int *cell_start, int *order, const int *n_dev = nullptr /* the count, where the host only knows a bound */)
{
   int i = blockIdx.x * blockDim.x + threadIdx.x;
   if (n_dev) n = min(n, *n_dev);
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
/* the same by a key that does not depend on how the beads arrived (decomposed runs: migrants and halo beads land in the order
 * atomics and messages deliver them): the bead's gid, then -- copies of one bead in a halo -- its shift code, then the index */
__global__ void k_sort_cells_key(int ncell, const int *cell_start, const int *cell_cnt, int *order, const uint64_t *__restrict__ key, const int *__restrict__ key2)
{
   int c = blockIdx.x * blockDim.x + threadIdx.x;
   if (c >= ncell) return;
   int s = cell_start[c], n = cell_cnt[c];
   auto before = [&](int a, int b)      /* does a come before b? */
   {
      const uint64_t ka = key[a], kb = key[b];
      if (ka != kb) return ka < kb;
      if (key2) { const int sa = key2[a], sb = key2[b]; if (sa != sb) return sa < sb; }
      return a < b;
   };
   for (int a = 1; a < n; a++)
   {
      int v = order[s + a];
      int b = a - 1;
      while (b >= 0 && before(v, order[s + b])) { order[s + b + 1] = order[s + b]; b--; }
      order[s + b + 1] = v;
   }
}
__device__ __forceinline__ void image_dirs(const GridParams &gp, const double4 &p, int d[3]);
/* nimg (one domain): the number of periodic self-images of the bead, counted here because the record is in registers anyway
 * (a pass of its own read all positions again: 33 us at 4 M beads) */
__global__ void k_gather_state(int nloc, const int *order,
                               const double4 *pos, const double *vx, const double *vy, const double *vz,
                               const int *species, const int *group, const uint64_t *gid, const int *orig,
                               double4 *pos2, double *vx2, double *vy2, double *vz2,
                               int *species2, int *group2, uint64_t *gid2, int *orig2, int *slot_of_orig, GridParams gp, int *nimg, int wrap,
                               const ulonglong2 *lcg, ulonglong2 *lcg2 /* the beads' LCG64 records, nullptr without them */)
{
   int k = blockIdx.x * blockDim.x + threadIdx.x;
   if (k >= nloc) return;
   int i = order[k];
   double4 p = pos[i];
   if (wrap) back_in_box(gp, p);      /* (k_wrap_cell sorted the beads by their wrapped positions) */
   if (nimg)
   {
      int d[3];
      image_dirs(gp, p, d);
      nimg[k] = (1 + (d[0] != 0)) * (1 + (d[1] != 0)) * (1 + (d[2] != 0)) - 1;
   }
   pos2[k] = p;
   vx2[k] = vx[i]; vy2[k] = vy[i]; vz2[k] = vz[i];
   species2[k] = species[i]; group2[k] = group[i]; gid2[k] = gid[i];
   int o = orig[i];
   orig2[k] = o;
   if (slot_of_orig) slot_of_orig[o] = k;      /* (a scattered store per bead: only where something names beads by caller index) */
   if (lcg) lcg2[k] = lcg[i];
}

__global__ void k_slots_from_orig(int nloc, const int *__restrict__ orig, int *slot_of_orig)
{
   int k = blockIdx.x * blockDim.x + threadIdx.x;
   if (k < nloc) slot_of_orig[orig[k]] = k;
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
__global__ void k_fill_images(GridParams gp, int nloc, const double4 *pos, const int *img_off, const int *nimg,
                              int *hsrc, int *hshift, int *hcid, int *hrank, int *cell_cnt_h, int cap /* images beyond it are dropped: the host sees the count and starts over */)
{
   int i = blockIdx.x * blockDim.x + threadIdx.x;
   if (i >= nloc || nimg[i] == 0) return;      /* three beads in four have no image: their records are not read */
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
            int side[3] = {ix, iy, iz};
            int c = halo_cell(gp, x, y, z, side);
            if (k >= cap) return;
            hsrc[k] = i;
            hshift[k] = (ix + 1) + 3 * (iy + 1) + 9 * (iz + 1);
            hcid[k] = c;
            hrank[k] = atomicAdd(&cell_cnt_h[c], 1);
            k++;
         }
}
__global__ void k_gather_halo(int nhalo, const int *horder, const int *hsrc_t, const int *hshift_t, int *halo_src, int *halo_shift, const int *n_dev)
{
   int h = blockIdx.x * blockDim.x + threadIdx.x;
   if (n_dev) nhalo = min(nhalo, *n_dev);
   if (h >= nhalo) return;
   int k = horder[h];
   halo_src[h] = hsrc_t[k];
   halo_shift[h] = hshift_t[k];
}
/* refresh image/halo beads every step (replaces ddcUpdate's position halo,
 * ddcUpdate.c:40-85): src >= 0 -> periodic self-image of owned bead src;
 * src < 0 -> bead -1-src of the buffer received from a neighbour domain (the sender
 * has already applied the periodic shift). */
/* hmax (decomposed runs whose pair kernel ends its rows early, NbTileArgs::hdisp): the largest squared distance of a RECEIVED bead
 * from where it lay when the list was built -- hrecv5 still holds the rebuild's records -- goes to hmax[par] (bit pattern of a
 * non-negative double, atomic max: one per workgroup of HU_PER beads); the word of the other parity, which the next step uses, is
 * zeroed here.  Self-images move with their owned source: the owned beads' bound covers them. */
#define HU_THREADS 256
#define HU_PER 1024
__global__ __launch_bounds__(HU_THREADS) void k_halo_update(int nloc, int nhalo, const int *halo_src, const int *halo_shift, double L0, double L1, double L2,
                              double4 *pos, uint64_t *gid, bool with_tags, const double *hrecv3, const double *hrecv5, const int *n_dev = nullptr,
                              unsigned long long *hmax = nullptr, int par = 0,
                              const int *horder = nullptr, const int *hsrc_t = nullptr, const int *hshift_t = nullptr, int *halo_src_w = nullptr, int *halo_shift_w = nullptr
                              /* rebuild: the sorted descriptors are gathered here (horder: sorted place -> descriptor) and written for the steps to come */)
{
   if (n_dev) nhalo = min(nhalo, *n_dev);
   double d2max = 0.0;
   const int hend = min(nhalo, ((int)blockIdx.x + 1) * HU_PER);
   for (int h = blockIdx.x * HU_PER + threadIdx.x; h < hend; h += HU_THREADS)
   {
      int s, code0 = 0;
      if (horder) { const int kd = horder[h]; s = hsrc_t[kd]; code0 = hshift_t[kd]; halo_src_w[h] = s; halo_shift_w[h] = code0; }
      else s = halo_src[h];
      if (s >= 0)
      {
         int code = horder ? code0 : halo_shift[h];
         double4 p = pos[s];
         p.x += (double)(code % 3 - 1) * L0;
         p.y += (double)((code / 3) % 3 - 1) * L1;
         p.z += (double)(code / 9 - 1) * L2;
         pos[nloc + h] = p;
         if (with_tags) gid[nloc + h] = gid[s];
      }
      else
      {
         int k = -1 - s;
         double4 p = pos[nloc + h];
         if (with_tags)
         {
            p.x = hrecv5[5 * k]; p.y = hrecv5[5 * k + 1]; p.z = hrecv5[5 * k + 2];
            p.w = hrecv5[5 * k + 3];
            gid[nloc + h] = (uint64_t)__double_as_longlong(hrecv5[5 * k + 4]);
         }
         else
         {
            p.x = hrecv3[3 * k]; p.y = hrecv3[3 * k + 1]; p.z = hrecv3[3 * k + 2];
            if (hmax)
            {
               const double dx = p.x - hrecv5[5 * k], dy = p.y - hrecv5[5 * k + 1], dz = p.z - hrecv5[5 * k + 2];
               d2max = fmax(d2max, dx * dx + dy * dy + dz * dz);
            }
         }
         pos[nloc + h] = p;
      }
   }
   if (hmax)
   {
      __shared__ double s_m[HU_THREADS / 64];
      for (int off = 32; off > 0; off >>= 1) d2max = fmax(d2max, __shfl_down(d2max, off, 64));
      if ((threadIdx.x & 63) == 0) s_m[threadIdx.x >> 6] = d2max;
      __syncthreads();
      if (threadIdx.x == 0)
      {
         double m = s_m[0];
         for (int w = 1; w < HU_THREADS / 64; w++) m = fmax(m, s_m[w]);
         if (m > 0.0) (void)atomicMax(hmax + par, (unsigned long long)__double_as_longlong(m));
         if (blockIdx.x == 0) hmax[par ^ 1] = 0ull;
      }
   }
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
 * (bioMartini.c:1392-1485) done at build time.
 *
 * One workgroup per TILE (4x4x4 cells, ~250 beads, a compact ~32 A cube).  The
 * tile's neighbourhood -- the 8x8x8 cells within two cells of it, ~2000 beads --
 * is written once as a staging list (global indices, raster order) and loaded
 * into LDS; every owned bead of the tile then scans the 5x5x5 cells around its
 * own cell out of LDS.  List entries are 16-bit indices into the tile's staged
 * set, stored slot-major per tile (ELL) and ordered by distance shell at build
 * time so that late slots are rejected by whole waves. */
template <int NW>
__device__ __forceinline__ int block_excl_scan(int v, int *tot, int *s_w)
{
   int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
   int inc = v;
#pragma unroll
   for (int off = 1; off < 64; off <<= 1)
   {
      int t = __shfl_up(inc, off, 64);
      if (lane >= off) inc += t;
   }
   if (lane == 63) s_w[w] = inc;
   __syncthreads();
   int base = 0, all = 0;
#pragma unroll
   for (int k = 0; k < NW; k++) { if (k < w) base += s_w[k]; all += s_w[k]; }
   *tot = all;
   __syncthreads();
   return base + inc - v;
}

#define TB_THREADS 512      /* k_tile_build workgroup: one lane per owned bead of the tile */
/* accepted words wait in a ring in LDS, [slot][lane], and leave as 16-byte pieces: packed entries (16-bit scratch words) a ring of
 * sixteen 2-byte slots per lane, eight words a piece; bare entries (32-bit words) eight 4-byte slots, four words a piece */
#define TB_RING_BYTES (8 * TB_THREADS * 4)
#define TB_CHUNK 64         /* rows of a scratch chunk = the lanes of the wave that fills it */
#ifndef NSHELL
#define NSHELL 8            /* distance shells of the list order */
#endif
/* shell 0 = r < r0, shells 1..NSHELL-1 = equal steps of r^2 up to the list radius (k_tile_build): boundaries only
 * steer the ORDER of a bead's entries, so single precision is plenty */
struct ShellCuts { float r0sq; int one; };      /* one: no skin, a single shell */
struct TileArgs
{
   int ntile, stage_stride, cap;        /* cap = LDS capacity in staged beads */
   int nloc;                            /* staged indices >= nloc are image/halo beads */
   int pack_type;                       /* entries are (staged slot << 4) | LJ type (nlj <= 16, cap < 4096); else the bare slot.
                                           2: nlj <= 8, bit 3 of the nibble marks a periodically shifted partner */
   const int *halo_shift;               /* halo_shift[j - nloc] != 13: bead j carries a periodic shift */
   const int *cell_start_o;             /* owned beads per cell: exclusive scan, [ncell+1] */
   const int *cell_start, *cell_cnt;    /* merged owned/halo cell ranges */
   int *stage_idx, *tile_nstage;
   long long *tile_base; int *tile_width, *tile_rows, *tile_work;
   unsigned short *nbr16; unsigned long long arena_cap; unsigned long long *arena_used;
   int *nbr_cnt;
   uint4 *nbr_cum;                      /* [bead] eight 16-bit counts: the bead's entries in shells 0..s (k_tile_transpose) -- what k_nonbond walks when later shells cannot matter yet */
   unsigned int *tmp32; int tmpw;       /* scratch list, tmpw words per bead.  Packed entries (pack_type != 0): 16-bit words,
                                           staged slot + 1 | distance shell << 12, and the slot's type nibble in tile_nib; bare entries: 32-bit words, entry | shell << 16.
                                           Layout: the rows of tile t start at row ts + TB_CHUNK t (every tile rounded up to whole chunks); inside a chunk of TB_CHUNK rows the
                                           16-byte piece q of row l lies at (q TB_CHUNK + l) 16 B -- piece-major, so the wave that fills a chunk writes whole cache lines
                                           (row-major rows took 8-byte stores into 64 cache lines per instruction: 3.1 x the bytes at the memory) and the
                                           eight lanes-per-row of k_tile_transpose still read 128 contiguous bytes per 8 rows */
   unsigned char *tile_nib;             /* [ntile][stage_stride] type nibble (+ shifted-copy bit) of every staged slot: k_tile_transpose finishes the entries with it */
   ShellCuts shc;
};

struct NbTileArgs
{
   int ntile, stage_stride, cap, nlj;
   const int *cell_start_o;
   const int *cell_start, *cell_cnt;    /* merged owned/halo cell ranges (k_merge_cells): the staged order follows from them */
   const int *stage_idx, *tile_nstage;
   const long long *tile_base; const int *tile_width, *tile_rows;
   const unsigned short *nbr16;
   const int *nbr_cnt;
   const int *sched;                    /* [9] range of each XCD in perm[] (schedule_tiles) */
   const int *perm;                     /* work items in launch order: tile | part << 24 | (nparts - 1) << 27 (schedule_tiles) */
   const int *tile_work;                /* bit 30: the tile stages image/halo beads */
   const int *halo_shift; int nloc;     /* halo_shift[j - nloc] != 13: bead j carries a periodic shift */
   /* Shells that cannot matter yet.  disp (not null) points at D = sum over the steps since the rebuild of max_i |dt v_i|: no bead has moved
    * further than D, no pair distance has changed by more than 2 D, so an entry that lay in shell s or beyond at the rebuild -- at
    * r^2 >= sh_r0sq + (s - 1) sh_step -- is outside the cut-off while sqrt(that) - 2 D > r_cut, and the walk of every row ends with shell s - 1
    * (nbr_cum).  Entries of later shells inside the last group walked are simply tested: they are real neighbours. */
   const double *disp; const uint4 *nbr_cum; double sh_r0sq, sh_step;
   /* decomposed runs: D bounds the OWNED beads' moves only; hdisp (not null) points at the largest squared distance of a received
    * halo bead from its place at the rebuild (k_halo_update), and a pair distance has changed by at most D + max(D, sqrt(*hdisp)) */
   const double *hdisp;
   /* bonded terms / restraints: their kernels ran first and left their force on every owned bead in fx, fy, fz (zero where a bead
    * has none); the pair kernel adds the bead's pair force to it -- in memory (plain launch) or in registers, in front of the
    * integrator's pass (FUSE; it hands the array back zeroed for the next step's bonded kernels) */
   int addf;
};
/* k_nonbond<..., FUSE>: the pair kernel's epilogue is the integrator's pass over the bead -- BACK half kick, kinetic terms, FRONT half
 * kick, drift (k_kick_ke_drift, bit for bit) -- for systems whose forces are complete when the list walk ends (no bonded terms,
 * restraints, constraints or barostat; FREE / BERENDSEN groups).  The force never goes to memory; the drifted positions go to the
 * second position buffer (the neighbours still read the old one), which the host swaps in after the launch. */
struct FuseArgs
{
   double dt;
   double lam;                          /* Berendsen scale factor of the FRONT kick (1 otherwise): one value -- steps whose groups differ take the split kernels */
   const double *invmass, *massv;
   double *vx, *vy, *vz;
   double4 *pos_new;
   double *kpartials;                   /* [item][8]: rk, tion[6] of the item's beads */
   int ke_off;                          /* LDS byte offset of the [waves][8] rows of kinetic sums */
};

/* The neighbour search of one tile (first half of the list build).
 *
 * LDS image of the tile's neighbourhood: 16 B per staged bead -- position relative to the tile centre in single
 * precision and, in the fourth word, the bead's finished 16-bit list entry ((staged slot + 1) << 4 | type nibble, or
 * the bare slot + 1); for molecular systems the word also carries the atom-in-molecule code (6 bits, 63 = "ask the
 * record") and the low byte of the molecule id, and the full ids sit in a second array.  One lane per owned bead walks
 * the 5x5 rows of cells around its own cell, four candidates per trip.  The trip is free of divergent code: distance,
 * three compares, the scratch word (entry | distance shell << 16: one fma, one conversion, one shift-or), a masked
 * 4-byte store to the bead's scratch row and a carry add for its count.  Two things leave the straight path, each
 * behind ONE wave-wide branch per trip: candidates inside the error band of the single-precision r^2 (re-tested from
 * the double positions: the list criterion stays the reference's r^2 < rlist^2, pairlist.c:262-282) and candidates of
 * the bead's own molecule (reOrgPairs, bioMartini.c:1392-1485: bonded partners go to the excluded list instead).
 * (The per-candidate accept branch of the first version -- parity logic for paired 8-byte stores, shell clamps, the
 * pack-type selects -- was 60 % of this kernel's vector instructions and most of its scalar branches.) */
template <bool HAS_MOL, int PACK>      /* HAS_MOL false: every molecule is a single bead, the molecule logic is compiled out; PACK = TileArgs::pack_type */
__global__ __launch_bounds__(TB_THREADS) void k_tile_build(GridParams gp, TileArgs ta, int npad, const double4 *__restrict__ pos, const uint64_t *__restrict__ gid,
                                                            const int *__restrict__ species,
                                                            int nmoltype, const int *moltype_sp, const int *mol_nspecies, const int *bpair_off,
                                                            const int *bpairI, const int *bpairJ, const unsigned long long *exmask,
                                                            int maxexcl, unsigned short *excl16, int *excl_cnt, int *flags)
{
   /* dynamic LDS only, so that the ring of accepted words starts at LDS address 0 (its address arithmetic is one and-or) */
   extern __shared__ float4 tb_smem[];
   float4 *P_s = tb_smem + TB_RING_BYTES / sizeof(float4);
   int *ofs_s = (int *)(P_s + ta.cap);                     /* [NRC+1] staged offset of each region cell */
   int *gst_s = ofs_s + NRC + 8;                           /* [NRC] global start of each region cell */
   int *s_w = gst_s + NRC + 8;                             /* [TB_THREADS / 64] scan scratch */
   float *s_amax = (float *)(s_w + TB_THREADS / 64);       /* [TB_THREADS / 64] */
   int *s_halo_p = (int *)(s_amax + TB_THREADS / 64);      /* [8]: [0] the neighbourhood holds image/halo beads, [1] a molecule id beyond 24 bits */
   unsigned short *M_s = (unsigned short *)(s_halo_p + 8); /* [cap] (HAS_MOL) bits 8-23 of the staged beads' molecule ids; bits 0-7 ride in the image */
   /* [cap] region cell of each staged slot: staging only, in the bytes that become the ring (behind everything else if it outgrows them: bare 16-bit entries) */
   unsigned short *cellof_s = (size_t)ta.cap * sizeof(unsigned short) <= TB_RING_BYTES ? (unsigned short *)tb_smem : M_s + (HAS_MOL ? ta.cap : 0);
   if ((unsigned)(__UINTPTR_TYPE__)(__attribute__((address_space(3))) void *)tb_smem != 0u) __builtin_trap();
#define s_halo (*s_halo_p)
   int t = blockIdx.x;
   int ts = ta.cell_start_o[TCELLS * t], te = ta.cell_start_o[TCELLS * t + TCELLS];
   int nown = te - ts;
   if (nown <= 0)
   {
      if (threadIdx.x == 0) { ta.tile_nstage[t] = 0; ta.tile_rows[t] = 0; ta.tile_width[t] = 0; ta.tile_base[t] = 0; for (int q = 0; q < 5; q++) ta.tile_work[q * ta.ntile + t] = 0; }
      return;
   }
   int tx = t % gp.T[0], ty = (t / gp.T[0]) % gp.T[1], tz = t / (gp.T[0] * gp.T[1]);
   /* phase 0: the NRC region cells (raster order, x fastest), their counts and staged offsets */
   constexpr int CPT = (NRC + 255) / 256;                  /* cells per thread: the first NRC/CPT threads carry them */
   static_assert(CPT * TB_THREADS >= NRC && TB_THREADS >= 256, "region cell count / block size mismatch");
   int v[CPT], g[CPT];
   int vsum = 0;
   if (threadIdx.x == 0) { s_halo = 0; s_halo_p[1] = 0; }
#pragma unroll
   for (int h = 0; h < CPT; h++)
   {
      int c = CPT * threadIdx.x + h;
      v[h] = 0; g[h] = 0;
      if (c >= NRC) continue;
      int cx = TCX * tx - 2 + (c % RGX), cy = TCY * ty - 2 + ((c / RGX) % RGY), cz = TCZ * tz - 2 + (c / (RGX * RGY));
      if (cx >= 0 && cy >= 0 && cz >= 0 && cx < gp.g[0] && cy < gp.g[1] && cz < gp.g[2])
      {
         int id = cell_linear(gp, cx, cy, cz);
         v[h] = ta.cell_cnt[id]; g[h] = ta.cell_start[id];
      }
      vsum += v[h];
   }
   /* does the neighbourhood hold image/halo beads?  (k_nonbond may run such tiles after the halo exchange) */
   bool halo_here = false;
#pragma unroll
   for (int h = 0; h < CPT; h++) halo_here |= (v[h] > 0 && g[h] >= ta.nloc);
   int tot;
   int ex = block_excl_scan<TB_THREADS / 64>(vsum, &tot, s_w);          /* (its barriers also order the s_halo reset) */
   if (halo_here) s_halo = 1;
#pragma unroll
   for (int h = 0; h < CPT; h++)
   {
      if (CPT * threadIdx.x + h < NRC)
      {
         ofs_s[CPT * threadIdx.x + h] = ex;
         gst_s[CPT * threadIdx.x + h] = g[h];
      }
      ex += v[h];
   }
   if (threadIdx.x == 0) { ofs_s[NRC] = tot; ta.tile_nstage[t] = tot; }
   if (tot > ta.cap || tot > (PACK ? 4095 : 65534))     /* staged slot 0 is the sentinel */
   {
      if (threadIdx.x == 0) { atomicMax(&flags[4], tot); ta.tile_rows[t] = 0; ta.tile_width[t] = 0; ta.tile_base[t] = 0; for (int q = 0; q < 5; q++) ta.tile_work[q * ta.ntile + t] = 0; }
      return;      /* LDS capacity too small: the host retries with a larger cap */
   }
   __syncthreads();
   /* phase 1: staging list (global indices) + the LDS image */
   int *sidx = ta.stage_idx + (size_t)t * ta.stage_stride;
   const double ox = gp.lo[0] + (TCX * tx - gp.m[0] + 0.5 * TCX) / gp.cinv[0], oy = gp.lo[1] + (TCY * ty - gp.m[1] + 0.5 * TCY) / gp.cinv[1],
                oz = gp.lo[2] + (TCZ * tz - gp.m[2] + 0.5 * TCZ) / gp.cinv[2];
   /* one thread per staged slot, four gathers in flight: a slot -> cell map in LDS gives every slot its global
    * index (a loop over each cell's beads by the thread that owns the cell serialised a dozen memory round trips) */
#pragma unroll
   for (int h = 0; h < CPT; h++)
   {
      const int c = CPT * threadIdx.x + h;
      if (c < NRC) { const int o = ofs_s[c]; for (int k = 0; k < v[h]; k++) cellof_s[o + k] = (unsigned short)c; }
   }
   __syncthreads();
   float amax = 0.0f;      /* largest staged coordinate: sizes the band of the exact test */
   for (int k0 = threadIdx.x; k0 < tot; k0 += 4 * TB_THREADS)
   {
      int gj[4];
      double4 p4[4];
      int hs4[4];
#pragma unroll
      for (int u = 0; u < 4; u++)
      {
         const int k = k0 + u * TB_THREADS;
         gj[u] = ts;
         if (k < tot) { const int c = cellof_s[k]; gj[u] = gst_s[c] + (k - ofs_s[c]); }
      }
#pragma unroll
      for (int u = 0; u < 4; u++)
      {
         p4[u] = pos[gj[u]];
         hs4[u] = (PACK == 2 && gj[u] >= ta.nloc) ? ta.halo_shift[gj[u] - ta.nloc] : 13;
      }
#pragma unroll
      for (int u = 0; u < 4; u++)
      {
         const int k = k0 + u * TB_THREADS;
         if (k < tot)
         {
            sidx[k] = gj[u];
            const double4 p = p4[u];
            const unsigned long long w = (unsigned long long)__double_as_longlong(p.w);
            const unsigned lo = (unsigned)w;
            unsigned nib = lo & 0xfu;
            if (PACK == 2 && hs4[u] != 13) nib |= 8u;      /* a periodically shifted copy: bit 3 of the entry's type nibble */
            unsigned wv = PACK ? (((unsigned)(k + 1) << 4) | nib) : (unsigned)(k + 1);      /* the bead's list entry, finished */
            if (PACK) ta.tile_nib[(size_t)t * ta.stage_stride + k] = (unsigned char)nib;
            if (HAS_MOL)
            {
               const unsigned mol = (unsigned)(w >> 32);
               wv |= (min((lo >> 8) & 0xffu, 63u) << 16) | ((mol & 0xffu) << 24);
               M_s[k] = (unsigned short)(mol >> 8);
               if (mol >> 24) s_halo_p[1] = 1;      /* (more than 16.7 M molecules: a match of the 24 staged bits is confirmed from the record) */
            }
            const float4 ps = make_float4((float)(p.x - ox), (float)(p.y - oy), (float)(p.z - oz), __uint_as_float(wv));
            amax = fmaxf(amax, fmaxf(fabsf(ps.x), fmaxf(fabsf(ps.y), fabsf(ps.z))));
            P_s[k] = ps;
         }
      }
   }
   const double rl2 = gp.rlist * gp.rlist;
   /* Error of the single-precision r^2: the staged coordinates are rounded once (half an ulp of the largest coordinate
    * A relative to the tile centre: A 2^-24), so a separation component is off by < 2 A 2^-24 and r^2 at r = rlist by
    * < 2 sqrt(3) rlist 2 A 2^-24 + 4 rlist^2 2^-24 of arithmetic rounding.  The band is four times that: 6e-6 relative
    * for rlist = 16 A (A = 48 A); beads far outside an open box widen it. */
#pragma unroll
   for (int off = 32; off > 0; off >>= 1) amax = fmaxf(amax, __shfl_xor(amax, off, 64));
   if ((threadIdx.x & 63) == 0) s_amax[threadIdx.x >> 6] = amax;
   __syncthreads();
#pragma unroll
   for (int q = 0; q < TB_THREADS / 64; q++) amax = fmaxf(amax, s_amax[q]);
   const double band = 4.0 * (4.0 * 1.7320508 * gp.rlist * (double)amax + 4.0 * rl2) * 5.9604645e-8 / rl2;
   const float rl2_hi = (float)(rl2 * (1.0 + band)), rl2_lo = (float)(rl2 * (1.0 - band));
   /* distance shell of an accepted candidate (the order of a bead's entries, k_tile_transpose): 0 below r0, then
    * NSHELL-1 equal steps of r^2 up to the list radius -- one fma and one saturating conversion (negative -> 0; every
    * accepted r^2 is < rl2_hi, which maps below NSHELL: no clamp).  Boundaries only steer the ORDER: single precision */
   const float shA = ta.shc.one ? 0.0f : (float)(NSHELL - 1.01) / (rl2_hi - ta.shc.r0sq), shB = ta.shc.one ? 0.0f : 1.0f - ta.shc.r0sq * shA;
   const int rows = (nown + 63) & ~63;
   const bool mol_wide = HAS_MOL && s_halo_p[1] != 0;
   int mymax = 0;
   /* phase 2: ONE scan of the 5x5x5 cells around each bead.  Accepted neighbours go to the bead's own row of a
    * row-major scratch list tagged with their distance shell; k_tile_transpose lays them out slot-major in shell order */
   constexpr unsigned SCRB = PACK ? 2u : 4u;      /* bytes of a scratch word */
   char *const trow = (char *)ta.tmp32 + ((size_t)ts + (size_t)TB_CHUNK * t) * ta.tmpw * SCRB;      /* the tile's scratch chunks (wave-uniform base, 32-bit lane offsets) */
   const int wlim = ta.tmpw - 4;                                /* a trip stores while its row has room for four more words */
   for (int al = threadIdx.x; al < nown; al += TB_THREADS)
   {
      const int a = ts + al;
      const double4 pi = pos[a];
      const float fx = (float)(pi.x - ox), fy = (float)(pi.y - oy), fz = (float)(pi.z - oz);
      int cx, cy, cz;
      cell_coords(gp, pi.x, pi.y, pi.z, true, cx, cy, cz);
      const int lx = cx - TCX * tx, ly = cy - TCY * ty, lz = cz - TCZ * tz;
      const int rc_own = (lz + 2) * (RGX * RGY) + (ly + 2) * RGX + (lx + 2);
      const int self = ofs_s[rc_own] + (a - gst_s[rc_own]);
      /* molecule data of the bead.  exmask[mt*64 + a]: atoms (codes < 63) of molecule type mt bonded to atom a; bit 63 of
       * entry a = 0 is set when the whole type can be decided by mask */
      uint64_t gi = 0;
      int mt = 0, mns = 1;
      unsigned long long mask_i = 0; bool by_mask = false;
      unsigned key_i = 0;
      if (HAS_MOL)
      {
         gi = gid[a]; mt = moltype_sp[species[a]]; mns = mol_nspecies[mt];
         const unsigned aI = (unsigned)(gi & 65535ull);
         if (mns > 1 && aI < 63u && (exmask[(size_t)mt * 64] >> 63)) { by_mask = true; mask_i = exmask[(size_t)mt * 64 + aI]; }
         key_i = ((unsigned)(gi >> 32) & 0xffu) << 24;
      }
      int ecnt = 0;
      /* c11 / f11: words accepted / flushed so far, in units of RING_STEP (the byte stride of a ring slot: the ring address of word c is
       * one and-or away); gofs: byte offset of the row's next 16-byte group in the tile's scratch */
      typedef __attribute__((address_space(3))) unsigned lds_uint;
      typedef __attribute__((address_space(3))) unsigned short lds_ushort;
      /* PACK: sixteen 2-byte ring slots, a piece = eight words; else eight 4-byte slots, a piece = four words: 16 bytes either way */
      constexpr unsigned RING_STEP = TB_THREADS * SCRB, RING_SLOTS = PACK ? 16u : 8u, RING_MASK = (RING_SLOTS - 1u) * RING_STEP, PIECE_W = PACK ? 8u : 4u;
      static_assert((RING_STEP & (RING_STEP - 1)) == 0 && RING_SLOTS * RING_STEP <= TB_RING_BYTES, "the ring: a power-of-two stride, inside its LDS block");
      const unsigned tid4 = threadIdx.x * SCRB, lim11 = (unsigned)wlim * RING_STEP;
      /* the row's pieces inside its chunk: piece q at (q TB_CHUNK + lane) 16 bytes */
      const unsigned gofs0 = (unsigned)(al & ~(TB_CHUNK - 1)) * (unsigned)ta.tmpw * SCRB + (unsigned)(al & (TB_CHUNK - 1)) * 16u;
      unsigned c11 = 0, f11 = 0, gofs = gofs0;
      bool ovf = false;      /* an accepted candidate found its row full: the host grows the rows and builds again */
      /* the piece that starts at flushed count f: ring slots [0, half) or [half, all) of this lane, as one 16-byte value */
      auto ring_piece = [&](const unsigned f) -> uint4
      {
         const unsigned ra = (f & (PIECE_W * RING_STEP)) | tid4;      /* f counts whole pieces: the first or the second half of the ring */
         uint4 o;
         if (PACK)
         {
            unsigned h[8];
#pragma unroll
            for (int k = 0; k < 8; k++) h[k] = *(lds_ushort *)(__UINTPTR_TYPE__)(ra + (unsigned)k * RING_STEP);
            o = make_uint4(h[0] | (h[1] << 16), h[2] | (h[3] << 16), h[4] | (h[5] << 16), h[6] | (h[7] << 16));
         }
         else
         {
            o.x = *(lds_uint *)(__UINTPTR_TYPE__)(ra);
            o.y = *(lds_uint *)(__UINTPTR_TYPE__)(ra + RING_STEP);
            o.z = *(lds_uint *)(__UINTPTR_TYPE__)(ra + 2u * RING_STEP);
            o.w = *(lds_uint *)(__UINTPTR_TYPE__)(ra + 3u * RING_STEP);
         }
         return o;
      };
      /* one row of cells: candidates [s0, s1) of the LDS image.  SELF: the row holds the bead itself */
      auto scan_row = [&](const int s0, const int s1, auto self_row)
      {
         constexpr bool SELF = decltype(self_row)::value;
         /* four candidates per trip: the LDS reads of a trip are independent (ILP at low occupancy) */
         for (int sj0 = s0; sj0 < s1; sj0 += 4)
         {
            float4 q4[4];
#pragma unroll
            for (int u = 0; u < 4; u++) q4[u] = P_s[sj0 + u];      /* past s1: another cell's bead or the tables behind P_s, masked by u < nrem */
            const int nrem = s1 - sj0;
            float r2[4];
            bool ok[4];
            unsigned long long rare = 0;      /* lanes with a candidate inside the error band (wave-wide: scalar mask arithmetic, one scalar branch) */
#pragma unroll
            for (int u = 0; u < 4; u += 2)
            {
               /* two candidates per packed multiply / fma (the six differences are scalar subtractions into register pairs: packing their operands would cost moves) */
               typedef float f2 __attribute__((ext_vector_type(2)));
               float d[6];
               const float pc[3] = {fx, fy, fz}, qa[3] = {q4[u].x, q4[u].y, q4[u].z}, qb[3] = {q4[u + 1].x, q4[u + 1].y, q4[u + 1].z};
#pragma unroll
               for (int k = 0; k < 3; k++)
               {
                  asm("v_sub_f32 %0, %1, %2" : "=v"(d[2 * k]) : "v"(pc[k]), "v"(qa[k]));
                  asm("v_sub_f32 %0, %1, %2" : "=v"(d[2 * k + 1]) : "v"(pc[k]), "v"(qb[k]));
               }
               const f2 x = {d[0], d[1]}, y = {d[2], d[3]}, z = {d[4], d[5]};
               const f2 rr = __builtin_elementwise_fma(z, z, __builtin_elementwise_fma(y, y, x * x));
               r2[u] = rr.x; r2[u + 1] = rr.y;
            }
#pragma unroll
            for (int u = 0; u < 4; u++)
            {
               /* (ballots of plain compares are the compares' own lane masks: the band test costs one compare per candidate, the rest is scalar) */
               const bool in = u < nrem, lt = r2[u] < rl2_hi;
               ok[u] = in & lt;
               unsigned long long rm = __builtin_amdgcn_ballot_w64(in) & __builtin_amdgcn_ballot_w64(lt) & __builtin_amdgcn_ballot_w64(r2[u] > rl2_lo);
               if (SELF) { const bool ns = sj0 + u != self; ok[u] &= ns; rm &= __builtin_amdgcn_ballot_w64(ns); }
               rare |= rm;
            }
            if (rare)
            {
               /* boundary band: the reference's test on the double positions */
#pragma unroll
               for (int u = 0; u < 4; u++)
                  if (ok[u] && r2[u] > rl2_lo)
                  {
                     const double4 pj = pos[sidx[sj0 + u]];
                     const double X = pi.x - pj.x, Y = pi.y - pj.y, Z = pi.z - pj.z;
                     ok[u] = X * X + Y * Y + Z * Z < rl2;
                  }
            }
            if (HAS_MOL)
            {
               /* candidates of the bead's own molecule (the low byte of the id rides in the image: a filter, confirmed
                * against the full id) are pruned if the molecule has one species, or if the two atoms are a bonded pair
                * of the residue (bpairList: bonds, exclusions, constraints) */
               bool sm[4];
               unsigned long long anys = 0;
#pragma unroll
               for (int u = 0; u < 4; u++) { sm[u] = ok[u] & ((__float_as_uint(q4[u].w) ^ key_i) < (1u << 24)); anys |= __builtin_amdgcn_ballot_w64(sm[u]); }
               if (anys)
               {
#pragma unroll
                  for (int u = 0; u < 4; u++)
                     if (sm[u] && M_s[sj0 + u] == (unsigned short)(gi >> 40) &&
                         (!mol_wide || (unsigned)((unsigned long long)__double_as_longlong(pos[sidx[sj0 + u]].w) >> 32) == (unsigned)(gi >> 32)))
                     {
                        const int sj = sj0 + u;
                        const unsigned wj = __float_as_uint(q4[u].w);
                        bool pruned = true;
                        if (mns > 1)
                        {
                           const unsigned aI = (unsigned)(gi & 65535ull);
                           unsigned aJ = (wj >> 16) & 63u;
                           if (by_mask && aJ < 63u) pruned = (mask_i >> aJ) & 1ull;
                           else
                           {
                              if (aJ == 63u)
                              {
                                 /* the image holds codes up to 62: the record's tag has 8 bits, and 255 there sends us to the gid */
                                 const int gj = sidx[sj];
                                 aJ = (unsigned)(((unsigned long long)__double_as_longlong(pos[gj].w) >> 8) & 0xffull);
                                 if (aJ == 255u) aJ = (unsigned)(gid[gj] & 65535ull);
                              }
                              if (by_mask && aJ < 63u) pruned = (mask_i >> aJ) & 1ull;
                              else
                              {
                                 pruned = false;
                                 for (int k = bpair_off[mt]; k < bpair_off[mt + 1]; k++)
                                 {
                                    const unsigned eI = (unsigned)bpairI[k], eJ = (unsigned)bpairJ[k];
                                    if ((aI == eI && aJ == eJ) || (aJ == eI && aI == eJ)) { pruned = true; break; }
                                 }
                              }
                           }
                        }
                        if (pruned)
                        {
                           /* the pair kernel finds the partner among the staged beads (same molecule: always inside the tile's
                            * neighbourhood): an entry in the list's own format.  (ddcmi_get_list derives the partner's index from it.) */
                           if (ecnt < maxexcl) excl16[(size_t)ecnt * npad + a] = (unsigned short)(wj & 0xffffu);
                           ecnt++;
                           ok[u] = false;
                        }
                     }
               }
            }
            /* accepted words go to the lane's ring in LDS ([slot][lane]: conflict-free) and leave as 16-byte stores once four
             * are waiting: 4-byte stores straight to the scratch row -- one per candidate slot, each lane its own cache
             * line -- ran into the rate at which L2 takes write requests (2.5 ms per build at 4 M beads against 2.3 for
             * paired 8-byte stores behind three times the vector instructions) */
            /* a row that cannot take four more words starts over (and says so: the host grows the rows and builds again) --
             * one test per trip instead of a mask term per candidate */
            if (c11 > lim11) { ovf = true; c11 = 0; f11 = 0; gofs = gofs0; }
#pragma unroll
            for (int u = 0; u < 4; u++)
            {
               unsigned sh;
               const float st = fmaf(r2[u], shA, shB);
               asm("v_cvt_u32_f32 %0, %1" : "=v"(sh) : "v"(st));      /* saturating: negative -> 0 (a C cast of a negative float is undefined) */
               /* packed entries: the scratch word is 16 bits, staged slot + 1 | shell << 12 (the type nibble waits in tile_nib) */
               const unsigned wq = __float_as_uint(q4[u].w);
               const unsigned word = PACK ? ((sh << 12) | (HAS_MOL ? ((wq >> 4) & 0xfffu) : (wq >> 4))) : ((sh << 16) | (HAS_MOL ? (wq & 0xffffu) : wq));
               if (ok[u])
               {
                  if (PACK) *(lds_ushort *)(__UINTPTR_TYPE__)((c11 & RING_MASK) | tid4) = (unsigned short)word;
                  else *(lds_uint *)(__UINTPTR_TYPE__)((c11 & RING_MASK) | tid4) = word;
                  c11 += RING_STEP;
               }
            }
            if (c11 - f11 >= PIECE_W * RING_STEP)
            {
               *(uint4 *)(trow + gofs) = ring_piece(f11);      /* wave-uniform base + 32-bit lane offset; the lanes of a wave fill the same few KB */
               gofs += TB_CHUNK * 16u; f11 += PIECE_W * RING_STEP;
            }
         }
      };
      /* Of the 5x5x5 cells around the bead's cell only those within the list radius of the BEAD are walked: per
       * (y,z) row of cells the gap between the bead and the row's band, and from it the reach along x -- on average
       * 60 % of the candidates of the full cube.  Conservative: gaps are measured to the cells' geometric bounds
       * (a bead clamped into an edge cell from outside the grid lies further out, never nearer), the bead's own
       * cell column is always inside the range, and the radius carries the margin of the single-precision image. */
      const float ux = (float)((pi.x - gp.lo[0]) * gp.cinv[0]) + (float)(gp.m[0] - (TCX * tx - 2));      /* bead in region-cell units */
      const float gfy = (float)((pi.y - gp.lo[1]) * gp.cinv[1]) + (float)(gp.m[1] - (TCY * ty - 2)) - (float)(ly + 2);
      const float gfz = (float)((pi.z - gp.lo[2]) * gp.cinv[2]) + (float)(gp.m[2] - (TCZ * tz - 2)) - (float)(lz + 2);
      const float csy = (float)(1.0 / gp.cinv[1]), csz = (float)(1.0 / gp.cinv[2]), cix = (float)gp.cinv[0];
      const float rl2p = (float)(rl2 * (1.0 + 4.0e-4));
#pragma unroll 1
      for (int dz = 0; dz < 5; dz++)
      {
         const float gz = fmaxf(dz < 2 ? (gfz + (float)(1 - dz)) * csz : dz > 2 ? ((float)(dz - 2) - gfz) * csz : 0.0f, 0.0f);
#pragma unroll 1
         for (int dy = 0; dy < 5; dy++)
         {
            const float gy = fmaxf(dy < 2 ? (gfy + (float)(1 - dy)) * csy : dy > 2 ? ((float)(dy - 2) - gfy) * csy : 0.0f, 0.0f);
            const float d2yz = gy * gy + gz * gz;
            int s0 = 0, s1 = 0;
            if (d2yz < rl2p)
            {
               const float wx = __builtin_amdgcn_sqrtf(rl2p - d2yz) * cix * 1.0001f + 1.0e-4f;
               const int xlo = max(min((int)floorf(ux - wx), lx + 2), lx), xhi = min(max((int)floorf(ux + wx), lx + 2), lx + 4);
               const int rowb = (lz + dz) * (RGX * RGY) + (ly + dy) * RGX;      /* consecutive cells in x are contiguous */
               s0 = ofs_s[rowb + xlo]; s1 = ofs_s[rowb + xhi + 1];
            }
            if (dz == 2 && dy == 2) scan_row(s0, s1, std::true_type()); else scan_row(s0, s1, std::false_type());
         }
      }
      if (c11 != f11) *(uint4 *)(trow + gofs) = ring_piece(f11);      /* the last words (the piece's tail is never read: the row's count says so) */
      const int cnt = (int)(c11 / RING_STEP);
      mymax = max(mymax, min(cnt, ta.tmpw));
      ta.nbr_cnt[a] = min(cnt, ta.tmpw);
      excl_cnt[a] = min(ecnt, maxexcl);
      if (ecnt > maxexcl) atomicMax(&flags[1], ecnt);
      if (ovf) atomicMax(&flags[5], ta.tmpw + ta.tmpw / 4);
   }
   /* block max -> ELL width of this tile; one thread takes the arena slice */
   int m = mymax;
#pragma unroll
   for (int off = 32; off > 0; off >>= 1) m = max(m, __shfl_down(m, off, 64));
   if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = m;
   __syncthreads();
   if (threadIdx.x == 0)
   {
      int width = 0;
      for (int q = 0; q < TB_THREADS / 64; q++) width = max(width, s_w[q]);
      width = (max(min(width, ta.tmpw), 1) + 7) & ~7;      /* slots are stored in groups of 8 (one 16-byte load per lane) */
      unsigned long long need = (unsigned long long)rows * width;
      unsigned long long b0 = atomicAdd(ta.arena_used, need);
      long long sb;
      if (b0 + need > ta.arena_cap) { sb = -1; atomicMax(&flags[0], 1); }
      else sb = (long long)b0;
      ta.tile_base[t] = sb < 0 ? 0 : sb;
      ta.tile_width[t] = width; ta.tile_rows[t] = sb < 0 ? 0 : rows;
      ta.tile_work[4 * ta.ntile + t] = width;
   }
   /* statistics: entries of this tile.  They leave as per-tile numbers that the host adds up (it reads the tiles' cost
    * estimates anyway): with one atomic per wave on a device-wide total -- eighteen same-address atomics per tile, all on
    * one cache line with the arena counter -- the atomics were this kernel: 1.8 ms of its 2.3 at 4 M beads for an
    * instance that staged its neighbourhood and searched nothing */
   int mine = 0, mex = 0;
   for (int al = threadIdx.x; al < nown; al += TB_THREADS) { mine += ta.nbr_cnt[ts + al]; mex += excl_cnt[ts + al]; }
   for (int off = 32; off > 0; off >>= 1) { mine += __shfl_down(mine, off, 64); mex += __shfl_down(mex, off, 64); }
   __syncthreads();
   if ((threadIdx.x & 63) == 0) { s_w[threadIdx.x >> 6] = mine; s_amax[threadIdx.x >> 6] = __int_as_float(mex); }
   __syncthreads();
   if (threadIdx.x == 0)
   {
      int te_ = 0, tx_ = 0;
      for (int q = 0; q < TB_THREADS / 64; q++) { te_ += s_w[q]; tx_ += __float_as_int(s_amax[q]); }
      ta.tile_work[2 * ta.ntile + t] = te_; ta.tile_work[3 * ta.ntile + t] = tx_;
      /* residency of the tile's workgroup in k_nonbond: (passes x list groups per lane),
       * scaled so that a full tile counts its list entries, + staging */
      constexpr int NWAVES = NB_THREADS / 64;
      const int ngrp = (ta.tile_width[t] + 7) >> 3;
      int work = 0;
      for (int row0 = 0; row0 < nown; row0 += 64 * NWAVES)      /* k_nonbond's passes over a tile with more beads than threads */
      {
         const int nhere = min(nown - row0, 64 * NWAVES);
         int R = 64;
         while (R > 1 && (R >> 1) * NWAVES >= nhere) R >>= 1;
         const int parts = 64 / R;
         work += ((ngrp + parts - 1) / parts) * 8 * 64 * NWAVES;
      }
      /* [t]: the list walk (bit 30: the tile stages image/halo beads); [ntile + t]: staging, in the same unit --
       * calibrated on per-workgroup timelines: a full tile walks ~80 k units in 25 us and stages 2400-3000 beads in 5-7.5 us */
      ta.tile_work[t] = (work + 1) | (s_halo ? (1 << 30) : 0);
      ta.tile_work[ta.ntile + t] = 7 * tot;
   }
#undef s_halo
}

/* second half of the build: row-major scratch -> the tile's slot-major ELL slice with entries ordered by distance
 * shell.  Every WAVE works alone on eight rows at a time (eight lanes per row): no workgroup barriers, no staging of
 * the scratch rows in LDS.  A lane loads its share of the row -- the 16-byte quads q, q+8, q+16, ... of the row, so the
 * eight lanes of a row read 128 contiguous bytes per load -- and keeps the words in registers through both passes of a
 * counting sort by shell: counts and cursors are per-lane columns of an LDS table ([shell][thread]: conflict-free),
 * advanced by LDS atomics; a prefix over the eight lanes of a row turns counts into cursors; the entries land in a
 * small LDS image of the eight rows ([slot][row]) and leave as 16-byte stores, 128 contiguous bytes per slot group.
 * The order inside a shell is (lane, quad, word): fixed by the data alone, so a run repeats bit for bit.
 * (The first version staged 32 rows per workgroup in LDS and read every word back twice: the LDS pipe was busy 60 % of
 * that kernel's 0.95 ms at 4 M beads, three barriers per chunk kept its waves in step.) */
#define TR_THREADS 256
#define TR_WROWS 8                         /* rows a wave sorts together */
#define TR_S 9                             /* row stride of the wave's image in 16-bit entries: [slot][TR_S] */
template <int NQ, bool SCR16>              /* quads a lane may hold: rows of up to 32 NQ (SCR16: 64 NQ) scratch words; SCR16: 16-bit words (packed entries) */
__global__ __launch_bounds__(TR_THREADS) void k_tile_transpose(TileArgs ta)
{
   extern __shared__ unsigned int tr_smem[];
   __shared__ unsigned cur_s[NSHELL * TR_THREADS];   /* [shell][thread]: counts, then cursors */
   static_assert(NSHELL == 8, "two words of four 16-bit shell counters");
   constexpr int EPQ = SCR16 ? 8 : 4;                /* scratch words in a 16-byte quad */
   const int t = blockIdx.x;
   const int ts = ta.cell_start_o[TCELLS * t];
   const int nown = ta.cell_start_o[TCELLS * t + TCELLS] - ts;
   const int rows = ta.tile_rows[t];
   if (nown <= 0 || rows <= 0) return;
   const int width = ta.tile_width[t];
   const long long base = ta.tile_base[t];
   const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
   const int rl = lane >> 3, q = lane & 7;      /* row of the batch, lane of the row */
   unsigned short *img = (unsigned short *)tr_smem + (size_t)w * ta.tmpw * TR_S;      /* this wave's image: [width][TR_S] */
   /* SCR16: the type nibble of every staged slot (slot + 1 indexes it; [0] = the sentinel's), behind the waves' images */
   unsigned char *nib_s = (unsigned char *)((unsigned short *)tr_smem + (size_t)(TR_THREADS / 64) * ta.tmpw * TR_S);
   if (SCR16)
   {
      const int ns = ta.tile_nstage[t];
      const unsigned char *src = ta.tile_nib + (size_t)t * ta.stage_stride;
      for (int i = threadIdx.x; i <= ns; i += TR_THREADS) nib_s[i] = i ? src[i - 1] : (unsigned char)0;
      __syncthreads();      /* the one barrier of the kernel: from here on every wave works alone */
   }
   unsigned *mycur = cur_s + threadIdx.x;
   const int ngrp = width >> 3;
   for (int r0 = w * TR_WROWS; r0 < rows; r0 += TR_WROWS * (TR_THREADS / 64))
   {
      const int row = r0 + rl;
      const int cnt = row < nown ? ta.nbr_cnt[ts + row] : 0;
      const int nq = (cnt + EPQ - 1) / EPQ;
      /* the row's 16-byte pieces: piece p of row l of a chunk at (p TB_CHUNK + l) 16 bytes (TileArgs::tmp32) -- eight rows side by side
       * are 128 contiguous bytes per piece */
      const int rowc = min(row, nown - 1);
      const uint4 *src = (const uint4 *)((const char *)ta.tmp32 + ((size_t)ts + (size_t)TB_CHUNK * t + (size_t)(rowc & ~(TB_CHUNK - 1))) * ta.tmpw * (SCR16 ? 2 : 4)) + (rowc & (TB_CHUNK - 1));      /* tmpw is a multiple of 8 */
      uint4 wv[NQ];
#pragma unroll
      for (int j = 0; j < NQ; j++) wv[j] = (q + 8 * j < nq) ? src[(size_t)(q + 8 * j) * TB_CHUNK] : make_uint4(0, 0, 0, 0);
#pragma unroll
      for (int sh = 0; sh < NSHELL; sh++) mycur[sh * TR_THREADS] = 0u;
      /* padding of the row (entry 0 = the sentinel bead): slots cnt .. width-1 */
      for (int s = cnt + q; s < width; s += 8) img[s * TR_S + rl] = 0;
      /* the scratch words of quad j as {shell, entry}: 32-bit words carry the finished entry, 16-bit ones the staged slot */
      auto unpack = [&](const uint4 &v, int i, unsigned &sh, unsigned &ent)
      {
         const unsigned x[4] = {v.x, v.y, v.z, v.w};
         if (SCR16) { const unsigned e = (i & 1) ? (x[i >> 1] >> 16) : (x[i >> 1] & 0xffffu); sh = e >> 12; ent = e & 0xfffu; }
         else { sh = x[i] >> 16; ent = x[i] & 0xffffu; }
      };
      /* counts per shell */
#pragma unroll
      for (int j = 0; j < NQ; j++)
      {
         const int k = EPQ * (q + 8 * j);
#pragma unroll
         for (int i = 0; i < EPQ; i++)
         {
            unsigned sh, ent;
            unpack(wv[j], i, sh, ent);
            if (k + i < cnt) atomicAdd(mycur + sh * TR_THREADS, 1u);      /* the shell, from k_tile_build */
         }
      }
      /* offsets: shells in order, inside a shell the row's lanes in order -- on the eight counts packed as 16-bit
       * fields of two 64-bit words (a row holds < 65536 entries) */
      unsigned long long c0w = 0, c1w = 0;
#pragma unroll
      for (int sh = 0; sh < 4; sh++)
      {
         c0w |= (unsigned long long)mycur[sh * TR_THREADS] << (16 * sh);
         c1w |= (unsigned long long)mycur[(sh + 4) * TR_THREADS] << (16 * sh);
      }
      unsigned long long i0 = c0w, i1 = c1w;
#pragma unroll
      for (int off = 1; off < 8; off <<= 1)
      {
         unsigned long long v0 = __shfl_up(i0, off, 8), v1 = __shfl_up(i1, off, 8);
         if (q >= off) { i0 += v0; i1 += v1; }
      }
      const unsigned long long t0 = __shfl(i0, 7, 8), t1 = __shfl(i1, 7, 8);
      /* field i of (x << 16) + (x << 32) + (x << 48) = sum of the fields below i */
      const unsigned long long b0 = (t0 << 16) + (t0 << 32) + (t0 << 48);
      const unsigned long long n03 = ((b0 + t0) >> 48) & 0xffffull;                  /* entries in shells 0..3 */
      const unsigned long long b1 = n03 * 0x0001000100010001ull + (t1 << 16) + (t1 << 32) + (t1 << 48);
      const unsigned long long s0 = b0 + i0 - c0w, s1 = b1 + i1 - c1w;
      if (q == 0 && row < nown)
      {
         /* entries in shells 0..s, s = 0..7, as eight 16-bit fields */
         const unsigned long long n0 = b0 + t0, n1 = b1 + t1;
         ta.nbr_cum[ts + row] = make_uint4((unsigned)n0, (unsigned)(n0 >> 32), (unsigned)n1, (unsigned)(n1 >> 32));
      }
#pragma unroll
      for (int sh = 0; sh < 4; sh++)
      {
         mycur[sh * TR_THREADS] = (unsigned)((s0 >> (16 * sh)) & 0xffffull);
         mycur[(sh + 4) * TR_THREADS] = (unsigned)((s1 >> (16 * sh)) & 0xffffull);
      }
      /* placement (16-bit scratch: the entry is finished here, staged slot << 4 | the slot's nibble) */
#pragma unroll
      for (int j = 0; j < NQ; j++)
      {
         const int k = EPQ * (q + 8 * j);
         unsigned slot[EPQ], ent[EPQ];
#pragma unroll
         for (int i = 0; i < EPQ; i++)
         {
            unsigned sh;
            unpack(wv[j], i, sh, ent[i]);
            slot[i] = (k + i < cnt) ? atomicAdd(mycur + sh * TR_THREADS, 1u) : 0u;
            if (SCR16) ent[i] = (ent[i] << 4) | nib_s[ent[i]];
         }
#pragma unroll
         for (int i = 0; i < EPQ; i++) if (k + i < cnt) img[slot[i] * TR_S + rl] = (unsigned short)ent[i];
      }
      /* the wave's LDS operations complete in order: the image is whole when the reads below are issued */
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      /* slice layout: [slot group g][row][8 slots] -> a lane of k_nonbond reads its 8 slots of a group with one 16-byte
       * load; written the same way: lane (g, r) packs the group's 8 entries of its row, 8 rows = 128 contiguous bytes */
      {
         const int ro = lane & 7;
         for (int g = lane >> 3; g < ngrp; g += 8)
         {
            const unsigned short *e = img + (8 * g) * TR_S + ro;
            uint4 o;
            o.x = (unsigned)e[0] | ((unsigned)e[TR_S] << 16);
            o.y = (unsigned)e[2 * TR_S] | ((unsigned)e[3 * TR_S] << 16);
            o.z = (unsigned)e[4 * TR_S] | ((unsigned)e[5 * TR_S] << 16);
            o.w = (unsigned)e[6 * TR_S] | ((unsigned)e[7 * TR_S] << 16);
            *(uint4 *)(ta.nbr16 + base + ((size_t)g * rows + r0 + ro) * 8) = o;
         }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();      /* (the next batch rewrites the image) */
   }
}

/* ------------------------------------------------------------------------- */
/* THE hot kernel: martiniNonBond (bioMartini.c:989-1122) + martiniIntraMoleReaction
 * (:1124-1208) over the full list.
 *
 * One 512-thread workgroup per WORK ITEM: a tile (8x4x4 cells, ~500 owned beads) or, in the last round of a
 * launch, one of several row ranges of a tile (schedule_tiles).  Two workgroups per CU (72 KB of LDS each).
 *   1. staging: the tile's neighbourhood (the 12x8x8 cells around it, ~3000 beads, owned and image/halo alike)
 *      goes to LDS as z[] and {x,y}[] (24 B per bead).  Its order is the list build's: region cells in raster
 *      order; the global index of a staged slot comes from the cell tables, not from a per-tile index list.
 *      Slot 0 is a sentinel bead at 1e30: list padding points at it, so the walk has no validity masks.
 *   2. walk: one wave per chunk of R rows, one lane per bead (R = 64 for a full tile; thin tiles and row parts
 *      give each bead 64/R lanes that split its list).  List entries are 16 bits, (staged slot + 1) << 4 | type
 *      [| shifted-copy bit], stored per tile as [group of 8 slots][row][8]: one 16-byte load per lane and group,
 *      kept two groups ahead.  Per slot: two LDS gathers by raw byte offset, the distance test, and under it
 *        ir2  = 1/r2   (v_rcp_f32 seed + 2 Newton steps; with charges ir = 1/sqrt(r2) from v_rsq_f32)
 *        s2   = sigma^2 ir2 ; s6 = s2^3 ; s12 = s6^2
 *        vLJ += 4eps(s12-s6)+shift ; dvdr = 24eps(s6-2s12) ir2
 *        vEle+= kqij(ir + krf r2 - crf) ; dvdr += kqij(2krf - ir^3)       (kqij from the type-pair table)
 *        f_i -= dvdr d
 *      Rows are ordered by distance shell at build time, so late groups are rejected by whole waves.
 *   3. excluded same-molecule pairs (charged systems): reaction-field term only, from a short global list.
 *   4. the bead's force is stored (full list: no atomics, no force return); energy and virial partial sums of the
 *      item go to partials[item][8].  Virial: 2 F_i (x) r_i per bead for unshifted partners, per pair for
 *      shifted copies and excluded pairs (see below).
 * Bound: FP64 issue and LDS gathers behind s_waitcnt at 4 waves per SIMD -- DESIGN.md section 4 has the
 * counters, the ablations and the per-CU timelines. */
template <bool HAS_Q, bool PACKED, bool SHBIT, int NB_BLOCK, int WPE, int CH, int ZOFF, bool FUSE>
__global__ __launch_bounds__(NB_BLOCK, WPE) void k_nonbond(GridParams gp, NbTileArgs ta, int npad,
                                                         const double4 *__restrict__ pos, const double *__restrict__ kqtab,
                                                         const unsigned short *__restrict__ excl16, const int *__restrict__ excl_cnt,
                                                         const double4 *__restrict__ ljtab,
                                                         double rc2, double krf, double crf, double keR,
                                                         double *__restrict__ fx, double *__restrict__ fy, double *__restrict__ fz,
                                                         double *__restrict__ partials, FuseArgs fa)
{
   /* LDS: staged neighbourhood as {x,y} pairs + z (24 B per bead), LJ table, and --
    * only when needed -- per-bead LJ types (nlj > 16) and charges */
   extern __shared__ double2 smem[];
   /* staged positions, 24 B per bead.  ZOFF > 0 (neighbourhoods of up to ZOFF/16 beads: every Martini system): {x,y} [cap] at LDS
    * address 0 and z [cap] at the compile-time byte offset ZOFF, so a gather's addresses are the entry's slot bits themselves
    * (slot * 16 for {x,y}; slot * 8 + the instruction's immediate offset for z).  ZOFF = 0: z [cap], then {x,y} [cap] at a run-time offset */
   double2 *XY_s = ZOFF ? (double2 *)smem : (double2 *)((double *)smem + ta.cap);
   double *Z_s = ZOFF ? (double *)((char *)smem + ZOFF) : (double *)smem;
   double4 *s_lj = ZOFF ? (double4 *)(Z_s + ta.cap) : (double4 *)(XY_s + ta.cap);
   /* charges: a bead's "type" is its (LJ type, charge) class, so ke/eps_r q_i q_j is one more
    * per-type-pair table entry -- no per-bead charge array in LDS (it cost 8 B/bead: one
    * workgroup per CU instead of two) and no charge gather per pair */
   unsigned char *T_s = (unsigned char *)(s_lj + ta.nlj * ta.nlj);
   unsigned char *S_s = T_s + (PACKED ? 0 : ta.cap);      /* 1: the staged bead is a periodically shifted copy */
   /* The pair loop addresses the staged beads by raw LDS byte offsets (z at slot * 8,
    * {x,y} at xy_off + slot * 16): the kernel has no static LDS, so the dynamic region
    * starts at LDS address 0 and the z gather needs no base add.  Checked by the host before the first launch, not assumed. */
   typedef __attribute__((address_space(3))) const double lds_cdouble;
   typedef double xy_t __attribute__((ext_vector_type(2)));
   typedef __attribute__((address_space(3))) const xy_t lds_cxy;
   const unsigned xy_off = ZOFF ? 0u : (unsigned)(__UINTPTR_TYPE__)(__attribute__((address_space(3))) void *)XY_s;      /* (the dynamic region starts at LDS address 0: launch_forces asks the runtime, lds_starts_at_zero) */
   /* XCD-aware mapping: hardware deals workgroups round-robin over the 8 XCDs, so
    * give XCD x one contiguous tile range (schedule_tiles: equal work per XCD):
    * neighbouring tiles, which stage overlapping neighbourhoods, then share one L2.
    * Speed only. */
   const int xcd = blockIdx.x & 7;
   const int slot = ta.sched[xcd] + (int)(blockIdx.x >> 3);
   const bool mine = slot < ta.sched[xcd + 1];
   /* a work item is a tile or -- in the last round of a launch, where whole tiles would leave most CUs idle --
    * one of nparts row ranges of a tile: every part stages the tile's neighbourhood and walks its share of the rows
    * with all eight waves (the sub-64-row chunks below give each bead several lanes) */
   const int item = mine ? ta.perm[slot] : 0;
   const int t = item & 0xffffff, part = (item >> 24) & 7, nparts = ((item >> 27) & 7) + 1;
   double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};   /* vLJ, vEle, xx,yy,zz,xy,xz,yz */
   /* the last shell this launch walks (NbTileArgs::disp) */
   int smax = NSHELL - 1;
   if (ta.disp)
   {
      const double Down = *ta.disp, Dhalo = ta.hdisp ? sqrt(*ta.hdisp) : 0.0;
      const double twoD = Down + fmax(Down, Dhalo), rc = sqrt(rc2);
      while (smax >= 1 && sqrt(ta.sh_r0sq + (double)(smax - 1) * ta.sh_step) * (1.0 - 1e-4) - twoD > rc) smax--;
   }
   int nown = 0, ts = 0, r_lo = 0, r_hi = 0;
   if (mine)
   {
      ts = ta.cell_start_o[TCELLS * t];
      nown = ta.cell_start_o[TCELLS * t + TCELLS] - ts;
      r_lo = (int)(((long long)nown * part) / nparts); r_hi = (int)(((long long)nown * (part + 1)) / nparts);
   }
   if (nown > 0)
   {
      for (int k = threadIdx.x; k < ta.nlj * ta.nlj; k += NB_BLOCK) { double4 e_ = ljtab[k]; if (HAS_Q) e_.w = kqtab[k]; s_lj[k] = e_; }
      int ns = ta.tile_nstage[t];
      /* Virial.  A pair of two unshifted beads contributes f_ij (x) (r_i - r_j) from i's side and
       * the mirror term from j's side; the two add up to 2 f_ij (x) r_i + 2 f_ji (x) r_j, so each
       * side may book 2 f_ij (x) r_i instead -- summed over j that is 2 F_i (x) r_i, six FMAs per
       * bead rather than per pair.  Pairs with a periodically shifted partner (and the excluded
       * pairs) keep the per-pair form on both sides.  Only tiles that stage image/halo beads can
       * hold shifted partners. */
      const bool tshift = (ta.tile_work[t] >> 30) & 1;
      /* Stage the neighbourhood.  The staged order is k_tile_build's: the cells of the 12x8x8 region in raster
       * order, each cell's beads as they lie in the sorted arrays.  The global index of staged slot k follows from
       * the cell tables (a few KB that stay in L2) instead of a per-tile index list (12 KB per tile and step from
       * HBM, and a dependent load in front of every record gather): region cell counts -> block scan -> a
       * slot -> cell map in LDS (aliased onto the not yet written position arrays) -> one index per thread and
       * round in registers -> all record gathers of a batch in flight -> LDS writes. */
      constexpr int SU = NB_SU, MAXR = 8;      /* record gathers in flight per thread and batch; rounds of the index-free path */
      constexpr int NWV = NB_BLOCK / 64;
      if (ns <= MAXR * NB_BLOCK)
      {
         int *ofs_s = (int *)smem;                                    /* [NRC]  staged offset of each region cell */
         int *gst_s = ofs_s + NRC + 8;                                /* [NRC]  global index of its first bead */
         int *s_w = gst_s + NRC + 8;                                  /* [NWV]  scan scratch */
         unsigned short *cellof = (unsigned short *)(s_w + 16);       /* [ns]   region cell of each staged slot */
         constexpr int CPT = (NRC + NB_BLOCK - 1) / NB_BLOCK;
         const int tx = t % gp.T[0], ty = (t / gp.T[0]) % gp.T[1], tz = t / (gp.T[0] * gp.T[1]);
         int v[CPT], g[CPT], vsum = 0;
#pragma unroll
         for (int h = 0; h < CPT; h++)
         {
            const int c = CPT * (int)threadIdx.x + h;
            v[h] = 0; g[h] = 0;
            if (c < NRC)
            {
               const int cx = TCX * tx - 2 + (c % RGX), cy = TCY * ty - 2 + ((c / RGX) % RGY), cz = TCZ * tz - 2 + (c / (RGX * RGY));
               if (cx >= 0 && cy >= 0 && cz >= 0 && cx < gp.g[0] && cy < gp.g[1] && cz < gp.g[2])
               {
                  const int id = cell_linear(gp, cx, cy, cz);
                  v[h] = ta.cell_cnt[id]; g[h] = ta.cell_start[id];
               }
            }
            vsum += v[h];
         }
         int inc = vsum;
         {
            const int ln = threadIdx.x & 63;
            inc = wave_scan_inclusive_dpp(inc);
            if (ln == 63) s_w[threadIdx.x >> 6] = inc;
         }
         __syncthreads();
         int ex = inc - vsum;
#pragma unroll
         for (int q = 0; q < NWV; q++) if (q < (int)(threadIdx.x >> 6)) ex += s_w[q];
#pragma unroll
         for (int h = 0; h < CPT; h++)
         {
            const int c = CPT * (int)threadIdx.x + h;
            if (c < NRC)
            {
               ofs_s[c] = ex; gst_s[c] = g[h];
               for (int j = 0; j < v[h]; j++) cellof[ex + j] = (unsigned short)c;
            }
            ex += v[h];
         }
         __syncthreads();
         int gj[MAXR];
#pragma unroll
         for (int u = 0; u < MAXR; u++)
         {
            const int k = (int)threadIdx.x + u * NB_BLOCK;
            gj[u] = ts;      /* rounds past the end re-read the tile's first bead (a cache hit) and drop it */
            if (k < ns) { const int c = cellof[k]; gj[u] = gst_s[c] + (k - ofs_s[c]); }
         }
         __syncthreads();      /* the tables are dead: their bytes become staged positions */
#pragma unroll
         for (int b = 0; b < MAXR; b += SU)
         {
            if (b * NB_BLOCK >= ns) break;
            double4 pp[SU];
            int sh[SU];
#pragma unroll
            for (int u = 0; u < SU; u++)
            {
               pp[u] = pos[gj[b + u]];
               sh[u] = (!SHBIT && tshift && gj[b + u] >= ta.nloc) ? ta.halo_shift[gj[b + u] - ta.nloc] : 13;
            }
#pragma unroll
            for (int u = 0; u < SU; u++)
            {
               const int k = (int)threadIdx.x + (b + u) * NB_BLOCK;
               if (k < ns)
               {
                  XY_s[k + 1] = make_double2(pp[u].x, pp[u].y);
                  Z_s[k + 1] = pp[u].z;
                  if (!PACKED) T_s[k + 1] = (unsigned char)(__double_as_longlong(pp[u].w) & 0xff);
                  if (tshift && !SHBIT) S_s[k + 1] = (unsigned char)(sh[u] != 13);
               }
            }
         }
      }
      else
      {
      /* neighbourhoods beyond 4096 beads (bare 16-bit entries): through the tile's index list -- all index loads
       * first, then all record gathers, then the LDS writes */
      const int *sidx = ta.stage_idx + (size_t)t * ta.stage_stride;
      for (int k0 = threadIdx.x; k0 < ns; k0 += SU * NB_BLOCK)
      {
         int gj[SU], sh[SU];
         double4 pp[SU];
#pragma unroll
         for (int u = 0; u < SU; u++) { int k = k0 + u * NB_BLOCK; gj[u] = (k < ns) ? sidx[k] : 0; }
#pragma unroll
         for (int u = 0; u < SU; u++)
         {
            pp[u] = pos[gj[u]];
            sh[u] = (!SHBIT && tshift && gj[u] >= ta.nloc) ? ta.halo_shift[gj[u] - ta.nloc] : 13;
         }
#pragma unroll
         for (int u = 0; u < SU; u++)
         {
            int k = k0 + u * NB_BLOCK;
            if (k < ns)
            {
               XY_s[k + 1] = make_double2(pp[u].x, pp[u].y);
               Z_s[k + 1] = pp[u].z;
               if (!PACKED) T_s[k + 1] = (unsigned char)(__double_as_longlong(pp[u].w) & 0xff);
               if (tshift && !SHBIT) S_s[k + 1] = (unsigned char)(sh[u] != 13);
            }
         }
      }
      }
      if (FUSE && threadIdx.x < (NB_BLOCK / 64) * 8) ((double *)((char *)smem + fa.ke_off))[threadIdx.x] = 0.0;
      if (threadIdx.x == 0)
      {
         /* staged slot 0: a bead far outside every cutoff.  List padding (entry 0) points
          * at it, so the walk needs no per-slot validity masks. */
         XY_s[0] = make_double2(1e30, 1e30); Z_s[0] = 1e30;
         if (!PACKED) T_s[0] = 0;
         if (!SHBIT) S_s[0] = 0;
      }
      __syncthreads();
      long long base = ta.tile_base[t];
      int rows = ta.tile_rows[t];
      const int nlj = ta.nlj;
      /* one wave per chunk of R rows, R = the smallest power of two that spreads the
       * tile over all waves (64 for a full tile).  With R < 64 -- thin edge tiles, or the
       * partial last chunk -- every bead gets `parts` lanes that split its list, so a
       * tile of 120 beads is done in a quarter of a full tile's time instead of
       * keeping the LDS of the CU busy with two working waves. */
      const int lane = threadIdx.x & 63;
      constexpr int NWAVES = NB_BLOCK / 64;
      /* a tile with more beads than threads: the rows beyond the first 64*NWAVES are again spread over all
       * waves (small R, many lanes per bead) instead of queueing behind the first waves as whole chunks */
      for (int row0 = r_lo; row0 < r_hi; row0 += 64 * NWAVES)
      {
      const int nhere = min(r_hi - row0, 64 * NWAVES);
      int R = 64;
      while (R > 1 && (R >> 1) * NWAVES >= nhere) R >>= 1;
      const int nchunks = (nhere + R - 1) / R;
      for (int chunk = threadIdx.x >> 6; chunk < nchunks; chunk += NWAVES)
      {
         int kb = min(R, nhere - chunk * R);
         int parts = 64 / R;
         while (parts * 2 * kb <= 64) parts *= 2;
         int sub = lane & (parts - 1);
         int ain = lane / parts;
         bool active = ain < kb;
         int al = row0 + chunk * R + (active ? ain : 0);
         int a = ts + al;
         double4 pi = pos[a];
         int ti = (int)(__double_as_longlong(pi.w) & 0xffll);
         int cnt_full = active ? ta.nbr_cnt[a] : 0;
         if (smax < NSHELL - 1 && active)
         {
            const uint4 cq = ta.nbr_cum[a];
            const unsigned cw = smax < 2 ? cq.x : smax < 4 ? cq.y : smax < 6 ? cq.z : cq.w;
            cnt_full = (int)((smax & 1) ? (cw >> 16) : (cw & 0xffffu));
         }
         /* this lane walks slot groups sub, sub+parts, ... (8 slots each) */
         int ng_full = (cnt_full + 7) >> 3;
         int ngl = (ng_full > sub) ? (ng_full - sub + parts - 1) / parts : 0;
         double fxi = 0, fyi = 0, fzi = 0;
         double fsx = 0, fsy = 0, fsz = 0;          /* part of f_i from shifted / excluded partners (virial booked per pair) */
         /* 32-bit indexing inside the tile's slice (uniform 64-bit base + lane offset) */
         const uint4 *slice = (const uint4 *)(ta.nbr16 + base);
         const unsigned col = (unsigned)(sub * rows + al), cstride = (unsigned)(parts * rows);
         /* wave-uniform trip count; the list is read two groups ahead (one 16-byte load per
          * lane and group, 1 KiB per wave) so the HBM/L2 latency of the list stream overlaps
          * the pair math; the 8 distance tests of a group are independent (ILP) */
         const int wmax = wave_max_dpp(ngl);
         /* The list stream: one 16-byte load per lane and group, kept two groups ahead of
          * the pair loop.  Three named buffers (the loop is unrolled by three) rather
          * than a rotating one, so each wait covers exactly the oldest load; the loads
          * are unconditional global loads from a clamped group index and masked
          * afterwards -- a conditional load here becomes a select of two addresses in
          * different address spaces, i.e. a flat load that the LDS gathers then wait on. */
         const int glast = max(ngl - 1, 0);
         auto load_group = [&](int g) -> uint4
         {
            /* a lane without groups (inactive, or its part of a short row is empty) must not form an address
             * from its column: with many parts per bead that column lies beyond the tile's slice -- for the
             * last tile beyond the arena.  It reads entry 0 of the slice and masks it. */
            uint4 v = slice[ngl > 0 ? col + (unsigned)min(g, glast) * cstride : 0u];
            /* lanes past their own last group (sub-lane split, short rows) get padding */
            if (g >= ngl) v = make_uint4(0, 0, 0, 0);
            return v;
         };
         auto do_group = [&](const uint4 &q0)
         {
            const unsigned qw[4] = {q0.x, q0.y, q0.z, q0.w};
            /* pair math for slot u of the part; WD_ = the dword holding its entry, HI_ = upper half */
#define NB_PAIR(u, WD_, HI_) do { \
                  const int nib_ = (int)(((WD_) >> ((HI_) ? 16 : 0)) & 0xfu); \
                  int tjj = PACKED ? (SHBIT ? (nib_ & 7) : nib_) : (int)T_s[o[u] >> 4]; \
                  double4 lj = s_lj[ti * nlj + tjj];            /* {sigma^2, 4eps, shift, 24eps} */ \
                  double ir = 0.0, ir2; \
                  if (HAS_Q) { ir = rsqrt_f64_pair(r2[u]); ir2 = ir * ir; } \
                  else ir2 = rcp_f64_pair(r2[u]); \
                  double s2 = lj.x * ir2; \
                  double s4 = s2 * s2; \
                  double s6 = s4 * s2; \
                  double s12 = s6 * s6; \
                  acc[0] += lj.y * (s12 - s6) + lj.z; \
                  /* charged systems: the table's fourth entry is ke/eps_r q_i q_j (24 eps = 6 x 4 eps is formed here): one LDS read per pair less */ \
                  double dvdr = (HAS_Q ? 6.0 * lj.y : lj.w) * (s6 - 2.0 * s12) * ir2; \
                  if (HAS_Q) \
                  { \
                     double kqij = lj.w; \
                     acc[1] += kqij * (ir + krf * r2[u] - crf); \
                     dvdr += kqij * (2.0 * krf - ir2 * ir); \
                  } \
                  double fxij = -dvdr * x[u], fyij = -dvdr * y[u], fzij = -dvdr * z[u]; \
                  fxi += fxij; fyi += fyij; fzi += fzij; \
                  if (tshift && (SHBIT ? (nib_ & 8) : (int)S_s[o[u] >> 4])) \
                  { \
                     fsx += fxij; fsy += fyij; fsz += fzij; \
                     acc[2] += fxij * x[u]; acc[3] += fyij * y[u]; acc[4] += fzij * z[u]; \
                     acc[5] += fxij * y[u]; acc[6] += fxij * z[u]; acc[7] += fyij * z[u]; \
                  } } while (0)
            /* the group is walked in 8 / CH parts; the CH gathers and tests of a part are independent (ILP) */
#pragma unroll
            for (int h = 0; h < 8 / CH; h++)
            {
               /* 16 x staged slot of the part's neighbours */
               unsigned o[CH];
               double x[CH], y[CH], z[CH], r2[CH];
#pragma unroll
               for (int u = 0; u < CH; u++)
               {
                  unsigned wd = qw[(h * CH + u) >> 1];
                  if (PACKED) o[u] = ((u & 1) ? (wd >> 16) : wd) & 0xfff0u;
                  else o[u] = ((u & 1) ? (wd >> 16) : (wd & 0xffffu)) << 4;
                  xy_t pxy = *(lds_cxy *)(__UINTPTR_TYPE__)(xy_off + o[u]);
                  double pz = *(lds_cdouble *)(__UINTPTR_TYPE__)((o[u] >> 1) + (unsigned)ZOFF);
                  double px = pxy.x, py = pxy.y;
                  x[u] = pi.x - px; y[u] = pi.y - py; z[u] = pi.z - pz;
                  r2[u] = x[u] * x[u] + y[u] * y[u] + z[u] * z[u];
               }
#pragma unroll
               for (int u = 0; u < CH; u++)
                  if (r2[u] < rc2) NB_PAIR(u, qw[(h * CH + u) >> 1], u & 1);
            }
#undef NB_PAIR
         };
         /* charged systems: the bead's excluded partners (a few 2-byte entries, one memory round trip each if asked for after the walk) */
         int ecnt_pre = 0;
         unsigned epre[4] = {0u, 0u, 0u, 0u};
         if (HAS_Q)
         {
            ecnt_pre = (active && sub == 0) ? excl_cnt[a] : 0;
#pragma unroll
            for (int k = 0; k < 4; k++) epre[k] = (k < ecnt_pre) ? (unsigned)excl16[(size_t)k * npad + a] : 0u;
         }
         uint4 qa = load_group(0), qb = load_group(1), qc;
         int gi = 0;
         for (; gi + 3 <= wmax; gi += 3)
         {
            qc = load_group(gi + 2); do_group(qa);
            qa = load_group(gi + 3); do_group(qb);
            qb = load_group(gi + 4); do_group(qc);
         }
         if (gi < wmax) do_group(qa);
         if (gi + 1 < wmax) do_group(qb);
         if (HAS_Q)
         {
            /* excluded (same-molecule bonded) pairs: reaction-field correction only
             * (martiniIntraMoleReaction); few per bead, gathered from global memory */
            const int ecnt = ecnt_pre;
            for (int k = 0; k < ecnt; k++)
            {
               /* the partner out of LDS, like a list entry (a global gather per excluded pair at the end of every wave was a
                * memory round trip nothing overlapped); the first four entries were requested before the list walk */
               const unsigned e16 = k < 4 ? epre[k] : (unsigned)excl16[(size_t)k * npad + a];
               const unsigned oe = PACKED ? (e16 & 0xfff0u) : (e16 << 4);
               const xy_t pxy = *(lds_cxy *)(__UINTPTR_TYPE__)(xy_off + oe);
               const double pz = *(lds_cdouble *)(__UINTPTR_TYPE__)((oe >> 1) + (unsigned)ZOFF);
               const int tje = PACKED ? (SHBIT ? (int)(e16 & 7u) : (int)(e16 & 0xfu)) : (int)T_s[oe >> 4];
               double x = pi.x - pxy.x, y = pi.y - pxy.y, z = pi.z - pz;
               double r2 = x * x + y * y + z * z;
               if (r2 < rc2)
               {
                  double kqij = s_lj[ti * nlj + tje].w;
                  acc[1] += kqij * (krf * r2 - crf);
                  double dvdr = kqij * (2.0 * krf);
                  double fxij = -dvdr * x, fyij = -dvdr * y, fzij = -dvdr * z;
                  fxi += fxij; fyi += fyij; fzi += fzij;
                  fsx += fxij; fsy += fyij; fsz += fzij;
                  acc[2] += fxij * x; acc[3] += fyij * y; acc[4] += fzij * z;
                  acc[5] += fxij * y; acc[6] += fxij * z; acc[7] += fyij * z;
               }
            }
         }
         /* unshifted partners: 2 F (x) r_i; every sub-lane books its own share of F */
         {
            double px = 2.0 * (fxi - fsx), py = 2.0 * (fyi - fsy), pz = 2.0 * (fzi - fsz);
            acc[2] += px * pi.x; acc[3] += py * pi.y; acc[4] += pz * pi.z;
            acc[5] += px * pi.y; acc[6] += px * pi.z; acc[7] += py * pi.z;
         }
         /* the bead's lanes add up their shares (parts is uniform over the wave): butterflies by DPP inside the rows of 16 lanes */
         if (parts > 1) { fxi += dpp_move<0xB1>(fxi); fyi += dpp_move<0xB1>(fyi); fzi += dpp_move<0xB1>(fzi); }
         if (parts > 2) { fxi += dpp_move<0x4E>(fxi); fyi += dpp_move<0x4E>(fyi); fzi += dpp_move<0x4E>(fzi); }
         if (parts > 4) { fxi += dpp_move<0x141>(fxi); fyi += dpp_move<0x141>(fyi); fzi += dpp_move<0x141>(fzi); }
         if (parts > 8) { fxi += dpp_move<0x140>(fxi); fyi += dpp_move<0x140>(fyi); fzi += dpp_move<0x140>(fzi); }
         for (int off = 16; off < parts; off <<= 1)
         {
            fxi += __shfl_xor(fxi, off, 64); fyi += __shfl_xor(fyi, off, 64); fzi += __shfl_xor(fzi, off, 64);
         }
         if (!FUSE)
         {
            if (active && sub == 0)
            {
               if (ta.addf) { fxi += fx[a]; fyi += fy[a]; fzi += fz[a]; }
               fx[a] = fxi; fy[a] = fyi; fz[a] = fzi;
            }
         }
         else
         {
            /* k_kick_ke_drift on the bead, with the force still in registers (the same operations in the same order) */
            double ke[7] = {0, 0, 0, 0, 0, 0, 0};
            float v2max = 0.0f;      /* |v|^2 of the velocity the bead drifts with, rounded up: feeds the displacement bound D (NbTileArgs::disp) */
            if (active && sub == 0)
            {
               /* (asked for here, not before the walk: held across it these twelve registers spill, and the reload costs what the load does) */
               const int sp = (int)((__double_as_longlong(pi.w) >> 16) & 0xffffll);
               const double hk = (0.5 * fa.dt) * fa.invmass[sp], m = fa.massv[sp], lam = fa.lam;
               if (ta.addf)
               {
                  /* + the bonded terms' force on the bead (the same sum the plain launch leaves in memory); the array goes back zeroed */
                  fxi += fx[a]; fyi += fy[a]; fzi += fz[a];
                  fx[a] = 0.0; fy[a] = 0.0; fz[a] = 0.0;
               }
               double x = fma(hk, fxi, fa.vx[a]), y = fma(hk, fyi, fa.vy[a]), z = fma(hk, fzi, fa.vz[a]);
               const double vxx = x * x, vyy = y * y, vzz = z * z;
               ke[0] = 0.5 * m * (vxx + vyy + vzz);
               ke[1] = m * vxx; ke[2] = m * vyy; ke[3] = m * vzz;
               ke[4] = m * (x * y); ke[5] = m * (x * z); ke[6] = m * (y * z);
               if (lam != 1.0) { x *= lam; y *= lam; z *= lam; }
               x = fma(hk, fxi, x); y = fma(hk, fyi, y); z = fma(hk, fzi, z);
               fa.vx[a] = x; fa.vy[a] = y; fa.vz[a] = z;
               v2max = __double2float_ru(x * x + y * y + z * z);
               double4 p = pi;
               p.x = fma(fa.dt, x, p.x); p.y = fma(fa.dt, y, p.y); p.z = fma(fa.dt, z, p.z);
               fa.pos_new[a] = p;
            }
            /* the wave's row of kinetic sums (only this wave touches it; the rows are added in index order at the end) */
            double *ke_row = (double *)((char *)smem + fa.ke_off) + (threadIdx.x >> 6) * 8;
            double mine = 0.0;
#pragma unroll
            for (int k = 0; k < 7; k++)
            {
               const double sv = wave_sum_dpp(ke[k]);      /* (uniform over the wave) */
               if (lane == k) mine = sv;
            }
            {
               /* the wave's largest |v|^2 (non-negative floats order like their bit patterns) */
               int vb = __float_as_int(v2max);
               vb = wave_max_dpp(vb);
               if (lane == 7) mine = fmax(ke_row[7], (double)__int_as_float(vb));
            }
            if (lane < 7) ke_row[lane] += mine;      /* one read-modify-write for the seven sums */
            else if (lane == 7) ke_row[7] = mine;
         }
      }
      }
   }
   if (mine)
   {
      /* the tile's LDS doubles as reduction scratch: no static LDS in this kernel, so the
       * staged arrays start at LDS address 0 and need no base add per gather */
      __syncthreads();
      double *s_red = (double *)smem;
      const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
      for (int k = 0; k < 8; k++)
      {
         double sv = wave_sum_dpp(acc[k]);
         if (lane == 0) s_red[w * 8 + k] = sv;
      }
      __syncthreads();
      if (threadIdx.x < 8)
      {
         double a = s_red[threadIdx.x];
#pragma unroll
         for (int q = 1; q < NB_BLOCK / 64; q++) a += s_red[q * 8 + threadIdx.x];
         partials[(size_t)slot * 8 + threadIdx.x] = a;      /* one row per work item */
      }
      if (FUSE && threadIdx.x >= 64 && threadIdx.x < 64 + 8)
      {
         const int k = threadIdx.x - 64;
         const double *ke_s = (const double *)((char *)smem + fa.ke_off);
         double a = 0.0;
         if (nown > 0 && k < 7) for (int q = 0; q < NB_BLOCK / 64; q++) a += ke_s[q * 8 + k];
         if (nown > 0 && k == 7) for (int q = 0; q < NB_BLOCK / 64; q++) a = fmax(a, ke_s[q * 8 + 7]);
         fa.kpartials[(size_t)slot * 8 + k] = a;
      }
   }
}

/* zero forces (nonbonded excluded via excludePotentialTerm) */
__global__ void k_zero3(int n, double *a, double *b, double *c)
{
   int i = blockIdx.x * blockDim.x + threadIdx.x;
   if (i < n) { a[i] = 0; b[i] = 0; c[i] = 0; }
}

/* Fixed-order reductions of per-workgroup partials (8 doubles per row), one workgroup per
 * job: thread (g = tid/8, k = tid%8) adds column k of rows g, g+128, ...; a tree over the
 * 128 row groups finishes all columns at once.  Same order every run => bitwise
 * reproducible.  A step runs its two jobs -- the nonbonded sums (+ the final energies)
 * and the kinetic terms -- in one launch. */
struct RedJob { const double *partials; int nblocks; int nv; double *out; int finish;
                double disp_dt; double *disp; /* disp_dt > 0: column 7 holds max |v|^2 of the step's drift -- reduced by max, then *disp += disp_dt * sqrt(max) (NbTileArgs::disp) */ };
__device__ void finish_energy(double *r, double self_ele);
/* RED_SPLIT workgroups share a job (one workgroup reading the 0.5 MB of per-tile partials of
 * a 4 M-bead box took 21 us); each leaves its 8 sums in tmp, the last one to arrive (ticket)
 * adds the RED_SPLIT rows in index order.  tmp: [2 jobs][RED_SPLIT][8] doubles, then 2 ticket
 * counters (left at zero). */
#define RED_SPLIT 8
__device__ __forceinline__ void reduce_jobs_block(const RedJob &j, const int bx, const int by, double *results, double self_ele, double *tmp)
{
   __shared__ double s[1024];
   __shared__ int s_last;
   const int k = threadIdx.x & 7, g = threadIdx.x >> 3;
   const bool mx = k == 7 && j.disp_dt > 0.0;      /* this thread's column is a maximum (of non-negative numbers) */
   double a = 0.0;
   if (mx)
   {
      for (int b = g + 128 * bx; b < j.nblocks; b += 128 * RED_SPLIT) a = fmax(a, j.partials[(size_t)b * 8 + 7]);
   }
   else if (k < j.nv)
   {
      /* independent partial sums: a single chain of dependent loads is latency-bound */
      double p[4] = {0, 0, 0, 0};
      int b = g + 128 * bx;
      const int stride = 128 * RED_SPLIT;
      for (; b + 3 * stride < j.nblocks; b += 4 * stride)
      {
#pragma unroll
         for (int u = 0; u < 4; u++) p[u] += j.partials[(size_t)(b + u * stride) * 8 + k];
      }
      for (int u = 0; b < j.nblocks; b += stride, u++) p[u] += j.partials[(size_t)b * 8 + k];
      a = (p[0] + p[1]) + (p[2] + p[3]);
   }
   s[threadIdx.x] = a;
   __syncthreads();
   for (int off = 512; off >= 8; off >>= 1)
   {
      if (threadIdx.x < off) s[threadIdx.x] = mx ? fmax(s[threadIdx.x], s[threadIdx.x + off]) : s[threadIdx.x] + s[threadIdx.x + off];
      __syncthreads();
   }
   double *mytmp = tmp + ((size_t)by * RED_SPLIT + bx) * 8;
   unsigned int *ticket = (unsigned int *)(tmp + 2 * RED_SPLIT * 8) + by;
   if (threadIdx.x < 8) { mytmp[threadIdx.x] = s[threadIdx.x]; __threadfence(); }
   __syncthreads();
   if (threadIdx.x == 0)
   {
      unsigned int t = atomicAdd(ticket, 1u);
      s_last = (t == RED_SPLIT - 1);
      if (s_last) { *ticket = 0u; __threadfence(); }
   }
   __syncthreads();
   if (!s_last) return;
   if (threadIdx.x == 7 && j.disp_dt > 0.0)
   {
      const double *row = tmp + (size_t)by * RED_SPLIT * 8 + 7;
      double t = 0.0;
      __threadfence();
#pragma unroll
      for (int q = 0; q < RED_SPLIT; q++) t = fmax(t, row[q * 8]);
      *j.disp += j.disp_dt * sqrt(t) * (1.0 + 1e-7);      /* (rounded up: |v|^2 came as a float rounded up) */
   }
   if (threadIdx.x < (unsigned)j.nv)
   {
      const double *row = tmp + (size_t)by * RED_SPLIT * 8 + threadIdx.x;
      double t = 0.0;
      __threadfence();          /* acquire: the other workgroups' rows, written on other XCDs */
#pragma unroll
      for (int q = 0; q < RED_SPLIT; q++) t += row[q * 8];
      j.out[threadIdx.x] = t;
   }
   if (j.finish)
   {
      __syncthreads();          /* orders the out[] stores before thread 0 reads them */
      if (threadIdx.x == 0) finish_energy(results, self_ele);
   }
}
__global__ __launch_bounds__(1024) void k_reduce_jobs(RedJob j0, RedJob j1, double *results, double self_ele, double *tmp)
{
   reduce_jobs_block(blockIdx.y ? j1 : j0, (int)blockIdx.x, (int)blockIdx.y, results, self_ele, tmp);
}
/* the same two jobs and, in further workgroups of the same launch, the periodic images of a single domain brought up to the positions
 * the fused pair kernel has just drifted to (k_halo_update's self-image arm): both only wait for that kernel, one launch instead of two */
struct ImageJob { int nloc, nhalo; const int *halo_src, *halo_shift; double L0, L1, L2; double4 *pos; };
/* ... or, in a decomposed run, the halo messages packed from the drifted positions (k_pack_halo, width 3) */
struct PackJob { int nsend; const unsigned *send_map; int shift[27][3]; double L0, L1, L2; const double4 *pos; double *out; };
__global__ __launch_bounds__(1024) void k_reduce_jobs_images(RedJob j0, RedJob j1, double *results, double self_ele, double *tmp, ImageJob im, PackJob pk)
{
   const int b = (int)blockIdx.x;
   if (b < 2 * RED_SPLIT) { reduce_jobs_block(b < RED_SPLIT ? j0 : j1, b % RED_SPLIT, b / RED_SPLIT, results, self_ele, tmp); return; }
   const int nimb = (im.nhalo + 1023) / 1024;
   if (b >= 2 * RED_SPLIT + nimb)
   {
      const int k = (b - 2 * RED_SPLIT - nimb) * 1024 + (int)threadIdx.x;
      if (k >= pk.nsend) return;
      const unsigned m = pk.send_map[k];
      const int i = (int)(m & 0x7ffffffu), code = (int)(m >> 27);
      const double4 p = pk.pos[i];
      double *o = pk.out + (size_t)k * 3;
      o[0] = p.x + pk.shift[code][0] * pk.L0;
      o[1] = p.y + pk.shift[code][1] * pk.L1;
      o[2] = p.z + pk.shift[code][2] * pk.L2;
      return;
   }
   const int h = (b - 2 * RED_SPLIT) * 1024 + (int)threadIdx.x;
   if (h >= im.nhalo) return;
   const int src = im.halo_src[h], code = im.halo_shift[h];
   double4 p = im.pos[src];
   p.x += (double)(code % 3 - 1) * im.L0;
   p.y += (double)((code / 3) % 3 - 1) * im.L1;
   p.z += (double)(code / 9 - 1) * im.L2;
   im.pos[im.nloc + h] = p;
}

/* final energies / virial: full list counts every pair twice */
__device__ void finish_energy(double *r, double self_ele)
{
   double lj = 0.5 * r[R_NB_LJ];
   double ele = 0.5 * r[R_NB_ELE] + self_ele;
   r[R_E + DDCMI_E_LJ] = lj;
   r[R_E + DDCMI_E_ELE] = ele;
   /* bonded scratch: bond {e,vir6} angle {e,vir6} tors {e_tors,e_impr,vir6} */
   double eb[4] = {r[R_SCR_BOND], r[R_SCR_ANGLE], r[R_SCR_TORS], r[R_SCR_TORS + 1]};
   double etot = lj + ele;
   for (int k = 0; k < 4; k++) { r[R_E + DDCMI_E_BOND + k] = eb[k]; etot += eb[k]; }
   r[R_E + DDCMI_E_RESTRAINT] = r[R_SCR_REST];
   r[R_E + DDCMI_E_TOTAL] = etot + r[R_SCR_REST];
   for (int k = 0; k < 6; k++)
      r[R_VIR + k] = 0.5 * r[R_NB_VIR + k] + ((r[R_SCR_BOND + 1 + k] + r[R_SCR_ANGLE + 1 + k]) + r[R_SCR_TORS + 2 + k]) + r[R_SCR_REST + 1 + k];
}

__global__ void k_finish_energy(double *r, double self_ele)
{
   if (threadIdx.x == 0 && blockIdx.x == 0) finish_energy(r, self_ele);
}

/* ------------------------------------------------------------------------- */
/* NGLF integrator kernels (nglf.c:67-112)                                    */
/* FRONT half kick (free.c:13-28 / berendsen.c:64-89) fused with the drift
 * (nglf.c:80-87).  The wrap of nglf.c:90 is applied at rebuild/download time
 * instead (positions stay continuous between rebuilds so image atoms and the
 * list remain valid); the downloaded coordinates are identical up to rounding. */
/* per-group data of the velocity updates, by value.  v = Berendsen scale factor of the FRONT
 * kick (1 otherwise); groups in lang_mask use the Langevin update (langevin.c:92-128, vcm = 0):
 *   FRONT  v = a v + c f + d g        BACK  v = a (v + c f + d g)
 * a = exp(-dt_half/tau), c = dt_half/m, d = sqrt(2 dt_half kB T/(m tau)) = dfac/sqrt(m), g = three unit normals.
 * The reference draws g from a per-particle LCG64 stream stored with the particle; here it is a
 * counter-based stream keyed by (seed, gid, 2*loop + BACK): the same numbers whatever the domain
 * decomposition or launch shape -- statistical, not bitwise, parity with ddcMD. */
struct GroupLambda { double v[32]; double a[32]; double dfac[32]; unsigned lang_mask; unsigned long long seed, counter_front, counter_back;
                     double scale[3]; /* barostat: positions are scaled by this (adjustPosn) before the drift; 1 otherwise */
                     unsigned vcm_mask; double vw[32][3]; /* Langevin groups with a drift velocity (langevin.c:106,167 `vcm`): v = vcm + a (v - vcm) + ... adds vw = (1 - a) vcm to either update */
                     ulonglong2 *lcg; /* RANDOM type LCG64 (ddcmi_set_random_lcg64): the beads' own streams (the reference's), in slot order; nullptr = the counter-based stream */ };
__device__ __forceinline__ unsigned long long smix64(unsigned long long z)
{
   z += 0x9E3779B97F4A7C15ull;
   z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
   z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
   return z ^ (z >> 31);
}
__device__ __forceinline__ void gauss3(unsigned long long seed, unsigned long long gid, unsigned long long counter, double &g0, double &g1, double &g2)
{
   const unsigned long long key = smix64(seed ^ smix64(gid)) + 4ull * counter;
   const double two53 = 1.0 / 9007199254740992.0;
   double u1 = ((double)(smix64(key) >> 11) + 0.5) * two53, u2 = ((double)(smix64(key + 1) >> 11) + 0.5) * two53;
   double u3 = ((double)(smix64(key + 2) >> 11) + 0.5) * two53, u4 = ((double)(smix64(key + 3) >> 11) + 0.5) * two53;
   double r = sqrt(-2.0 * log(u1)), t = 6.283185307179586476925 * u2;
   g0 = r * cos(t); g1 = r * sin(t);
   g2 = sqrt(-2.0 * log(u3)) * cos(6.283185307179586476925 * u4);
}
/* The reference's own noise: lcg64_2 (lcg64.c:137-146) under gasdev3d (random.c:135-160) -- two polar draws, x and y from the
 * first accepted pair, z from the second.  A record is LCG64_PARM {state; multID, prime}; it travels with its bead (sorts,
 * migration) and only its state changes.  The products and the sum of rsq are rounded one by one, as the accept test of the CPU code sees them. */
__device__ __forceinline__ void lcg_pair(unsigned long long &st, unsigned long long mult, unsigned long long prime, double &x, double &y, double &rsq)
{
   do
   {
      st = mult * st + prime; x = __dmul_rn((double)st, 5.4210108624275222e-20);
      st = mult * st + prime; y = __dmul_rn((double)st, 5.4210108624275222e-20);
      x = __dadd_rn(__dmul_rn(2.0, x), -1.0); y = __dadd_rn(__dmul_rn(2.0, y), -1.0);
      rsq = __dadd_rn(__dmul_rn(x, x), __dmul_rn(y, y));
   } while (rsq >= 1.0 || rsq == 0.0);
}
__device__ __forceinline__ void lcg_gauss3(ulonglong2 *lcg, int o, double &g0, double &g1, double &g2)
{
   const ulonglong2 q = lcg[o];
   const unsigned id = (unsigned)q.y;
   const unsigned long long mult = id == 0 ? 0x27bb2ee687b0b0fdull : id == 1 ? 0x2c6fe96ee78b6955ull : 0x369dea0f31a53f85ull, prime = q.y >> 32;
   unsigned long long st = q.x;
   double x, y, rsq;
   lcg_pair(st, mult, prime, x, y, rsq);
   double fac = sqrt(-2.0 * log(rsq) / rsq);
   g0 = x * fac; g1 = y * fac;
   lcg_pair(st, mult, prime, x, y, rsq);
   fac = sqrt(-2.0 * log(rsq) / rsq);
   g2 = x * fac;
   lcg[o].x = st;
}
__device__ __forceinline__ void group_gauss3(const GroupLambda &gl, int i, const uint64_t *gid, unsigned long long counter, double &g0, double &g1, double &g2)
{
   if (gl.lcg) lcg_gauss3(gl.lcg, i, g0, g1, g2);
   else gauss3(gl.seed, gid[i], counter, g0, g1, g2);
}
/* the largest |v|^2 of a workgroup's drifting beads, rounded up, into column 7 of its row of partials: the displacement bound of the
 * shell-limited walk (NbTileArgs::disp) adds dt * sqrt(max over the rows) per step (k_reduce_jobs) */
template <int NW>
__device__ __forceinline__ void block_vmax_store(float v2, double *row)
{
   __shared__ int s_vm[NW];
   const int m = wave_max_dpp(__float_as_int(v2));      /* non-negative floats order like their bit patterns */
   if ((threadIdx.x & 63) == 0) s_vm[threadIdx.x >> 6] = m;
   __syncthreads();
   if (threadIdx.x == 0)
   {
      int t = s_vm[0];
#pragma unroll
      for (int q = 1; q < NW; q++) t = max(t, s_vm[q]);
      row[7] = (double)__int_as_float(t);
   }
}
__global__ void k_kick_drift(int nloc, double dt, const double *__restrict__ invmass, const int *__restrict__ species,
                             const int *__restrict__ group, GroupLambda glambda, const uint64_t *__restrict__ gid,
                             const double *__restrict__ fx, const double *__restrict__ fy, const double *__restrict__ fz,
                             double *__restrict__ vx, double *__restrict__ vy, double *__restrict__ vz, double4 *__restrict__ pos, int mode,
                             double *__restrict__ vpart /* mode 3, not null: [block][8], column 7 = the block's largest |v|^2 of the drift */)
{
   const int i = blockIdx.x * blockDim.x + threadIdx.x;
   float v2 = 0.0f;
   if (i < nloc)
   {
      bool done = false;
      if (mode != 3)
      {
         /* nglfconstraint splits the pass around the FRONT constraint solve: mode 1 = barostat scaling of the
          * positions (adjustPosn) + kick, mode 2 = drift with the constrained velocities */
         double4 p = pos[i];
         if (mode == 2) { p.x = fma(dt, vx[i], p.x); p.y = fma(dt, vy[i], p.y); p.z = fma(dt, vz[i], p.z); pos[i] = p; done = true; }
         else if (glambda.scale[0] != 1.0 || glambda.scale[1] != 1.0 || glambda.scale[2] != 1.0)
         { p.x *= glambda.scale[0]; p.y *= glambda.scale[1]; p.z *= glambda.scale[2]; pos[i] = p; }
      }
      if (!done)
      {
         const double im = invmass[species[i]];
         double a = (0.5 * dt) * im;
         const int gr = group[i] & 31;
         double lam = glambda.v[gr];
         double x = vx[i], y = vy[i], z = vz[i];
         if (glambda.lang_mask >> gr & 1u)
         {
            double g0, g1, g2, d = glambda.dfac[gr] * sqrt(im), al = glambda.a[gr];
            group_gauss3(glambda, i, gid, glambda.counter_front, g0, g1, g2);
            x = fma(d, g0, fma(a, fx[i], al * x)); y = fma(d, g1, fma(a, fy[i], al * y)); z = fma(d, g2, fma(a, fz[i], al * z));
            if (glambda.vcm_mask >> gr & 1u) { x += glambda.vw[gr][0]; y += glambda.vw[gr][1]; z += glambda.vw[gr][2]; }
         }
         else
         {
            if (lam != 1.0) { x *= lam; y *= lam; z *= lam; }
            /* explicit fma: k_kick_ke_drift must produce the same bits as this kernel */
            x = fma(a, fx[i], x); y = fma(a, fy[i], y); z = fma(a, fz[i], z);
         }
         vx[i] = x; vy[i] = y; vz[i] = z;
         if (mode == 3)
         {
            double4 p = pos[i];
            p.x = fma(dt, x, glambda.scale[0] * p.x); p.y = fma(dt, y, glambda.scale[1] * p.y); p.z = fma(dt, z, glambda.scale[2] * p.z);
            pos[i] = p;
            v2 = __double2float_ru(x * x + y * y + z * z);
         }
      }
   }
   if (vpart) block_vmax_store<4>(v2, vpart + (size_t)blockIdx.x * 8);      /* (uniform: every thread of the block gets here) */
}
__global__ void k_scale_pos(int n, double s0, double s1, double s2, double4 *pos)
{
   int i = blockIdx.x * blockDim.x + threadIdx.x;
   if (i >= n) return;
   double4 p = pos[i];
   p.x *= s0; p.y *= s1; p.z *= s2;
   pos[i] = p;
}
#define KE_PER 4
/* BACK half kick (nglf.c:100-104) fused with kinetic_terms (energy.c:48-163):
 * rk = sum 1/2 m v^2, tion = sum m v (x) v */
__global__ __launch_bounds__(DDCMI_BLOCK) void k_kick_ke(int nloc, double dt, const double *__restrict__ invmass, const double *__restrict__ massv,
                                                         const int *__restrict__ species,
                                                         const double *__restrict__ fx, const double *__restrict__ fy, const double *__restrict__ fz,
                                                         double *__restrict__ vx, double *__restrict__ vy, double *__restrict__ vz,
                                                         double *__restrict__ partials, int do_kick,
                                                         const int *__restrict__ group, GroupLambda glambda, const uint64_t *__restrict__ gid)
{
   double acc[7] = {0, 0, 0, 0, 0, 0, 0};
   /* KE_PER beads per thread: a quarter of the partial rows for the reduction launch to read */
#pragma unroll
   for (int u = 0; u < KE_PER; u++)
   {
      int i = (blockIdx.x * KE_PER + u) * DDCMI_BLOCK + threadIdx.x;
      if (i >= nloc) continue;
      int sp = species[i];
      double x = vx[i], y = vy[i], z = vz[i];
      if (do_kick)
      {
         const double im = invmass[sp];
         double a = (0.5 * dt) * im;
         const int gr = group[i] & 31;
         if (glambda.lang_mask >> gr & 1u)
         {
            double g0, g1, g2, d = glambda.dfac[gr] * sqrt(im), al = glambda.a[gr];
            group_gauss3(glambda, i, gid, glambda.counter_back, g0, g1, g2);
            x = al * fma(d, g0, fma(a, fx[i], x)); y = al * fma(d, g1, fma(a, fy[i], y)); z = al * fma(d, g2, fma(a, fz[i], z));
            if (glambda.vcm_mask >> gr & 1u) { x += glambda.vw[gr][0]; y += glambda.vw[gr][1]; z += glambda.vw[gr][2]; }
         }
         else { x = fma(a, fx[i], x); y = fma(a, fy[i], y); z = fma(a, fz[i], z); }
         vx[i] = x; vy[i] = y; vz[i] = z;
      }
      double m = massv[sp];
      double vxx = x * x, vyy = y * y, vzz = z * z;
      acc[0] += 0.5 * m * (vxx + vyy + vzz);
      acc[1] += m * vxx; acc[2] += m * vyy; acc[3] += m * vzz;
      acc[4] += m * (x * y); acc[5] += m * (x * z); acc[6] += m * (y * z);
   }
   block_reduce_store<7>(acc, partials + (size_t)blockIdx.x * 8);
}
/* The BACK half kick + kinetic terms of step n and the FRONT half kick + drift of step
 * n+1 use the same forces: inside a batch of steps they are one pass over v and f
 * (k_kick_ke followed by k_kick_drift, bit for bit). */
__global__ __launch_bounds__(DDCMI_BLOCK) void k_kick_ke_drift(int nloc, double dt, const double *__restrict__ invmass, const double *__restrict__ massv,
                                                               const int *__restrict__ species, const int *__restrict__ group, GroupLambda glambda,
                                                               const double *__restrict__ fx, const double *__restrict__ fy, const double *__restrict__ fz,
                                                               double *__restrict__ vx, double *__restrict__ vy, double *__restrict__ vz,
                                                               double4 *__restrict__ pos, double *__restrict__ partials, const uint64_t *__restrict__ gid)
{
   double acc[7] = {0, 0, 0, 0, 0, 0, 0};
   float v2 = 0.0f;
#pragma unroll
   for (int u = 0; u < KE_PER; u++)
   {
      int i = (blockIdx.x * KE_PER + u) * DDCMI_BLOCK + threadIdx.x;
      if (i >= nloc) continue;
      int sp = species[i];
      const double im = invmass[sp];
      double a = (0.5 * dt) * im;
      double f0 = fx[i], f1 = fy[i], f2 = fz[i];
      const int gr = group[i] & 31;
      const bool lang = glambda.lang_mask >> gr & 1u;
      double x, y, z, g0, g1, g2, dl = 0.0, al = 0.0;
      if (lang)
      {
         dl = glambda.dfac[gr] * sqrt(im); al = glambda.a[gr];
         group_gauss3(glambda, i, gid, glambda.counter_back, g0, g1, g2);
         x = al * fma(dl, g0, fma(a, f0, vx[i])); y = al * fma(dl, g1, fma(a, f1, vy[i])); z = al * fma(dl, g2, fma(a, f2, vz[i]));
         if (glambda.vcm_mask >> gr & 1u) { x += glambda.vw[gr][0]; y += glambda.vw[gr][1]; z += glambda.vw[gr][2]; }
      }
      else { x = fma(a, f0, vx[i]); y = fma(a, f1, vy[i]); z = fma(a, f2, vz[i]); }
      double m = massv[sp];
      double vxx = x * x, vyy = y * y, vzz = z * z;
      acc[0] += 0.5 * m * (vxx + vyy + vzz);
      acc[1] += m * vxx; acc[2] += m * vyy; acc[3] += m * vzz;
      acc[4] += m * (x * y); acc[5] += m * (x * z); acc[6] += m * (y * z);
      if (lang)
      {
         group_gauss3(glambda, i, gid, glambda.counter_front, g0, g1, g2);
         x = fma(dl, g0, fma(a, f0, al * x)); y = fma(dl, g1, fma(a, f1, al * y)); z = fma(dl, g2, fma(a, f2, al * z));
         if (glambda.vcm_mask >> gr & 1u) { x += glambda.vw[gr][0]; y += glambda.vw[gr][1]; z += glambda.vw[gr][2]; }
      }
      else
      {
         double lam = glambda.v[gr];
         if (lam != 1.0) { x *= lam; y *= lam; z *= lam; }
         x = fma(a, f0, x); y = fma(a, f1, y); z = fma(a, f2, z);
      }
      vx[i] = x; vy[i] = y; vz[i] = z;
      v2 = fmaxf(v2, __double2float_ru(x * x + y * y + z * z));
      double4 p = pos[i];
      p.x = fma(dt, x, glambda.scale[0] * p.x); p.y = fma(dt, y, glambda.scale[1] * p.y); p.z = fma(dt, z, glambda.scale[2] * p.z);
      pos[i] = p;
   }
   block_reduce_store<7>(acc, partials + (size_t)blockIdx.x * 8);
   block_vmax_store<DDCMI_BLOCK / 64>(v2, partials + (size_t)blockIdx.x * 8);      /* column 7: the displacement bound's share of this drift */
}
/* neighborCheck (neighbor.c:117-208), constant box: displacement of every owned bead since the
 * list was built, measured relative to the centroid of the domain's beads (positions are not
 * wrapped between rebuilds here, so r - r0 needs no image logic).  Pass 1: sum of r - r0. */
__global__ __launch_bounds__(DDCMI_BLOCK) void k_disp_sum(int nloc, const double4 *__restrict__ pos, const double4 *__restrict__ pos0, double *__restrict__ partials)
{
   int i = blockIdx.x * blockDim.x + threadIdx.x;
   double acc[3] = {0, 0, 0};
   if (i < nloc) { double4 p = pos[i], q = pos0[i]; acc[0] = p.x - q.x; acc[1] = p.y - q.y; acc[2] = p.z - q.z; }
   block_reduce_store<3>(acc, partials + (size_t)blockIdx.x * 8);
}
/* pass 2: max_i |(r_i - r0_i) - mean|^2; non-negative doubles order like their bit patterns */
__global__ __launch_bounds__(DDCMI_BLOCK) void k_disp_max(int nloc, const double4 *__restrict__ pos, const double4 *__restrict__ pos0, const double *__restrict__ sum, unsigned long long *out)
{
   int i = blockIdx.x * blockDim.x + threadIdx.x;
   double d2 = 0.0;
   if (i < nloc)
   {
      double inv = 1.0 / (double)nloc;
      double4 p = pos[i], q = pos0[i];
      double x = (p.x - q.x) - sum[0] * inv, y = (p.y - q.y) - sum[1] * inv, z = (p.z - q.z) - sum[2] * inv;
      d2 = x * x + y * y + z * z;
   }
#pragma unroll
   for (int off = 32; off > 0; off >>= 1) d2 = fmax(d2, __shfl_down(d2, off, 64));
   if ((threadIdx.x & 63) == 0 && d2 > 0.0) atomicMax(out, (unsigned long long)__double_as_longlong(d2));
}
/* per-group kinetic energy and member count (energy.c:124-133) */
#define GKE_BLOCKS 512
/* per-group kinetic energy and bead count: GKE_BLOCKS workgroups stride over the beads and leave one
 * partial pair per group; k_group_ke_sum adds them in a fixed order */
__global__ __launch_bounds__(DDCMI_BLOCK) void k_group_ke(int nloc, int ngroup, const double *__restrict__ massv, const int *__restrict__ species,
                                                          const int *__restrict__ group,
                                                          const double *__restrict__ vx, const double *__restrict__ vy, const double *__restrict__ vz,
                                                          double *partials /* [GKE_BLOCKS][2*ngroup] */)
{
   __shared__ double s_red[DDCMI_BLOCK / 64][2];
   for (int g = 0; g < ngroup; g++)
   {
      double k = 0.0, c = 0.0;
      for (int i = blockIdx.x * DDCMI_BLOCK + threadIdx.x; i < nloc; i += GKE_BLOCKS * DDCMI_BLOCK)
         if (group[i] == g)
         {
            double m = massv[species[i]];
            k += 0.5 * m * (vx[i] * vx[i] + vy[i] * vy[i] + vz[i] * vz[i]);
            c += 1.0;
         }
      k = wave_sum(k); c = wave_sum(c);
      if ((threadIdx.x & 63) == 0) { s_red[threadIdx.x >> 6][0] = k; s_red[threadIdx.x >> 6][1] = c; }
      __syncthreads();
      if (threadIdx.x < 2)
      {
         double a = 0.0;
         for (int w = 0; w < DDCMI_BLOCK / 64; w++) a += s_red[w][threadIdx.x];
         partials[(size_t)blockIdx.x * 2 * ngroup + 2 * g + threadIdx.x] = a;
      }
      __syncthreads();
   }
}
__global__ void k_group_ke_sum(int ngroup, const double *__restrict__ partials, double *out)
{
   int q = threadIdx.x;
   if (q >= 2 * ngroup) return;
   double a = 0.0;
   for (int b = 0; b < GKE_BLOCKS; b++) a += partials[(size_t)b * 2 * ngroup + q];
   out[q] = a;
}

/* The per-group and per-species copies of kinetic_terms (energy.c:104-147) and the thermal flux: for every class c (a
 * group or a species) {rk, tion xx yy zz xy xz yz, mass, number, J x y z} with J_k = (K_k + U_k) v_k - 1/2 S_k v_k; on this
 * path the per-atom potential energy U_k and stress S_k are zero (martiniNonBond and the bonded terms book e->eion and
 * e->virial only, bioMartini.c:1111-1120), so J = sum K v and a class's eion stays 0.  Read at print steps only: one pass
 * per class, fixed-order sums (bitwise reproducible) like k_group_ke. */
#define KD_NV 12
__global__ __launch_bounds__(DDCMI_BLOCK) void k_class_kinetic(int nloc, int nclass, int by_species, const double *__restrict__ massv, const int *__restrict__ species,
                                                                const int *__restrict__ group,
                                                                const double *__restrict__ vx, const double *__restrict__ vy, const double *__restrict__ vz,
                                                                double *partials /* [GKE_BLOCKS][nclass][16] */)
{
   for (int c = 0; c < nclass; c++)
   {
      double acc[KD_NV] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
      for (int i = blockIdx.x * DDCMI_BLOCK + threadIdx.x; i < nloc; i += GKE_BLOCKS * DDCMI_BLOCK)
      {
         const int sp = species[i];
         if ((by_species ? sp : group[i]) != c) continue;
         const double m = massv[sp], x = vx[i], y = vy[i], z = vz[i];
         const double K = 0.5 * m * (x * x + y * y + z * z);
         acc[0] += K;
         acc[1] += m * (x * x); acc[2] += m * (y * y); acc[3] += m * (z * z);
         acc[4] += m * (x * y); acc[5] += m * (x * z); acc[6] += m * (y * z);
         acc[7] += m; acc[8] += 1.0;
         acc[9] += K * x; acc[10] += K * y; acc[11] += K * z;
      }
      block_reduce_store<KD_NV>(acc, partials + ((size_t)blockIdx.x * nclass + c) * 16);
      __syncthreads();      /* the reduction's scratch is reused by the next class */
   }
}
__global__ void k_class_kinetic_sum(int nclass, const double *__restrict__ partials, double *out)
{
   const int q = blockIdx.x * blockDim.x + threadIdx.x;
   if (q >= nclass * KD_NV) return;
   const int c = q / KD_NV, k = q % KD_NV;
   double a = 0.0;
   for (int b = 0; b < GKE_BLOCKS; b++) a += partials[((size_t)b * nclass + c) * 16 + k];
   out[q] = a;
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
/* record tag w (bit-cast into pos.w): [63:32] molecule id (gid>>32, bioGid.h) [31:16] species
 * [15:8] atom-in-molecule code (gid & 0xffff when < 255, else 255 = "look at the gid")
 * [7:0] LJ type.  The atom code lets the list build decide bonded-pair exclusions from
 * LDS instead of two dependent global loads per same-molecule candidate. */
__global__ void k_init_state(int n, const double *rx, const double *ry, const double *rz, const int *species, const int *ljtype_sp,
                             const uint64_t *gid, double4 *pos, int *orig, int *slot)
{
   int i = blockIdx.x * blockDim.x + threadIdx.x;
   if (i >= n) return;
   int sp = species[i];
   unsigned code = (unsigned)(gid[i] & DDCMI_GID_ATMGRPMASK);
   long long w = (long long)(gid[i] & DDCMI_GID_MOLMASK) | ((long long)(sp & 0xffff) << 16) | ((long long)min(code, 255u) << 8) | (long long)(ljtype_sp[sp] & 0xff);
   pos[i] = make_double4(rx[i], ry[i], rz[i], __longlong_as_double(w));
   orig[i] = i;
   slot[i] = i;
}
/* test/inspection export: decode the tile ELL (16-bit staged indices) into CSR
 * over caller-order indices; image atoms map back to their source bead */
__global__ void k_tilelist_to_csr(NbTileArgs ta, int pack_type, int nloc, const int *orig, const int *halo_src, const int *start, int *jout)
{
   int t = blockIdx.x;
   int ts = ta.cell_start_o[TCELLS * t];
   int nown = ta.cell_start_o[TCELLS * t + TCELLS] - ts;
   if (nown <= 0) return;
   const int *sidx = ta.stage_idx + (size_t)t * ta.stage_stride;
   long long base = ta.tile_base[t];
   int rows = ta.tile_rows[t];
   for (int al = threadIdx.x; al < nown; al += blockDim.x)
   {
      int a = ts + al;
      int s = start[orig[a]];
      int cnt = ta.nbr_cnt[a];
      for (int k = 0; k < cnt; k++)
      {
         int ee = ta.nbr16[base + ((size_t)(k >> 3) * rows + al) * 8 + (k & 7)];
         int j = sidx[(pack_type ? (ee >> 4) : ee) - 1];      /* (pack_type 2: the nibble's shift bit is not part of the slot) */
         if (j >= nloc) j = halo_src[j - nloc];
         jout[s + k] = orig[j];
      }
   }
}
/* the same for the excluded (same-molecule bonded) pairs: their list-format entries (excl16) name staged slots of the bead's tile */
__global__ void k_tileexcl_to_csr(NbTileArgs ta, int pack_type, int nloc, int npad, const unsigned short *excl16, const int *excl_cnt,
                                  const int *orig, const int *halo_src, const int *start, int *jout)
{
   int t = blockIdx.x;
   int ts = ta.cell_start_o[TCELLS * t];
   int nown = ta.cell_start_o[TCELLS * t + TCELLS] - ts;
   if (nown <= 0) return;
   const int *sidx = ta.stage_idx + (size_t)t * ta.stage_stride;
   for (int al = threadIdx.x; al < nown; al += blockDim.x)
   {
      int a = ts + al;
      int s = start[orig[a]];
      int cnt = excl_cnt[a];
      for (int k = 0; k < cnt; k++)
      {
         int ee = excl16[(size_t)k * npad + a];
         int j = sidx[(pack_type ? (ee >> 4) : ee) - 1];
         if (j >= nloc) j = halo_src[j - nloc];
         jout[s + k] = orig[j];
      }
   }
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
   ctx->no_shell_skip = getenv("DDCMI_NO_SHELL_SKIP") != nullptr;
   ctx->no_image_hint = getenv("DDCMI_NO_IMAGE_HINT") != nullptr;
   /* test hook, armed only together with DDCMI_DEBUG_HOOKS=1 (a stray value alone does nothing; read per context: a test sets it between two of them) */
   ctx->debug_image_bound = (getenv("DDCMI_DEBUG_HOOKS") && getenv("DDCMI_DEBUG_IMAGE_BOUND")) ? atoi(getenv("DDCMI_DEBUG_IMAGE_BOUND")) : 0;
   /* the small host-side count arrays inside the context (migration / halo counts) become DMA targets */
   ctx->self_pinned = hipHostRegister(ctx, sizeof(ddcmi_ctx), hipHostRegisterDefault) == hipSuccess;
   if (!ctx->self_pinned) (void)hipGetLastError();
   if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess ||
       hipMalloc((void **)&ctx->d_results, R_SIZE * sizeof(double)) != hipSuccess ||
       hipHostMalloc((void **)&ctx->h_results, R_SIZE * sizeof(double), hipHostMallocDefault) != hipSuccess ||
       hipMalloc((void **)&ctx->d_flags, DDCMI_NFLAGS * sizeof(int)) != hipSuccess ||
       hipHostMalloc((void **)&ctx->h_flags, 64 * sizeof(int), hipHostMallocDefault) != hipSuccess)
   {
      g_create_err = "context allocation failed";
      if (ctx->self_pinned) (void)hipHostUnregister(ctx);
      delete ctx;
      return DDCMI_ENOMEM;
   }
   (void)hipMemset(ctx->d_results, 0, R_SIZE * sizeof(double));
   (void)hipMemset(ctx->d_flags, 0, DDCMI_NFLAGS * sizeof(int));
   if (ctx->red_tmp.ensure(2 * RED_SPLIT * 8 + 8)) { g_create_err = "context allocation failed"; if (ctx->self_pinned) (void)hipHostUnregister(ctx); delete ctx; return DDCMI_ENOMEM; }
   (void)hipMemset(ctx->red_tmp.p, 0, (2 * RED_SPLIT * 8 + 8) * sizeof(double));      /* incl. the two ticket counters */
   (void)hipDeviceSynchronize();      /* null-stream memsets are not ordered with the context's non-blocking stream */
   memset(ctx->h_results, 0, R_SIZE * sizeof(double));
   { const char *ov = getenv("DDCMI_HALO_OVERLAP"); ctx->halo_overlap = (ov && atoi(ov) != 0); }
   { const char *gv = getenv("DDCMI_GRAPH_MAX_BEADS"); if (gv) ctx->graph_max_beads = atoi(gv); }      /* 0 switches the step graph off */
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
   if (ctx->ph_on > 0)
      for (int k = 0; k < 32; k++)
         if (ctx->ph_cnt[k]) fprintf(stderr, "ddcmi phase %2d %-28s %8.1f us x %ld\n", k, ctx->ph_name[k], ctx->ph_sum[k] / ctx->ph_cnt[k], ctx->ph_cnt[k]);
   ddcmi_comm_destroy(ctx);
   dbuf<double> *db[] = {&ctx->d_invmass, &ctx->d_mass, &ctx->d_charge_sp, &ctx->bpartials, &ctx->vx, &ctx->vy, &ctx->vz, &ctx->vx2, &ctx->vy2, &ctx->vz2,
                         &ctx->fx, &ctx->fy, &ctx->fz, &ctx->d_kqtab, &ctx->partials};
   for (auto b : db) b->release();
   dbuf<int> *ib[] = {&ctx->d_ljtype_sp, &ctx->d_moltype_sp, &ctx->d_mol_nspecies, &ctx->d_bpair_off, &ctx->d_bpairI, &ctx->d_bpairJ, &ctx->species, &ctx->species2,
                      &ctx->group, &ctx->group2, &ctx->orig, &ctx->orig2, &ctx->slot_of_orig, &ctx->cid, &ctx->crank, &ctx->order, &ctx->cell_cnt_o, &ctx->cell_start_o,
                      &ctx->cell_cnt_h, &ctx->cell_start_h, &ctx->cell_start, &ctx->cell_cnt, &ctx->nimg, &ctx->img_off, &ctx->hsrc_t, &ctx->hshift_t, &ctx->hcid, &ctx->hrank,
                      &ctx->horder, &ctx->halo_src, &ctx->halo_shift, &ctx->scan_tmp, &ctx->nbr_cnt, &ctx->excl, &ctx->excl_cnt,
                      &ctx->stage_idx, &ctx->tile_nstage, &ctx->tile_width, &ctx->tile_rows, &ctx->tile_work, &ctx->sched, &ctx->tile_perm};
   for (auto b : ib) b->release();
   ctx->pos.release(); ctx->pos2.release(); ctx->d_ljtab.release(); ctx->gid.release(); ctx->gid2.release();
   ctx->d_exmask.release(); ctx->rest_gid.release(); ctx->rest_fc.release(); ctx->rest_slot.release(); ctx->rest_r0.release(); ctx->rest_kb.release(); ctx->pos0.release(); ctx->disp.release(); ctx->atom_gid.release(); ctx->hkeys.release();
   for (auto b : {&ctx->cg_dist, &ctx->inc_bpar, &ctx->inc_apar, &ctx->inc_tpar}) b->release();
   for (auto b : {&ctx->cg_atom_off, &ctx->cg_atoms, &ctx->cg_pair_off, &ctx->cons_status, &ctx->mol_off, &ctx->mol_atoms}) b->release();
   ctx->cg_pa.release(); ctx->cg_pb.release();
   for (auto b : {&ctx->inc_boff, &ctx->inc_aoff, &ctx->inc_toff, &ctx->inc_brow, &ctx->inc_arow, &ctx->inc_trow, &ctx->inc_haoff, &ctx->inc_harow, &ctx->inc_hatoms, &ctx->inc_latoms, &ctx->slot_of_atom, &ctx->hvals}) b->release();
   ctx->tile_nib.release();
   ctx->tile_base.release(); ctx->nbr16.release(); ctx->excl16.release(); ctx->kpartials.release(); ctx->red_tmp.release(); ctx->tmp32.release();
   for (auto &e : ctx->ev) (void)hipEventDestroy(e);
   if (ctx->ev_drift) (void)hipEventDestroy(ctx->ev_drift);
   if (ctx->ev_halo) (void)hipEventDestroy(ctx->ev_halo);
   if (ctx->ev_build) (void)hipEventDestroy(ctx->ev_build);
   if (ctx->stream2) { (void)hipStreamSynchronize(ctx->stream2); (void)hipStreamDestroy(ctx->stream2); }
   if (ctx->stream_post) { (void)hipStreamSynchronize(ctx->stream_post); (void)hipStreamDestroy(ctx->stream_post); }
   if (ctx->d_results) (void)hipFree(ctx->d_results);
   if (ctx->h_results) (void)hipHostFree(ctx->h_results);
   if (ctx->d_flags) (void)hipFree(ctx->d_flags);
   if (ctx->graph_exec) (void)hipGraphExecDestroy(ctx->graph_exec);
   if (ctx->h_flags) (void)hipHostFree(ctx->h_flags);
   for (int k = 0; k < 3; k++) if (ctx->h_pin[k]) (void)hipHostFree(ctx->h_pin[k]);
   if (ctx->mbox_h) (void)hipHostFree(ctx->mbox_h);
   if (ctx->agree_h) (void)hipHostFree(ctx->agree_h);
   (void)hipStreamDestroy(ctx->stream);
   if (ctx->self_pinned) (void)hipHostUnregister(ctx);
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
   ctx->tables_dirty = true;
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
   ctx->list_valid = false;
   ctx->tables_dirty = true;
   return DDCMI_OK;
}

/* Pair tables of the nonbonded kernel.  The kernel's bead "type" (low byte of the record tag,
 * nibble of a list entry) is the class (LJ type, charge) of the species: table entry
 * [a*nnb + b] = {sigma^2, 4 eps, shift, 24 eps} of the two LJ types (the reference indexes
 * sj + nspecies*si, bioMartini.c:1052, on a symmetric table) and ke/eps_r q_a q_b. */
__global__ void k_retag(int n, const int *species, const int *nb_of_sp, double4 *pos)
{
   int i = blockIdx.x * blockDim.x + threadIdx.x;
   if (i >= n) return;
   long long w = __double_as_longlong(pos[i].w);
   pos[i].w = __longlong_as_double((w & ~0xffll) | (long long)(nb_of_sp[species[i]] & 0xff));
}
static int nb_tables(ddcmi_ctx *ctx)
{
   if (!ctx->tables_dirty) return DDCMI_OK;
   if (ctx->nlj <= 0 || ctx->nspecies <= 0) SETERR(ctx, DDCMI_EINVAL, "ddcmi_set_species and ddcmi_set_nonbonded must both be called before the state is uploaded");
   std::vector<int> nb(ctx->nspecies), cls_lj;
   std::vector<double> cls_q;
   for (int s = 0; s < ctx->nspecies; s++)
   {
      if (ctx->ljtype[s] < 0 || ctx->ljtype[s] >= ctx->nlj) SETERR(ctx, DDCMI_EINVAL, "species %d: LJ type %d outside the %d x %d table", s, ctx->ljtype[s], ctx->nlj, ctx->nlj);
      int c = -1;
      for (size_t k = 0; k < cls_lj.size(); k++) if (cls_lj[k] == ctx->ljtype[s] && cls_q[k] == ctx->charge[s]) { c = (int)k; break; }
      if (c < 0) { c = (int)cls_lj.size(); cls_lj.push_back(ctx->ljtype[s]); cls_q.push_back(ctx->charge[s]); }
      nb[s] = c;
   }
   const int nnb = (int)cls_lj.size();
   if (nnb > 255) SETERR(ctx, DDCMI_EUNSUPPORTED, "%d (LJ type, charge) classes: more than the 255 the record tag holds", nnb);
   std::vector<double4> tab((size_t)nnb * nnb);
   std::vector<double> kq((size_t)nnb * nnb);
   for (int a = 0; a < nnb; a++)
      for (int b = 0; b < nnb; b++)
      {
         int k = cls_lj[a] * ctx->nlj + cls_lj[b];
         tab[(size_t)a * nnb + b] = make_double4(ctx->sigma[k] * ctx->sigma[k], 4.0 * ctx->eps[k], ctx->shift[k], 24.0 * ctx->eps[k]);
         kq[(size_t)a * nnb + b] = ctx->keR * cls_q[a] * cls_q[b];
      }
   int rc;
   if ((rc = upload_vec(ctx, ctx->d_ljtab, tab.data(), tab.size())) || (rc = upload_vec(ctx, ctx->d_kqtab, kq.data(), kq.size())) ||
       (rc = upload_vec(ctx, ctx->d_ljtype_sp, nb.data(), nb.size()))) return rc;
   ctx->nnb = nnb;
   ctx->tables_dirty = false;
   if (ctx->nloc > 0 && ctx->pos.p)      /* parameters changed under an uploaded state: refresh the tags */
      hipLaunchKernelGGL(k_retag, dim3(cdiv(ctx->nloc, 256)), dim3(256), 0, ctx->stream, ctx->nloc, ctx->species.p, ctx->d_ljtype_sp.p, ctx->pos.p);
   ctx->list_valid = false;
   return DDCMI_OK;
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
   /* bonded-pair masks for the list build: usable for a molecule type whose pair codes are all < 63 */
   std::vector<unsigned long long> em((size_t)nmoltype * 64, 0ull);
   for (int m = 0; m < nmoltype; m++)
   {
      bool ok = true;
      for (int k = bpair_off[m]; k < bpair_off[m + 1]; k++) if (bpairI[k] < 0 || bpairJ[k] < 0 || bpairI[k] >= 63 || bpairJ[k] >= 63) ok = false;
      if (!ok) continue;
      for (int k = bpair_off[m]; k < bpair_off[m + 1]; k++)
      {
         em[(size_t)m * 64 + bpairI[k]] |= 1ull << bpairJ[k];
         em[(size_t)m * 64 + bpairJ[k]] |= 1ull << bpairI[k];
      }
      em[(size_t)m * 64] |= 1ull << 63;
   }
   if ((rc = upload_vec(ctx, ctx->d_exmask, em.data(), em.size()))) return rc;
   return DDCMI_OK;
}

extern "C" int ddcmi_set_neighbor(ddcmi_ctx *ctx, double deltaR, int updateRate)
{
   if (!ctx || deltaR < 0) return DDCMI_EINVAL;
   if (updateRate < 0) SETERR(ctx, DDCMI_EINVAL, "updateRate must be >= 0 (0 = rebuild when neighborCheck says so, ddcUpdateAll.c:64-71)");
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
      if (type[g] != DDCMI_FREE && type[g] != DDCMI_BERENDSEN && type[g] != DDCMI_LANGEVIN) SETERR(ctx, DDCMI_EUNSUPPORTED, "group %d: only FREE, BERENDSEN and LANGEVIN groups are supported", g);
      if (type[g] == DDCMI_LANGEVIN && (!tau || !(tau[g] > 0.0) || !Teq)) SETERR(ctx, DDCMI_EINVAL, "group %d: LANGEVIN needs Teq and tau > 0", g);
      if (Teq) ctx->gTeq[g] = Teq[g];
      if (tau) ctx->gtau[g] = tau[g];
      if (interval && interval[g] > 0) ctx->ginterval[g] = interval[g];
   }
   ctx->glambda.assign(ngroup, 1.0); ctx->gTsum.assign(ngroup, 0.0); ctx->gT.assign(ngroup, 0.0);
   ctx->gnT.assign(ngroup, 0); ctx->gdoScaling.assign(ngroup, 0);
   ctx->gvcm.clear();
   return DDCMI_OK;
}

static void graph_drop(ddcmi_ctx *ctx);
extern "C" int ddcmi_set_group_vcm(ddcmi_ctx *ctx, int ngroup, const double *vcm)
{
   if (!ctx || ngroup < 0 || ngroup > 32 || (ngroup > 0 && !vcm)) return DDCMI_EINVAL;
   if (ngroup != ctx->ngroup) SETERR(ctx, DDCMI_EINVAL, "ddcmi_set_group_vcm: %d groups, ddcmi_set_groups gave %d", ngroup, ctx->ngroup);
   ctx->gvcm.assign(vcm, vcm + 3 * (size_t)ngroup);
   graph_drop(ctx);
   return DDCMI_OK;
}
extern "C" int ddcmi_set_group_temperature(ddcmi_ctx *ctx, int group, double Teq)
{
   if (!ctx || group < 0 || group >= ctx->ngroup || !(Teq >= 0.0)) return DDCMI_EINVAL;
   ctx->gTeq[group] = Teq;      /* (host scalars: the next step's factors are formed from them) */
   graph_drop(ctx);
   return DDCMI_OK;
}
extern "C" int ddcmi_set_barostat(ddcmi_ctx *ctx, double T, double P0, double beta, double tau)
{
   if (!ctx || beta < 0.0 || (beta > 0.0 && !(tau > 0.0))) return DDCMI_EINVAL;
   if (beta > 0.0)
   {
      if (ctx->nranks > 1 || ctx->group_) SETERR(ctx, DDCMI_EUNSUPPORTED, "the barostat is implemented for a single domain");
   }
   ctx->baro_T = T; ctx->baro_P0 = P0; ctx->baro_beta = beta; ctx->baro_tau = tau;
   return DDCMI_OK;
}
extern "C" int ddcmi_get_box(const ddcmi_ctx *ctx, double h[9])
{
   if (!ctx || !h) return DDCMI_EINVAL;
   for (int k = 0; k < 9; k++) h[k] = ctx->h[k];
   return DDCMI_OK;
}
extern "C" int ddcmi_set_barostat_isotropic(ddcmi_ctx *ctx, int on)
{
   if (!ctx) return DDCMI_EINVAL;
   ctx->baro_iso = on != 0;
   return DDCMI_OK;
}
extern "C" int ddcmi_get_barostat_pressure(const ddcmi_ctx *ctx, double p[3])
{
   if (!ctx || !p) return DDCMI_EINVAL;
   for (int k = 0; k < 3; k++) p[k] = ctx->pmol[k];
   return DDCMI_OK;
}

extern "C" int ddcmi_set_random(ddcmi_ctx *ctx, uint64_t seed)
{
   if (!ctx) return DDCMI_EINVAL;
   ctx->rng_seed = seed;
   return DDCMI_OK;
}

/* RANDOM type LCG64: the particles' own streams (LCG64_PARM records in the caller order of ddcmi_upload_state).  On the device
 * they lie in slot order like every other per-bead array and move with the beads (k_gather_state, the migration records). */
static inline bool lcg_decomposed(const ddcmi_ctx *ctx) { return ctx->nranks > 1 || ctx->loopback || ctx->group_ != nullptr; }
extern "C" int ddcmi_set_random_lcg64(ddcmi_ctx *ctx, int n, const uint64_t *state, const uint32_t *multID, const uint32_t *prime)
{
   if (!ctx) return DDCMI_EINVAL;
   (void)hipSetDevice(ctx->device);
   if (n == 0 || !state) { ctx->lcg_on = false; return DDCMI_OK; }      /* back to the counter-based stream */
   if (!multID || !prime) SETERR(ctx, DDCMI_EINVAL, "ddcmi_set_random_lcg64: multID and prime are needed with the states");
   if (n != ctx->nloc) SETERR(ctx, DDCMI_EINVAL, "ddcmi_set_random_lcg64: %d records for %d uploaded beads", n, ctx->nloc);
   if (lcg_decomposed(ctx) && ctx->nrebuild > 0)
      SETERR(ctx, DDCMI_EINVAL, "ddcmi_set_random_lcg64: the beads of a decomposed run leave their caller order at the first list build; set the streams after ddcmi_upload_state");
   for (int i = 0; i < n; i++)      /* lcg64_checkValue (lcg64.c:111-120) */
      if (multID[i] > 2 || state[i] == 0 || prime[i] % 2 == 0) SETERR(ctx, DDCMI_EINVAL, "ddcmi_set_random_lcg64: record %d {%llx %u %x} is not a valid LCG64 state", i, (unsigned long long)state[i], multID[i], prime[i]);
   std::vector<int> orig((size_t)n);
   HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
   HIPCHK(ctx, hipMemcpy(orig.data(), ctx->orig.p, (size_t)n * sizeof(int), hipMemcpyDeviceToHost));
   std::vector<ulonglong2> h((size_t)n);
   for (int k = 0; k < n; k++)
   {
      const int i = orig[k];      /* slot k holds the bead of caller index i (identity until the first sort) */
      if (i < 0 || i >= n) SETERR(ctx, DDCMI_EINVAL, "ddcmi_set_random_lcg64: slot %d names caller index %d", k, i);
      h[k].x = state[i]; h[k].y = (unsigned long long)multID[i] | ((unsigned long long)prime[i] << 32);
   }
   ENSURE(ctx, ctx->lcg, (size_t)n + 1); ENSURE(ctx, ctx->lcg2, (size_t)n + 1);
   HIPCHK(ctx, hipMemcpy(ctx->lcg.p, h.data(), (size_t)n * sizeof(ulonglong2), hipMemcpyHostToDevice));
   ctx->lcg_on = true;
   return DDCMI_OK;
}
extern "C" int ddcmi_get_random_lcg64(ddcmi_ctx *ctx, int n, uint64_t *state, uint32_t *multID, uint32_t *prime)
{
   if (!ctx || !state) return DDCMI_EINVAL;
   (void)hipSetDevice(ctx->device);
   if (!ctx->lcg_on) SETERR(ctx, DDCMI_EINVAL, "ddcmi_get_random_lcg64: no LCG64 streams are set");
   if (n != ctx->nloc) SETERR(ctx, DDCMI_EINVAL, "ddcmi_get_random_lcg64: %d records asked, %d beads held", n, ctx->nloc);
   if (n == 0) return DDCMI_OK;
   std::vector<ulonglong2> h((size_t)n);
   std::vector<int> orig((size_t)n);
   HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
   HIPCHK(ctx, hipMemcpy(h.data(), ctx->lcg.p, (size_t)n * sizeof(ulonglong2), hipMemcpyDeviceToHost));
   const bool by_slot = lcg_decomposed(ctx);      /* a decomposed run: the order of ddcmi_download_particles */
   if (!by_slot) HIPCHK(ctx, hipMemcpy(orig.data(), ctx->orig.p, (size_t)n * sizeof(int), hipMemcpyDeviceToHost));
   for (int k = 0; k < n; k++)
   {
      const int i = by_slot ? k : orig[k];
      state[i] = h[k].x;
      if (multID) multID[i] = (uint32_t)h[k].y;
      if (prime) prime[i] = (uint32_t)(h[k].y >> 32);
   }
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
   { int rca = ddcmi_agree_poll(ctx); if (rca) return rca; }
   HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
   return DDCMI_OK;
}

/* ------------------------------------------------------------------------- */
extern "C" int ddcmi_upload_state(ddcmi_ctx *ctx, int nlocal, const double *rx, const double *ry, const double *rz,
                                  const double *vx, const double *vy, const double *vz,
                                  const uint64_t *gid, const int *species, const int *group)
{
   /* nlocal == 0: a domain of a decomposed run that holds no bead yet (vacuum, a droplet elsewhere) */
   if (!ctx || nlocal < 0 || (nlocal > 0 && (!rx || !ry || !rz || !species))) return DDCMI_EINVAL;
   if (ctx->nspecies <= 0) SETERR(ctx, DDCMI_EINVAL, "ddcmi_set_species must be called before ddcmi_upload_state");
   { int rct = nb_tables(ctx); if (rct) return rct; }
   (void)hipSetDevice(ctx->device);
   for (int i = 0; i < nlocal; i++)
      if (species[i] < 0 || species[i] >= ctx->nspecies) SETERR(ctx, DDCMI_EINVAL, "particle %d has species %d outside [0,%d)", i, species[i], ctx->nspecies);
   if (group)
      for (int i = 0; i < nlocal; i++)
         if (group[i] < 0 || group[i] >= ctx->ngroup) SETERR(ctx, DDCMI_EINVAL, "particle %d has group %d outside [0,%d)", i, group[i], ctx->ngroup);
   int n = nlocal;
   ctx->lcg_on = false;      /* the streams belong to the beads of the upload they followed: ddcmi_set_random_lcg64 again */
   ctx->nhalo_hint = 0;      /* (another system: the first rebuild waits for its image count) */
   size_t cap = (size_t)n + n / 4 + 1024;     /* room for image atoms; grown on demand */
   ENSURE(ctx, ctx->pos, cap); ENSURE(ctx, ctx->pos2, cap);
   ENSURE(ctx, ctx->gid, cap); ENSURE(ctx, ctx->gid2, cap);
   dbuf<double> *d3[] = {&ctx->vx, &ctx->vy, &ctx->vz, &ctx->vx2, &ctx->vy2, &ctx->vz2, &ctx->fx, &ctx->fy, &ctx->fz};
   for (auto b : d3) ENSURE(ctx, *b, n + 1);
   dbuf<int> *i1[] = {&ctx->species, &ctx->species2, &ctx->group, &ctx->group2, &ctx->orig, &ctx->orig2, &ctx->slot_of_orig, &ctx->cid, &ctx->crank, &ctx->order, &ctx->nimg, &ctx->img_off};
   for (auto b : i1) ENSURE(ctx, *b, n + 1);
   if (n == 0)
   {
      ctx->nloc = 0; ctx->nhalo = 0; ctx->npad = DDCMI_BLOCK; ctx->self_ele = 0.0;
      ctx->list_valid = false; ctx->forces_valid = false; ctx->f_zero = false;
      return DDCMI_OK;
   }
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
                      ctx->d_ljtype_sp.p, ctx->gid.p, ctx->pos.p, ctx->orig.p, ctx->slot_of_orig.p);
   HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
   ctx->slot_valid = true;
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
   ctx->list_valid = false; ctx->forces_valid = false; ctx->f_zero = false;
   return DDCMI_OK;
}

/* new positions of the same beads, caller order -> device order; each taken at the periodic image nearest to the
 * bead's previous position (the host integrator wraps into the box every step, the device keeps positions
 * continuous between rebuilds so that images and lists stay valid) */
__global__ void k_import_pos(int nloc, int pbc, double L0, double L1, double L2, const int *orig, const double *rx, const double *ry, const double *rz, double4 *pos)
{
   int i = blockIdx.x * blockDim.x + threadIdx.x;
   if (i >= nloc) return;
   const int o = orig[i];
   double4 p = pos[i];
   double x = rx[o], y = ry[o], z = rz[o];
   if (pbc & 1) x += L0 * rint((p.x - x) / L0);
   if (pbc & 2) y += L1 * rint((p.y - y) / L1);
   if (pbc & 4) z += L2 * rint((p.z - z) / L2);
   p.x = x; p.y = y; p.z = z;
   pos[i] = p;
}
__global__ void k_import3(int nloc, const int *orig, const double *a, const double *b, const double *c, double *oa, double *ob, double *oc)
{
   int i = blockIdx.x * blockDim.x + threadIdx.x;
   if (i >= nloc) return;
   const int o = orig[i];
   oa[i] = a[o]; ob[i] = b[o]; oc[i] = c[o];
}
extern "C" int ddcmi_upload_positions(ddcmi_ctx *ctx, const double *rx, const double *ry, const double *rz, const double *vx, const double *vy, const double *vz)
{
   if (ctx) { ctx->shell_skip = false; ctx->images_fresh = false; ctx->pack_fresh = false; }
   if (!ctx || !rx || !ry || !rz) return DDCMI_EINVAL;
   if (ctx->nloc <= 0) SETERR(ctx, DDCMI_EINVAL, "ddcmi_upload_positions needs an uploaded state (ddcmi_upload_state)");
   if (ctx->nranks > 1 || ctx->group_) SETERR(ctx, DDCMI_EUNSUPPORTED, "ddcmi_upload_positions: caller-order arrays do not survive migration between domains");
   (void)hipSetDevice(ctx->device);
   const int n = ctx->nloc, nb = cdiv(n, 256);
   hipStream_t st = ctx->stream;
   /* staging: vx2, vy2, vz2 are free between rebuilds */
   HIPCHK(ctx, hipMemcpyAsync(ctx->vx2.p, rx, n * sizeof(double), hipMemcpyHostToDevice, st));
   HIPCHK(ctx, hipMemcpyAsync(ctx->vy2.p, ry, n * sizeof(double), hipMemcpyHostToDevice, st));
   HIPCHK(ctx, hipMemcpyAsync(ctx->vz2.p, rz, n * sizeof(double), hipMemcpyHostToDevice, st));
   hipLaunchKernelGGL(k_import_pos, dim3(nb), dim3(256), 0, st, n, ctx->pbc, ctx->h[0], ctx->h[4], ctx->h[8], ctx->orig.p, ctx->vx2.p, ctx->vy2.p, ctx->vz2.p, ctx->pos.p);
   if (vx && vy && vz)
   {
      HIPCHK(ctx, hipStreamSynchronize(st));
      HIPCHK(ctx, hipMemcpyAsync(ctx->vx2.p, vx, n * sizeof(double), hipMemcpyHostToDevice, st));
      HIPCHK(ctx, hipMemcpyAsync(ctx->vy2.p, vy, n * sizeof(double), hipMemcpyHostToDevice, st));
      HIPCHK(ctx, hipMemcpyAsync(ctx->vz2.p, vz, n * sizeof(double), hipMemcpyHostToDevice, st));
      hipLaunchKernelGGL(k_import3, dim3(nb), dim3(256), 0, st, n, ctx->orig.p, ctx->vx2.p, ctx->vy2.p, ctx->vz2.p, ctx->vx.p, ctx->vy.p, ctx->vz.p);
   }
   HIPCHK(ctx, hipStreamSynchronize(st));
   ctx->forces_valid = false; ctx->f_zero = false; ctx->halo_fresh = false; ctx->drift_done = false;
   return DDCMI_OK;
}

extern "C" int ddcmi_download_state(ddcmi_ctx *ctx, int mask, double *rx, double *ry, double *rz, double *vx, double *vy, double *vz,
                                    double *fx, double *fy, double *fz)
{
   if (!ctx || ctx->nloc <= 0) return DDCMI_EINVAL;
   (void)hipSetDevice(ctx->device);
   { int rca = ddcmi_agree_poll(ctx); if (rca) return rca; }
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
      int P = ctx->pgrid[a];
      double W = L[a] / P;                       /* brick width of this rank */
      if (periodic && L[a] < 2.0 * rlist)
         SETERR(ctx, DDCMI_EUNSUPPORTED, "box length %g on axis %d is shorter than 2*(rmax+deltaR)=%g: the nearest-image convention the reference relies on breaks down", L[a], a, 2.0 * rlist);
      if (P > 1 && W < rlist)
         SETERR(ctx, DDCMI_EUNSUPPORTED, "domain width %g on axis %d is smaller than rmax+deltaR=%g: halo would reach beyond nearest-neighbour domains", W, a, rlist);
      gp.lo[a] = -0.5 * L[a] + ctx->pcoord[a] * W;
      double cmin = 0.5 * rlist;
      int n = (int)floor(W / cmin);
      if (n < 1) n = 1;
      {
         /* a last tile of the axis that would hold less than half of its cells is folded away when that costs at most 2 % of
          * cell width (4 M-bead water: 101 -> 100 cells on the 4-cell axes: no layer of quarter-filled tiles, 7 % fewer workgroups) */
         const int tdim0[3] = {TCX, TCY, TCZ};
         const int r = n % tdim0[a];
         if (r > 0 && 2 * r <= tdim0[a] && n - r >= tdim0[a] && (double)n / (double)(n - r) <= 1.02) n -= r;
      }
      gp.n[a] = n;
      gp.cinv[a] = (double)n / W;
      const int tdim[3] = {TCX, TCY, TCZ};
      gp.m[a] = (periodic || P > 1) ? tdim[a] : 0;      /* one whole tile of margin: interior tiles hold owned beads only */
      gp.g[a] = n + 2 * gp.m[a];
      gp.T[a] = (gp.g[a] + tdim[a] - 1) / tdim[a];
      ncell *= gp.T[a] * tdim[a];
   }
   if (ncell > 2000000000L) SETERR(ctx, DDCMI_EUNSUPPORTED, "cell grid too large");
   gp.ncell = (int)ncell;
   return DDCMI_OK;
}

/* rebuild phase 1: wrap into the box, cell ids, counting sort of the owned beads */
int ddcmi_bl_sort_owned(ddcmi_ctx *ctx)
{
   int rc = setup_grid(ctx);
   if (rc) return rc;
   GridParams &gp = ctx->gp;
   hipStream_t st = ctx->stream;
   int n = ctx->nloc, nb = cdiv(n, 256), ncell = gp.ncell, ncb = cdiv(ncell, 256);
   dbuf<int> *cb[] = {&ctx->cell_cnt_o, &ctx->cell_start_o, &ctx->cell_cnt_h, &ctx->cell_start_h, &ctx->cell_start, &ctx->cell_cnt};
   for (auto b : cb) ENSURE(ctx, *b, ncell + 2);
   /* the counters of the whole rebuild: cell counts, the capacity flags and totals of k_tile_build -- left zeroed by the last rebuild's
    * tail launch (k_rebuild_tail) unless this is the first rebuild, the grid changed, or a rebuild is being started over */
   {
      const bool clean = ctx->counters_clean && ctx->clean_ncell == ncell && ctx->clean_po == ctx->cell_cnt_o.p && ctx->clean_ph == ctx->cell_cnt_h.p;
      ctx->counters_clean = false;
      if (!clean)
         ddcmi_zero_ints(ctx, st, ZeroJobs().add(ctx->cell_cnt_o.p, ncell + 1).add(ctx->cell_cnt_h.p, ncell).add(ctx->d_flags, 8).add(ctx->d_flags + 12, 1)
                                     .add(ctx->d_flags + 32, 2));
   }
   if (n > 0)
   {
      /* (decomposed runs: the "beads are not numbers" flag also rides in slot 30 of the direction counters, so that the halo count
       * round of this rebuild tells every rank) */
      hipLaunchKernelGGL(k_wrap_cell, dim3(nb), dim3(256), 0, st, gp, n, ctx->pos.p, ctx->cid.p, ctx->crank.p, ctx->cell_cnt_o.p, ctx->d_flags + 12,
                         ((ctx->nranks > 1 || ctx->loopback) && ctx->dir_cnt.cap >= 32) ? ctx->dir_cnt.p + 30 : (int *)nullptr,
                         ctx->sort_renumbers ? ctx->orig.p : (int *)nullptr, ctx->sort_renumbers ? ctx->dir_cnt.p : (int *)nullptr);
      ctx->dir28_clean = ctx->sort_renumbers;
   }
   ctx->sort_renumbers = false;
   if ((rc = ddcmi_scan_exclusive(ctx, ctx->cell_cnt_o.p, ctx->cell_start_o.p, ncell + 1, nullptr))) return rc;     /* [ncell] = nloc */
   if (n > 0)
   {
      hipLaunchKernelGGL(k_scatter_order, dim3(nb), dim3(256), 0, st, n, ctx->cid.p, ctx->crank.p, ctx->cell_start_o.p, ctx->order.p);
      if (ctx->nranks > 1 || ctx->loopback || ctx->group_)      /* migrants arrive in message order: sort by gid, so that a run repeats bit for bit */
         hipLaunchKernelGGL(k_sort_cells_key, dim3(ncb), dim3(256), 0, st, ncell, ctx->cell_start_o.p, ctx->cell_cnt_o.p, ctx->order.p, ctx->gid.p, (const int *)nullptr);
      else
         hipLaunchKernelGGL(k_sort_cells, dim3(ncb), dim3(256), 0, st, ncell, ctx->cell_start_o.p, ctx->cell_cnt_o.p, ctx->order.p);
      /* caller index -> slot: a scattered store per bead, kept up only where something reads it every step */
      if (ctx->lcg_on) ENSURE(ctx, ctx->lcg2, (size_t)n + 1);
      const bool slots = (!ctx->bonded_gid && ctx->inc_nrow > 0) || (!ctx->cons_gid && ctx->ncgroup > 0) || (!ctx->mol_gid && ctx->nmol_multi > 0);
      hipLaunchKernelGGL(k_gather_state, dim3(nb), dim3(256), 0, st, n, ctx->order.p, ctx->pos.p, ctx->vx.p, ctx->vy.p, ctx->vz.p,
                         ctx->species.p, ctx->group.p, ctx->gid.p, ctx->orig.p,
                         ctx->pos2.p, ctx->vx2.p, ctx->vy2.p, ctx->vz2.p, ctx->species2.p, ctx->group2.p, ctx->gid2.p, ctx->orig2.p,
                         slots ? ctx->slot_of_orig.p : (int *)nullptr,
                         gp, (ctx->nranks == 1 && !ctx->loopback && !ctx->group_) ? ctx->nimg.p : (int *)nullptr, 1,
                         ctx->lcg_on ? ctx->lcg.p : (const ulonglong2 *)nullptr, ctx->lcg2.p);
      if (ctx->lcg_on) std::swap(ctx->lcg, ctx->lcg2);
      ctx->slot_valid = slots;
      std::swap(ctx->pos, ctx->pos2); std::swap(ctx->vx, ctx->vx2); std::swap(ctx->vy, ctx->vy2); std::swap(ctx->vz, ctx->vz2);
      std::swap(ctx->species, ctx->species2); std::swap(ctx->group, ctx->group2); std::swap(ctx->gid, ctx->gid2); std::swap(ctx->orig, ctx->orig2);
   }
   return DDCMI_OK;
}

/* make room for nh image/halo beads behind the owned ones */
int ddcmi_bl_reserve_halo(ddcmi_ctx *ctx, int nh)
{
   hipStream_t st = ctx->stream;
   int n = ctx->nloc;
   dbuf<int> *hb[] = {&ctx->hsrc_t, &ctx->hshift_t, &ctx->hcid, &ctx->hrank, &ctx->horder, &ctx->halo_src, &ctx->halo_shift};
   for (auto b : hb) ENSURE(ctx, *b, nh + 1);
   if (ctx->cons_gid && ctx->ncgroup > 0 && (size_t)(n + nh) > ctx->vx.cap)      /* the velocity halo of the constraint solves */
      if (ctx->vx.ensure(n + nh, true, st) || ctx->vy.ensure(n + nh, true, st) || ctx->vz.ensure(n + nh, true, st))
         SETERR(ctx, DDCMI_ENOMEM, "growing velocity arrays for %d halo beads failed", nh);
   if ((size_t)(n + nh) > ctx->pos.cap)
   {
      if (ctx->pos.ensure(n + nh, true, st) || ctx->pos2.ensure(n + nh) || ctx->gid.ensure(n + nh, true, st) || ctx->gid2.ensure(n + nh))
         SETERR(ctx, DDCMI_ENOMEM, "growing particle arrays for %d image atoms failed", nh);
   }
   return DDCMI_OK;
}

/* rebuild phase 2 (single domain): periodic self-images */
static int bl_self_images(ddcmi_ctx *ctx)
{
   ctx->hkey_valid = false;      /* self-images are laid out by a scan over the owned beads: already independent of timing */
   GridParams &gp = ctx->gp;
   hipStream_t st = ctx->stream;
   int n = ctx->nloc, nb = cdiv(n, 256), rc;
   /* (nimg was counted by k_gather_state) */
   if ((rc = ddcmi_scan_exclusive(ctx, ctx->nimg.p, ctx->img_off.p, n, ctx->d_flags + 8))) return rc;
   ctx->nhalo_dev = nullptr;
   int nh;
   if (ctx->nhalo_hint > 0 && !ctx->no_image_hint)
   {
      /* The image count moves by a fraction of a per cent between rebuilds: the kernels that lay the images out are launched for a bound
       * taken from the last rebuild and read the count on the device; the host learns it with the build's other results (ddcmi_bl_finish)
       * instead of waiting for it here -- one host round trip less per rebuild.  A count beyond the bound starts the rebuild over. */
      nh = ctx->nhalo_hint + ctx->nhalo_hint / 32 + 1024;
      if (ctx->debug_image_bound > 0) nh = std::min(nh, ctx->debug_image_bound);      /* (tests, DDCMI_DEBUG_HOOKS=1 only: force the start-over path) */
      ctx->nhalo_dev = ctx->d_flags + 8;
   }
   else
   {
      PostJobs pj;
      pj.add(ctx->d_flags + 8, 1);
      if ((rc = ddcmi_post(ctx, st, pj)) || (rc = ddcmi_post_wait(ctx, st))) return rc;
      nh = ctx->mbox_h[pj.off[0]];
   }
   ctx->nhalo = nh;
   if (nh > 0)
   {
      if ((rc = ddcmi_bl_reserve_halo(ctx, nh))) return rc;
      hipLaunchKernelGGL(k_fill_images, dim3(nb), dim3(256), 0, st, gp, n, ctx->pos.p, ctx->img_off.p, ctx->nimg.p, ctx->hsrc_t.p, ctx->hshift_t.p, ctx->hcid.p, ctx->hrank.p, ctx->cell_cnt_h.p, nh);
   }
   return DDCMI_OK;
}

/* rebuild phase 3: sort the halo descriptors by cell, place the halo beads */
int ddcmi_bl_halo_sort(ddcmi_ctx *ctx)
{
   GridParams &gp = ctx->gp;
   hipStream_t st = ctx->stream;
   int n = ctx->nloc, nh = ctx->nhalo, ncell = gp.ncell, ncb = cdiv(ncell, 256), rc;
   if (nh > 0)
   {
      int nhb = cdiv(nh, 256);
      if ((rc = ddcmi_scan_exclusive(ctx, ctx->cell_cnt_h.p, ctx->cell_start_h.p, ncell, nullptr))) return rc;
      hipLaunchKernelGGL(k_scatter_order, dim3(nhb), dim3(256), 0, st, nh, ctx->hcid.p, ctx->hrank.p, ctx->cell_start_h.p, ctx->horder.p, (const int *)ctx->nhalo_dev);
      if (ctx->hkey_valid)      /* decomposed runs: halo descriptors arrive in the order atomics filled the send lists */
         hipLaunchKernelGGL(k_sort_cells_key, dim3(ncb), dim3(256), 0, st, ncell, ctx->cell_start_h.p, ctx->cell_cnt_h.p, ctx->horder.p, ctx->hkey.p, ctx->hshift_t.p);
      else
         hipLaunchKernelGGL(k_sort_cells, dim3(ncb), dim3(256), 0, st, ncell, ctx->cell_start_h.p, ctx->cell_cnt_h.p, ctx->horder.p);
      /* the sorted descriptors and the beads they place, in one launch (a gather launch and an update launch before) */
      hipLaunchKernelGGL(k_halo_update, dim3(cdiv(nh, HU_PER)), dim3(HU_THREADS), 0, st, n, nh, (const int *)nullptr, (const int *)nullptr, gp.L[0], gp.L[1], gp.L[2], ctx->pos.p, ctx->gid.p, true,
                         ctx->hrecv3.p, ctx->hrecv5.p, (const int *)ctx->nhalo_dev, (unsigned long long *)nullptr, 0,
                         ctx->horder.p, ctx->hsrc_t.p, ctx->hshift_t.p, ctx->halo_src.p, ctx->halo_shift.p);
   }
   else HIPCHK(ctx, hipMemsetAsync(ctx->cell_start_h.p, 0, ncell * sizeof(int), st));
   hipLaunchKernelGGL(k_merge_cells, dim3(ncb), dim3(256), 0, st, ncell, n, ctx->cell_cnt_o.p, ctx->cell_start_o.p, ctx->cell_cnt_h.p, ctx->cell_start_h.p, ctx->cell_start.p, ctx->cell_cnt.p);
   return DDCMI_OK;
}

int ddcmi_ensure_slots(ddcmi_ctx *ctx)
{
   if (ctx->slot_valid || ctx->nloc <= 0) return DDCMI_OK;
   hipLaunchKernelGGL(k_slots_from_orig, dim3(cdiv(ctx->nloc, 256)), dim3(256), 0, ctx->stream, ctx->nloc, ctx->orig.p, ctx->slot_of_orig.p);
   ctx->slot_valid = true;
   return DDCMI_OK;
}

extern "C" int ddcmi_build_list(ddcmi_ctx *ctx)
{
   if (!ctx) return DDCMI_EINVAL;
   if (!ctx->have_box || ctx->nlj <= 0 || ctx->updateRate < 0 || (ctx->nloc <= 0 && ctx->nranks == 1))
      SETERR(ctx, DDCMI_EINVAL, "ddcmi_build_list needs box, nonbonded parameters, neighbor settings and an uploaded state");
   (void)hipSetDevice(ctx->device);
   int rc;
   ctx->pack_fresh = false;      /* (the rebuild's own exchange reuses the send buffer) */
   if ((rc = nb_tables(ctx))) return rc;
   if (ctx->nranks > 1 || ctx->loopback) return ddcmi_mg_rebuild(ctx);
   ctx->phase(-1, nullptr);
   for (int pass = 0;; pass++)
   {
      if ((rc = ddcmi_bl_sort_owned(ctx))) return rc;      /* (idempotent: a second pass sorts sorted beads) */
      ctx->phase(0, "sort_owned launched");
      if ((rc = bl_self_images(ctx))) return rc;
      ctx->phase(1, "self_images (sync)");
      if ((rc = ddcmi_bl_halo_sort(ctx))) return rc;
      rc = ddcmi_bl_finish(ctx);
      if (rc == DDCMI_RETRY_IMAGES && pass == 0) continue;      /* more periodic images than the last rebuild's count allowed for */
      if (rc) return rc;
      break;
   }
   ctx->nhalo_hint = ctx->nhalo;
   ctx->phase(14, "localize");
   return ddcmi_mol_split_finish(ctx);      /* one domain: no molecule is split */
}

/* Tile order and XCD ranges of k_nonbond, on the host from the per-tile cost estimates
 * (ntile ints read back with the rebuild's other flags).  The hardware deals workgroups
 * round-robin over the 8 XCDs; workgroup b takes the (b>>3)-th tile of range b&7, a
 * contiguous run of perm[] with 1/8 of the class's estimated work -- equal COUNTS would
 * leave the XCDs that own the thin edge tiles idle at the end of a launch.  Decomposed runs
 * can order the tiles in two classes, each with its own ranges: tiles whose neighbourhood is
 * all owned beads run while the halo exchange is in flight, the others after it
 * (DDCMI_HALO_OVERLAP=1; off by default: on one GPU through the RCCL loopback the split costs
 * more -- two launch tails, the exchange competing for the CUs -- than the 40 us it hides). */
static int schedule_tiles(ddcmi_ctx *ctx, int wg_per_cu)
{
   /* the tile costs came to the host with the build's flags (ddcmi_bl_finish), in pinned memory; the order
    * and the ranges leave from pinned memory too, so this function costs no host round trip of its own */
   const int ntile = ctx->ntile;
   const int *work = ctx->h_pin[0], *stage = work ? work + ntile : nullptr;
   const size_t cap_items = (size_t)ntile + 16 * 1024 + 64;      /* every tile once + the parts the tails may add: 8 XCD runs x 2 classes, at most 1024 items each */
   int *perm = ctx->pinned(1, cap_items + 64), *sched = perm ? perm + cap_items : nullptr;
   if (!work || !perm) SETERR(ctx, DDCMI_ENOMEM, "pinned staging for the tile schedule");
   for (int k = 0; k < 32; k++) sched[k] = 0;
   const bool two = ctx->halo_overlap && (ctx->nranks > 1 || ctx->loopback || ctx->group_);
   /* tiles without owned beads (the margin tiles, empty space) get no workgroup at all: a workgroup that finds
    * nothing to do still has to be dispatched with its 72 KB of LDS and eight waves, and at 4 M beads 45 % of the
    * grid were such workgroups -- the per-CU timeline showed one of the two slots of a CU empty a quarter of the
    * time. */
   std::vector<int> live;
   live.reserve(ntile);
   int n0 = 0;
   if (two)
   {
      for (int t = 0; t < ntile; t++) if ((work[t] & 0x3fffffff) > 0 && !(work[t] >> 30)) live.push_back(t);
      n0 = (int)live.size();
      for (int t = 0; t < ntile; t++) if ((work[t] & 0x3fffffff) > 0 && (work[t] >> 30)) live.push_back(t);
   }
   else { for (int t = 0; t < ntile; t++) if ((work[t] & 0x3fffffff) > 0) live.push_back(t); n0 = (int)live.size(); }
   const int nlive = (int)live.size();
   auto cost_list = [&](int t) { return (double)(work[t] & 0x3fffffff); };
   auto cost_stage = [&](int t) { return (double)stage[t]; };
   /* greedy list scheduling of a range's items on the S workgroup slots of one XCD: the makespan */
   const int S = std::max(1, wg_per_cu) * 32;
   static const bool no_split = getenv("DDCMI_NO_TAIL_SPLIT") != nullptr;
   std::vector<double> heap((size_t)S);
   auto makespan = [&](const int *tl, int n, int m, int k) -> double
   {
      /* the last m tiles are cut into k parts each */
      std::fill(heap.begin(), heap.end(), 0.0);      /* min-heap of slot finish times */
      auto push_item = [&](double c)
      {
         std::pop_heap(heap.begin(), heap.end(), std::greater<double>());
         heap.back() += c;
         std::push_heap(heap.begin(), heap.end(), std::greater<double>());
      };
      for (int q = 0; q < n; q++)
      {
         const int t = tl[q];
         if (q < n - m || k == 1) push_item(cost_stage(t) + cost_list(t));
         else for (int p = 0; p < k; p++) push_item(cost_stage(t) + 1.08 * cost_list(t) / k + 2000.0);
      }
      double mx = 0.0;
      for (double f : heap) mx = std::max(mx, f);
      return mx;
   };
   int nitems = 0;
   auto split = [&](int lo, int hi, int *out, int *longest)
   {
      /* 8 contiguous runs of equal estimated work, one per XCD (equal COUNTS would leave the XCDs that own the
       * thin edge tiles idle at the end of a launch) */
      double W = 0, run = 0;
      for (int q = lo; q < hi; q++) W += cost_list(live[q]) + cost_stage(live[q]);
      int cut[9];
      for (int x = 0; x < 9; x++) cut[x] = (x == 8) ? hi : lo;
      for (int q = lo; q < hi; q++)
      {
         double nxt = run + cost_list(live[q]) + cost_stage(live[q]);
         for (int x = 1; x < 8; x++)
         {
            double target = W * x / 8.0;
            if (run < target && nxt >= target) cut[x] = q + 1;
         }
         run = nxt;
      }
      for (int x = 1; x < 9; x++) cut[x] = std::max(cut[x], cut[x - 1]);
      *longest = 0;
      for (int x = 0; x < 8; x++)
      {
         int *tl = live.data() + cut[x];
         const int n = cut[x + 1] - cut[x];
         /* few rounds of workgroups per slot: the expensive tiles first, the thin edge tiles fill the end of the launch
          * (at many rounds the raster order wins: neighbouring tiles share their neighbourhoods in L2) */
         static const bool no_lpt = getenv("DDCMI_NO_LPT") != nullptr;
         static const int lpt_rounds = getenv("DDCMI_LPT_ROUNDS") ? atoi(getenv("DDCMI_LPT_ROUNDS")) : 8;
         if (!no_lpt && n < lpt_rounds * S)
            std::stable_sort(tl, tl + n, [&](int ta_, int tb_) { return cost_list(ta_) + cost_stage(ta_) > cost_list(tb_) + cost_stage(tb_); });
         /* the tail: how many of the run's last tiles to cut, and into how many parts, by simulated makespan */
         int best_m = 0, best_k = 1;
         /* (only where the last round weighs: with R rounds of workgroups per slot it is worth at most 1/(2R)).
          * The search simulates a dozen schedules per XCD; tile counts barely move between rebuilds, so its answer
          * is kept while the run's tile count stays within 3 % of the count it was found for and re-derived every
          * 64th rebuild. */
         int *cache = ctx->sched_cache[lo == 0 ? 0 : 1][x];
         const bool cached = cache[0] > 0 && abs(cache[0] - n) <= 2 + n / 32 && (ctx->nrebuild & 63) != 0;
         if (cached) { best_m = std::min(cache[1], n); best_k = cache[2]; }
         else if (!no_split && n > 0 && n < 8 * S)
         {
            double best = makespan(tl, n, 0, 1);
            const int r = n % S;
            const int cand[] = {r, r + S / 2, r + S, r + 2 * S, n};
            const int parts[] = {2, 3, 4, 6, 8};
            for (int m : cand)
            {
               if (m <= 0 || m > n) continue;
               for (int k : parts)
               {
                  if ((size_t)m * k > 64 * 8 * 2) continue;
                  double ms = makespan(tl, n, m, k);
                  if (ms < 0.985 * best) { best = ms; best_m = m; best_k = k; }
               }
            }
         }
         if (!cached) { cache[0] = n; cache[1] = best_m; cache[2] = best_k; }
         if (getenv("DDCMI_DEBUG_SCHED")) fprintf(stderr, "ddcmi sched: xcd %d tiles %d tail %d tiles x %d parts\n", x, n, best_m, best_k);
         if ((size_t)nitems + (size_t)n + (size_t)best_m * (best_k - 1) > cap_items) { best_m = 0; best_k = 1; }      /* (cannot happen: m k <= 1024 per run) */
         out[x] = nitems;
         for (int q = 0; q < n; q++)
         {
            if (q < n - best_m) perm[nitems++] = tl[q];
            else for (int p = 0; p < best_k; p++) perm[nitems++] = tl[q] | (p << 24) | ((best_k - 1) << 27);
         }
         *longest = std::max(*longest, nitems - out[x]);
      }
      out[8] = nitems;
   };
   if (ntile >= (1 << 24)) SETERR(ctx, DDCMI_EUNSUPPORTED, "%d tiles: more than a work item's 24 bits name", ntile);
   split(0, n0, &sched[0], &ctx->sched_longest[0]);
   split(n0, nlive, &sched[16], &ctx->sched_longest[1]);
   ctx->ntile_class[0] = sched[8] - sched[0]; ctx->ntile_class[1] = sched[24] - sched[16];
   ctx->nitems = nitems;
   ENSURE(ctx, ctx->tile_perm, cap_items + 1);
   ENSURE(ctx, ctx->sched, 32);
   ENSURE(ctx, ctx->partials, (size_t)(nitems + 8) * 8);
   {
      /* the tile order and the ranges, the displacement words of the shell-limited walk back to zero, and the NEXT rebuild's counters
       * cleared while nothing reads them (ddcmi_bl_sort_owned, mg_phase1_launch): one launch */
      TailJobs tj;
      tj.fetch(ctx->tile_perm.p, perm, std::max(nitems, 1)).fetch(ctx->sched.p, sched, 32);
      const int ncell = ctx->gp.ncell;
      tj.zero.add(ctx->d_results + R_DISP, 6);      /* three doubles */
      tj.zero.add(ctx->cell_cnt_o.p, ncell + 1).add(ctx->cell_cnt_h.p, ncell).add(ctx->d_flags, 8).add(ctx->d_flags + 12, 1).add(ctx->d_flags + 32, 2);
      tj.zero.add(ctx->d_flags + DDCMI_FLAG_AGREE, 2);
      const bool dirs = ctx->dir_cnt.p != nullptr && ctx->dir_cnt.cap >= 32;
      if (dirs) tj.zero.add(ctx->dir_cnt.p, 32);
      int rcf = ddcmi_rebuild_tail(ctx, ctx->stream, tj);
      if (rcf) return rcf;
      ctx->counters_clean = true; ctx->clean_ncell = ncell; ctx->clean_po = ctx->cell_cnt_o.p; ctx->clean_ph = ctx->cell_cnt_h.p;
      ctx->dircnt_clean = dirs;
   }
   return DDCMI_OK;       /* the pinned buffers are rewritten at the next rebuild, behind its own synchronisation */
}

/* rebuild phase 4: per-tile staging lists + full neighbour list (16-bit ELL per tile) */
static void graph_drop(ddcmi_ctx *ctx);
int ddcmi_bl_finish(ddcmi_ctx *ctx)
{
   GridParams &gp = ctx->gp;
   hipStream_t st = ctx->stream;
   int n = ctx->nloc;
   ctx->phase(10, "-> bl_finish");
   graph_drop(ctx);      /* a recorded step names this list's buffers, tile schedule and grid sizes */
   ctx->f_zero = false;  /* (the beads have new slots, a decomposed rank a new number of them) */
   /* 3. per-tile staging lists + full neighbour list (16-bit ELL per tile) */
   ctx->npad = std::max(1, cdiv(n, DDCMI_BLOCK)) * DDCMI_BLOCK;
   int ntile = gp.T[0] * gp.T[1] * gp.T[2];
   ctx->ntile = ntile;
   double vol = gp.L[0] * gp.L[1] * gp.L[2] / (double)ctx->nranks;
   double dens = (double)std::max(n, 1) / vol;
   if (ctx->stage_cap == 0)
   {
      double per_cell = dens / (gp.cinv[0] * gp.cinv[1] * gp.cinv[2]);
      ctx->stage_cap = (((int)((double)NRC * per_cell * 1.05) + 48) + 63) & ~63;
      if (ctx->stage_cap < 384) ctx->stage_cap = 384;      /* k_nonbond's cell tables alias the 24 B per bead position arrays */
      ctx->maxexcl = 1;
      for (int m = 0; m < ctx->nmoltype; m++) if (ctx->mol_nspecies[m] > 1) ctx->maxexcl = 16;
   }
   if (ctx->arena_cap == 0)
   {
      double expect = 4.0 / 3.0 * M_PI * gp.rlist * gp.rlist * gp.rlist * dens;
      ctx->tmpw = ((int)(expect * 1.25) + 24 + 7) & ~7;
      if (ctx->tmpw > 768) ctx->tmpw = 768;          /* k_tile_transpose keeps a row in the registers of eight lanes: at most 24 quads each */
      ctx->arena_cap = (unsigned long long)((double)n * (expect * 1.45 + 32.0)) + 65536ull;
   }
   ENSURE(ctx, ctx->nbr_cnt, ctx->npad); ENSURE(ctx, ctx->excl_cnt, ctx->npad); ENSURE(ctx, ctx->nbr_cum, ctx->npad);
   ENSURE(ctx, ctx->tile_nstage, ntile + 1); ENSURE(ctx, ctx->tile_width, ntile + 1); ENSURE(ctx, ctx->tile_rows, ntile + 1);
   ENSURE(ctx, ctx->tile_work, 5 * (size_t)ntile + 2);
   if (ctx->tile_base.ensure(ntile + 1)) SETERR(ctx, DDCMI_ENOMEM, "tile table allocation failed");
   double rcut = ctx->rmax, dR = ctx->deltaR;
   /* distance shells of the list order: entries a wave rejects as a whole come last.  Shell 0: r < rcut - dR/4; shells
    * 1..NSHELL-1: equal steps of r^2 from there to the list radius (0.7 A wide at the cut-off for the Martini numbers; the
    * last one, beyond rcut + 0.85 dR, holds what no drift brings inside the cut-off) */
   ShellCuts shc;
   {
      const double r0 = rcut - 0.25 * dR;
      shc.r0sq = (float)(r0 * r0); shc.one = !(dR > 1e-9 * rcut);      /* no skin: one shell */
   }
   unsigned long long *d_arena = (unsigned long long *)(ctx->d_flags + 32);    /* arena entries handed out: the one device-wide counter of the build, on a cache line of its own */
   for (int attempt = 0;; attempt++)
   {
      if (attempt == 8) SETERR(ctx, DDCMI_ENOMEM, "neighbour list capacity could not be settled");
      bool has_mol = false;
      for (int m = 0; m < ctx->nmoltype; m++) has_mol |= ctx->mol_nspecies[m] > 1;
      /* LDS image: 16 B per staged bead (+ 4 B molecule id when pairs can be excluded) + the region cell tables */
      /* LDS image: the ring of accepted words (16 KB), 16 B per staged bead (+ 2 B of molecule id when pairs can be excluded), the region cell tables */
      size_t lds = TB_RING_BYTES + (size_t)ctx->stage_cap * (has_mol ? 18 : 16) + (2 * NRC + 16 + 2 * (TB_THREADS / 64) + 8) * sizeof(int) + 16;
      if ((size_t)ctx->stage_cap * sizeof(unsigned short) > TB_RING_BYTES) lds += (size_t)ctx->stage_cap * sizeof(unsigned short);      /* (bare 16-bit entries: the slot -> cell map outgrows the ring) */
      if (lds > 160 * 1024) SETERR(ctx, DDCMI_EUNSUPPORTED, "a tile neighbourhood of %d beads does not fit the 160 KiB LDS", ctx->stage_cap);
      ENSURE(ctx, ctx->stage_idx, (size_t)ntile * ctx->stage_cap);
      if (ctx->nbr16.ensure(ctx->arena_cap)) SETERR(ctx, DDCMI_ENOMEM, "neighbour arena of %llu entries failed", ctx->arena_cap);
      if (ctx->excl16.ensure((size_t)ctx->maxexcl * ctx->npad)) SETERR(ctx, DDCMI_ENOMEM, "excluded-pair entries");
      if (attempt > 0) ddcmi_zero_ints(ctx, st, ZeroJobs().add(ctx->d_flags, 8).add(d_arena, 2));      /* first attempt: zeroed with the cell counters (ddcmi_bl_sort_owned) */
      ctx->pack_type = (ctx->stage_cap < 4096) ? (ctx->nnb <= 8 ? 2 : ctx->nnb <= 16 ? 1 : 0) : 0;
      TileArgs ta;
      ta.ntile = ntile; ta.stage_stride = ctx->stage_cap; ta.cap = ctx->stage_cap; ta.pack_type = ctx->pack_type; ta.nloc = n; ta.halo_shift = ctx->halo_shift.p;
      ta.cell_start_o = ctx->cell_start_o.p; ta.cell_start = ctx->cell_start.p; ta.cell_cnt = ctx->cell_cnt.p;
      ta.stage_idx = ctx->stage_idx.p; ta.tile_nstage = ctx->tile_nstage.p;
      ta.tile_base = ctx->tile_base.p; ta.tile_width = ctx->tile_width.p; ta.tile_rows = ctx->tile_rows.p; ta.tile_work = ctx->tile_work.p;
      ta.nbr16 = ctx->nbr16.p; ta.arena_cap = ctx->arena_cap; ta.arena_used = d_arena;
      ta.nbr_cnt = ctx->nbr_cnt.p; ta.nbr_cum = ctx->nbr_cum.p;
      if (ctx->tmp32.ensure(((size_t)ctx->npad + (size_t)TB_CHUNK * (ntile + 1)) * ctx->tmpw)) SETERR(ctx, DDCMI_ENOMEM, "scratch list allocation failed");      /* every tile rounded up to whole chunks */
      ta.tmp32 = ctx->tmp32.p; ta.tmpw = ctx->tmpw; ta.shc = shc;
      if (ctx->pack_type && ctx->tile_nib.ensure((size_t)ntile * ctx->stage_cap + 16)) SETERR(ctx, DDCMI_ENOMEM, "nibble table allocation failed");
      ta.tile_nib = ctx->tile_nib.p;
      ctx->phase(17, "bl_finish: buffers");
      auto kbuild = has_mol ? (ctx->pack_type == 2 ? k_tile_build<true, 2> : ctx->pack_type == 1 ? k_tile_build<true, 1> : k_tile_build<true, 0>)
                            : (ctx->pack_type == 2 ? k_tile_build<false, 2> : ctx->pack_type == 1 ? k_tile_build<false, 1> : k_tile_build<false, 0>);
      HIPCHK(ctx, dyn_lds_limit(ctx->device, (const void *)kbuild, (int)lds));
      hipLaunchKernelGGL(kbuild, dim3(ntile), dim3(TB_THREADS), lds, st, gp, ta, ctx->npad, ctx->pos.p, ctx->gid.p, ctx->species.p,
                         ctx->nmoltype, ctx->d_moltype_sp.p, ctx->d_mol_nspecies.p, ctx->d_bpair_off.p, ctx->d_bpairI.p, ctx->d_bpairJ.p, ctx->d_exmask.p,
                         ctx->maxexcl, ctx->excl16.p, ctx->excl_cnt.p, ctx->d_flags);
      /* everything the host decides on (capacity flags, totals, the tiles' cost estimates) is final when k_tile_build
       * ends: it travels behind an event, and the host reads it -- and orders the tiles -- while k_tile_transpose runs */
      ctx->phase(18, "bl_finish: build launch");
      unsigned long long tot[3];
      int *h_work = ctx->pinned(0, 5 * (size_t)ntile + 8);      /* per tile: list cost, staging cost, entries, excluded entries, width */
      if (!h_work) SETERR(ctx, DDCMI_ENOMEM, "pinned staging for the tile costs");
      PostJobs pj;
      pj.add(ctx->d_flags, 64).add(ctx->tile_work.p, 5 * (size_t)ntile);      /* flags + the tiles' costs and totals: one post, read while the transposition runs */
      /* (on a stream of its own behind the build: its trip over the host link no longer stands between the build and the transposition) */
      if (!ctx->stream_post) HIPCHK(ctx, hipStreamCreateWithFlags(&ctx->stream_post, hipStreamNonBlocking));
      if (!ctx->ev_build) HIPCHK(ctx, hipEventCreateWithFlags(&ctx->ev_build, hipEventDisableTiming));
      HIPCHK(ctx, hipEventRecord(ctx->ev_build, st));
      HIPCHK(ctx, hipStreamWaitEvent(ctx->stream_post, ctx->ev_build, 0));
      { int rcp = ddcmi_post(ctx, ctx->stream_post, pj); if (rcp) return rcp; }
      {
         const bool scr16 = ctx->pack_type != 0;
         const size_t lds2 = (size_t)(TR_THREADS / 64) * ctx->tmpw * TR_S * sizeof(unsigned short) + (scr16 ? (size_t)ctx->stage_cap + 16 : 0);
         auto ktr = scr16 ? (ctx->tmpw <= 192 ? k_tile_transpose<3, true> : ctx->tmpw <= 384 ? k_tile_transpose<6, true> : k_tile_transpose<12, true>)
                          : (ctx->tmpw <= 192 ? k_tile_transpose<6, false> : ctx->tmpw <= 384 ? k_tile_transpose<12, false> : k_tile_transpose<24, false>);
         HIPCHK(ctx, dyn_lds_limit(ctx->device, (const void *)ktr, (int)lds2));
         hipLaunchKernelGGL(ktr, dim3(ntile), dim3(TR_THREADS), lds2, st, ta);
      }
      HIPCHK(ctx, hipGetLastError());
      ctx->phase(11, "build+transpose launched");
      { int rcp = ddcmi_post_wait(ctx, ctx->stream_post); if (rcp) return rcp; }
      memcpy(ctx->h_flags, ctx->mbox_h + pj.off[0], 64 * sizeof(int));
      if (ctx->nhalo_dev)
      {
         /* the image count the rebuild was launched without (bl_self_images) */
         const int nh_true = ctx->h_flags[8], bound = ctx->nhalo;
         ctx->nhalo_dev = nullptr;
         if (nh_true > bound)
         {
            HIPCHK(ctx, hipStreamSynchronize(st));
            ctx->nhalo_hint = 0;
            return DDCMI_RETRY_IMAGES;      /* (ddcmi_build_list starts over, waiting for the count this time) */
         }
         ctx->nhalo = nh_true;
      }
      memcpy(h_work, ctx->mbox_h + pj.off[1], 5 * (size_t)ntile * sizeof(int));
      ctx->phase(12, "wait for the build");
      tot[0] = tot[1] = 0;
      int maxw = 0;
      for (int t = 0; t < ntile; t++) { tot[0] += (unsigned)h_work[2 * (size_t)ntile + t]; tot[1] += (unsigned)h_work[3 * (size_t)ntile + t]; maxw = std::max(maxw, h_work[4 * (size_t)ntile + t]); }
      memcpy(&tot[2], ctx->h_flags + 32, sizeof(unsigned long long));
      if (ctx->h_flags[12] > 0)
         SETERR(ctx, DDCMI_EINVAL, "%d beads have non-finite coordinates or lie more than a box length outside the box at loop %lld: the run is unstable (time step, overlapping start, singular bonded term?)", ctx->h_flags[12], (long long)ctx->loop);
      bool again = false;
      if (ctx->h_flags[4] > 0) { ctx->stage_cap = (((int)(ctx->h_flags[4] * 1.05) + 32) + 63) & ~63; again = true; }
      if (ctx->h_flags[0] > 0) { ctx->arena_cap = (unsigned long long)((double)tot[2] * 1.10) + 65536ull; again = true; }
      if (ctx->h_flags[1] > 0) { ctx->maxexcl = ctx->h_flags[1] + 4; again = true; }
      if (ctx->h_flags[5] > 0) { ctx->tmpw = ((int)(ctx->h_flags[5] * 1.1) + 8 + 7) & ~7; again = true; }
      if (ctx->tmpw > 768) SETERR(ctx, DDCMI_EUNSUPPORTED, "neighbour lists of more than 768 entries per bead (list radius %g) are not supported", gp.rlist);
      if (again) HIPCHK(ctx, hipStreamSynchronize(st));      /* the transposition still runs on buffers the next attempt may grow */
      if (!again)
      {
         ctx->list_entries = (int64_t)tot[0]; ctx->excl_entries = (int64_t)tot[1];
         ctx->maxnbr = maxw;
         break;
      }
   }
   {
      /* workgroups of k_nonbond a CU holds: its LDS image of a neighbourhood (launch_forces), at most two by registers */
      const size_t capl = (size_t)ctx->stage_cap + 2;
      const size_t lds_nb = (capl * 16 <= NB_ZOFF ? NB_ZOFF + capl * 8 : capl * 24) + (size_t)ctx->nnb * ctx->nnb * sizeof(double4) + (ctx->pack_type ? 0 : capl) + (ctx->pack_type == 2 ? 0 : capl);
      int rcs = schedule_tiles(ctx, (int)std::min<size_t>(2, std::max<size_t>(1, (160 * 1024) / std::max<size_t>(lds_nb, 1))));
      if (rcs) return rcs;
   }
   if (ctx->updateRate == 0)
   {
      /* neighborRef (neighbor.c:209-246): remember where every owned bead was */
      if (ctx->pos0.ensure((size_t)std::max(n, 1))) SETERR(ctx, DDCMI_ENOMEM, "reference positions");
      HIPCHK(ctx, hipMemcpyAsync(ctx->pos0.p, ctx->pos.p, (size_t)n * sizeof(double4), hipMemcpyDeviceToDevice, st));
   }
   if (ctx->nrebuild == 0 && getenv("DDCMI_DEBUG_SCHED")) fprintf(stderr, "ddcmi build: stage_cap %d tmpw %d maxexcl %d pack_type %d tiles %d\n", ctx->stage_cap, ctx->tmpw, ctx->maxexcl, ctx->pack_type, ctx->ntile);
   ctx->phase(13, "schedule_tiles");
   ctx->list_valid = true;
   ctx->nrebuild++;
   {
      /* the displacement bound of the shell-limited walk starts from this list's positions (NbTileArgs::disp).  D covers the owned beads
       * (and their periodic self-images); the beads a decomposed run receives from its neighbours are measured where they arrive:
       * k_halo_update keeps their largest distance from the rebuild's records (NbTileArgs::hdisp) */
      ctx->sh_r0sq = (double)shc.r0sq; ctx->sh_step = ((double)gp.rlist * gp.rlist - (double)shc.r0sq) / (double)(NSHELL - 1.01);
      ctx->shell_skip = !ctx->no_shell_skip && !shc.one && ctx->sh_step > 0.0;
      /* (D and the received beads' two displacement words were zeroed by the rebuild's tail launch, schedule_tiles) */
   }
   return ddcmi_bonded_localize(ctx);      /* terms given by gid: located among the owned + halo beads */
}

/* ------------------------------------------------------------------------- */
/* defer_reduce: the caller (a time step) folds the nonbonded reduction and the final
 * energies into the launch that reduces the kinetic terms */
static int launch_forces(ddcmi_ctx *ctx, bool defer_reduce = false, FuseArgs *fuse = nullptr /* in: the integrator's pass rides in the pair kernel; out: ->dt = 0 if this launch could not take it */)
{
   hipStream_t st = ctx->stream;
   int n = ctx->nloc, nh = ctx->nhalo;
   /* Decomposed runs, between rebuilds: the halo exchange (pack, one RCCL message per peer,
    * unpack) runs on a second stream while this stream computes the tiles whose
    * neighbourhoods hold owned beads only; the other tiles wait for it. */
   bool halo_pending = false;
   /* the received beads' displacement since the rebuild (NbTileArgs::hdisp): measured by the halo update of a decomposed run whose pair
    * kernel may end its rows early; the word of this step's parity is the one this step's pair kernel reads */
   const bool hdisp_on = ctx->shell_skip && nh > 0 && (ctx->nranks > 1 || ctx->loopback || ctx->group_);
   const int hpar = (int)(ctx->loop & 1);
   unsigned long long *hmax = hdisp_on ? (unsigned long long *)(ctx->d_results + R_DISP + 1) : nullptr;
   if ((ctx->nranks > 1 || ctx->loopback) && !ctx->halo_fresh && !ctx->halo_overlap)
   {
      int rc0 = ddcmi_mg_refresh_halo(ctx, st);
      if (rc0) return rc0;
      if (nh > 0)
         hipLaunchKernelGGL(k_halo_update, dim3(cdiv(nh, HU_PER)), dim3(HU_THREADS), 0, st, n, nh, ctx->halo_src.p, ctx->halo_shift.p,
                            ctx->gp.L[0], ctx->gp.L[1], ctx->gp.L[2], ctx->pos.p, ctx->gid.p, false, ctx->hrecv3.p, ctx->hrecv5.p, (const int *)nullptr, hmax, hpar);
   }
   else if ((ctx->nranks > 1 || ctx->loopback) && !ctx->halo_fresh)
   {
      if (!ctx->stream2)
      {
         HIPCHK(ctx, hipStreamCreateWithFlags(&ctx->stream2, hipStreamNonBlocking));
         HIPCHK(ctx, hipEventCreateWithFlags(&ctx->ev_drift, hipEventDisableTiming));
         HIPCHK(ctx, hipEventCreateWithFlags(&ctx->ev_halo, hipEventDisableTiming));
      }
      HIPCHK(ctx, hipEventRecord(ctx->ev_drift, st));                  /* positions of this step are final */
      HIPCHK(ctx, hipStreamWaitEvent(ctx->stream2, ctx->ev_drift, 0));
      int rc0 = ddcmi_mg_refresh_halo(ctx, ctx->stream2);
      if (rc0) return rc0;
      if (nh > 0)
         hipLaunchKernelGGL(k_halo_update, dim3(cdiv(nh, HU_PER)), dim3(HU_THREADS), 0, ctx->stream2, n, nh, ctx->halo_src.p, ctx->halo_shift.p,
                            ctx->gp.L[0], ctx->gp.L[1], ctx->gp.L[2], ctx->pos.p, ctx->gid.p, false, ctx->hrecv3.p, ctx->hrecv5.p, (const int *)nullptr, hmax, hpar);
      HIPCHK(ctx, hipEventRecord(ctx->ev_halo, ctx->stream2));
      halo_pending = true;
   }
   else if (nh > 0 && !ctx->images_fresh)      /* (the rebuild this step began with made the periodic images from these very positions) */
      hipLaunchKernelGGL(k_halo_update, dim3(cdiv(nh, HU_PER)), dim3(HU_THREADS), 0, st, n, nh, ctx->halo_src.p, ctx->halo_shift.p,
                         ctx->gp.L[0], ctx->gp.L[1], ctx->gp.L[2], ctx->pos.p, ctx->gid.p, false, ctx->hrecv3.p, ctx->hrecv5.p, (const int *)nullptr, hmax, hpar);
   ctx->images_fresh = false;
   const bool has_bonded = (ctx->nbond + ctx->nangle + ctx->ntors + ctx->nrest) > 0;
   const double self = ((ctx->excludePotentialTerm & 128) == 0) ? ctx->self_ele : 0.0;
   if ((ctx->excludePotentialTerm & 128) == 0)
   {
      /* Bonded terms and restraints FIRST, into a zeroed force array: the pair kernel then finishes every bead's force in ONE place --
       * in memory (plain launch: f = f_pair + f_bonded) or in registers in front of the integrator's pass (FUSE), so that systems
       * with bonded terms take the fused step too (the lipid box paid a separate 54 us kick kernel and a force store +
       * re-read).  Both kinds of launch form the same sum, so the fused and the split step stay bit for bit alike.  The fused
       * launch hands the array back zeroed (f_zero): between print steps no launch is spent on clearing it. */
      if (has_bonded && n > 0)
      {
         if (halo_pending) { HIPCHK(ctx, hipStreamWaitEvent(st, ctx->ev_halo, 0)); halo_pending = false; }      /* bonded partners may be halo beads */
         if (!ctx->f_zero) hipLaunchKernelGGL(k_zero3, dim3(cdiv(n, 256)), dim3(256), 0, st, n, ctx->fx.p, ctx->fy.p, ctx->fz.p);
         ctx->f_zero = false;
         int rcb = ddcmi_launch_bonded(ctx);
         if (rcb) return rcb;
      }
      int ntile = ctx->ntile;
      bool useq = ctx->has_charge;
      bool packed = ctx->pack_type != 0;
      const bool shbit = ctx->pack_type == 2;
      const size_t capl = (size_t)ctx->stage_cap + 2;      /* + sentinel slot 0, kept even so every LDS array stays 16-byte aligned */
      /* fixed LDS layout ({x,y} at 0, z at NB_ZOFF) for neighbourhoods of up to NB_ZOFF/16 beads, which is every Martini system; else the run-time layout */
      const bool zfix = capl * 16 <= NB_ZOFF;
      size_t lds = (zfix ? NB_ZOFF + capl * 8 : capl * 24) + (size_t)ctx->nnb * ctx->nnb * sizeof(double4) + (packed ? 0 : capl) + (shbit ? 0 : capl);      /* + shifted-copy flags unless the entries carry them */
      if (lds > 160 * 1024) SETERR(ctx, DDCMI_EUNSUPPORTED, "nonbonded kernel needs %zu bytes of LDS (> 160 KiB)", lds);
      FuseArgs fa;
      memset(&fa, 0, sizeof(fa));
      if (fuse)
      {
         /* the rows of kinetic sums ([waves][8] doubles): in the gap between the {x,y} array and z when there is one, else behind
          * everything -- but never at the price of the second workgroup per CU */
         const size_t rows = (NB_THREADS / 64) * 8 * sizeof(double);
         const bool gap = zfix && capl * 16 + rows <= NB_ZOFF;
         const size_t lds_f = gap ? lds : ((lds + 7) & ~(size_t)7) + rows;
         const bool keeps_two = lds_f * 2 <= 160 * 1024 || lds * 2 > 160 * 1024;
         if (zfix && keeps_two && lds_f <= 160 * 1024)
         {
            fa = *fuse;
            fa.ke_off = gap ? (int)(NB_ZOFF - rows) : (int)((lds + 7) & ~(size_t)7);
            lds = lds_f;
         }
         else { fuse->dt = 0.0; fuse = nullptr; }
      }
      NbTileArgs na;
      na.ntile = ntile; na.stage_stride = ctx->stage_cap; na.cap = (int)capl; na.nlj = ctx->nnb;
      na.cell_start_o = ctx->cell_start_o.p; na.stage_idx = ctx->stage_idx.p; na.tile_nstage = ctx->tile_nstage.p;
      na.cell_start = ctx->cell_start.p; na.cell_cnt = ctx->cell_cnt.p;
      na.tile_base = ctx->tile_base.p; na.tile_width = ctx->tile_width.p; na.tile_rows = ctx->tile_rows.p;
      na.nbr16 = ctx->nbr16.p; na.nbr_cnt = ctx->nbr_cnt.p; na.perm = ctx->tile_perm.p;
      na.tile_work = ctx->tile_work.p; na.halo_shift = ctx->halo_shift.p; na.nloc = n;
      na.disp = ctx->shell_skip ? ctx->d_results + R_DISP : nullptr; na.nbr_cum = ctx->nbr_cum.p; na.sh_r0sq = ctx->sh_r0sq; na.sh_step = ctx->sh_step;
      na.hdisp = hdisp_on ? ctx->d_results + R_DISP + 1 + hpar : nullptr;
      na.addf = (has_bonded && n > 0) ? 1 : 0;
#define LAUNCH_NB(Q, P, S, NT) do { if (zfix) LAUNCH_NBZ(Q, P, S, NT, NB_ZOFF); else LAUNCH_NBZ(Q, P, S, NT, 0); } while (0)
#define LAUNCH_NBZ(Q, P, S, NT, Z) LAUNCH_NBF(Q, P, S, NT, Z, false)
#define LAUNCH_NBF(Q, P, S, NT, Z, F) do { \
         size_t sb_ = 0; \
         if (!lds_starts_at_zero(ctx->device, (const void *)k_nonbond<Q, P, S, NT, NB_WPE, NB_CH, Z, F>, &sb_)) \
            SETERR(ctx, DDCMI_EUNSUPPORTED, "k_nonbond was built with %zu bytes of static LDS: its staged arrays no longer start at LDS address 0 (toolchain change) -- rebuild libddcmi.so with a compiler that gives it none", sb_); \
         HIPCHK(ctx, dyn_lds_limit(ctx->device, (const void *)k_nonbond<Q, P, S, NT, NB_WPE, NB_CH, Z, F>, (int)lds)); \
         hipLaunchKernelGGL((k_nonbond<Q, P, S, NT, NB_WPE, NB_CH, Z, F>), dim3(grid), dim3(NT), lds, st, ctx->gp, na, ctx->npad, ctx->pos.p, ctx->d_kqtab.p, \
                            ctx->excl16.p, ctx->excl_cnt.p, ctx->d_ljtab.p, ctx->rmax * ctx->rmax, ctx->krf, ctx->crf, ctx->keR, \
                            ctx->fx.p, ctx->fy.p, ctx->fz.p, ctx->partials.p, fa); } while (0)
#define LAUNCH_NB2(Q, P, S) LAUNCH_NB(Q, P, S, NB_THREADS)
      /* class 0: tiles with all-owned neighbourhoods (every tile on a single domain);
       * class 1: tiles that stage image/halo beads, after the halo exchange */
      for (int cls = 0; cls < 2; cls++)
      {
         if (cls == 1 && halo_pending) { HIPCHK(ctx, hipStreamWaitEvent(st, ctx->ev_halo, 0)); halo_pending = false; }
         if (ctx->ntile_class[cls] <= 0) continue;
         const double *hd_keep = na.hdisp;
         if (cls == 0 && halo_pending) na.hdisp = nullptr;      /* (tiles that stage owned beads only, while the exchange still writes the word) */
         const int grid = 8 * std::max(ctx->sched_longest[cls], 1);
         na.sched = ctx->sched.p + 16 * cls;
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
            ctx->ev_fused.resize(ctx->ev.size() / 2);
            ctx->ev_fused[ctx->ev_used / 2 - 1] = fuse ? 1 : 0;
            HIPCHK(ctx, hipEventRecord(e0, st));
         }
         if (fuse && useq && shbit) LAUNCH_NBF(true, true, true, NB_THREADS, NB_ZOFF, true);      /* (fuse: fixed LDS layout) */
         else if (fuse && useq && packed) LAUNCH_NBF(true, true, false, NB_THREADS, NB_ZOFF, true);
         else if (fuse && useq) LAUNCH_NBF(true, false, false, NB_THREADS, NB_ZOFF, true);
         else if (fuse && shbit) LAUNCH_NBF(false, true, true, NB_THREADS, NB_ZOFF, true);
         else if (fuse && packed) LAUNCH_NBF(false, true, false, NB_THREADS, NB_ZOFF, true);
         else if (fuse) LAUNCH_NBF(false, false, false, NB_THREADS, NB_ZOFF, true);
         else if (useq && shbit) LAUNCH_NB2(true, true, true);
         else if (useq && packed) LAUNCH_NB2(true, true, false);
         else if (useq) LAUNCH_NB2(true, false, false);
         else if (shbit) LAUNCH_NB2(false, true, true);
         else if (packed) LAUNCH_NB2(false, true, false);
         else LAUNCH_NB2(false, false, false);
         if (ctx->timing) HIPCHK(ctx, hipEventRecord(e1, st));
         na.hdisp = hd_keep;
      }
#undef LAUNCH_NB2
#undef LAUNCH_NB
#undef LAUNCH_NBZ
#undef LAUNCH_NBF
      if (ctx->timing) { ctx->t_launches++; if (fuse) ctx->t_launches_fused++; }          /* per force evaluation: the event pairs of both classes add up */
      if (fuse && na.addf) ctx->f_zero = true;      /* (the fused launch cleared what it consumed) */
      /* the final energies are formed in the same launch (the bonded kernels' sums are complete: they ran first) */
      if (!defer_reduce)
      {
         RedJob j0 = {ctx->partials.p, ctx->nitems, 8, ctx->d_results + R_NB_LJ, 1};
         hipLaunchKernelGGL(k_reduce_jobs, dim3(RED_SPLIT, 1), dim3(1024), 0, st, j0, j0, ctx->d_results, self, ctx->red_tmp.p);
      }
      ctx->forces_valid = true;
      return DDCMI_OK;
   }
   else
   {
      if (!defer_reduce) HIPCHK(ctx, hipMemsetAsync(ctx->d_results, 0, 8 * sizeof(double), st));
      hipLaunchKernelGGL(k_zero3, dim3(cdiv(n, 256)), dim3(256), 0, st, n, ctx->fx.p, ctx->fy.p, ctx->fz.p);
      ctx->f_zero = false;
   }
   if (halo_pending) HIPCHK(ctx, hipStreamWaitEvent(st, ctx->ev_halo, 0));      /* bonded partners may be halo beads */
   int rc = ddcmi_launch_bonded(ctx);
   if (rc) return rc;
   if (!defer_reduce && (has_bonded || (ctx->excludePotentialTerm & 128) != 0))
      hipLaunchKernelGGL(k_finish_energy, dim3(1), dim3(64), 0, st, ctx->d_results, self);
   ctx->forces_valid = true;
   return DDCMI_OK;
}

static int fetch_results(ddcmi_ctx *ctx)
{
   { int rca = ddcmi_agree_poll(ctx); if (rca) return rca; }      /* (a peer whose rebuild failed: say so instead of waiting behind an exchange it never joins) */
   HIPCHK(ctx, hipMemcpyAsync(ctx->h_results, ctx->d_results, R_SIZE * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
   HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
   return DDCMI_OK;
}

extern "C" int ddcmi_eval_forces(ddcmi_ctx *ctx, double *energies, double *virial)
{
   if (!ctx) return DDCMI_EINVAL;
   (void)hipSetDevice(ctx->device);
   int rc;
   if (ctx->group_) SETERR(ctx, DDCMI_EINVAL, "contexts of an in-process group evaluate forces with ddcmi_group_eval_forces");
   if (!ctx->list_valid) { if ((rc = ddcmi_build_list(ctx))) return rc; ctx->images_fresh = true; }
   if ((rc = launch_forces(ctx))) return rc;
   ctx->molv_valid = false;
   if ((rc = fetch_results(ctx))) return rc;
   if (energies) for (int k = 0; k < DDCMI_NE; k++) energies[k] = ctx->h_results[R_E + k];
   if (virial) for (int k = 0; k < 6; k++) virial[k] = ctx->h_results[R_VIR + k];
   return DDCMI_OK;
}

/* kinetic_terms (+ the BACK half kick); with_forces: the same launch also reduces the
 * nonbonded partials of the force evaluation just queued and forms the final energies */
static int launch_kinetic(ddcmi_ctx *ctx, double dt, int do_kick, bool with_forces = false, const GroupLambda *gk = nullptr, bool then_drift = false)
{
   GroupLambda plain;
   if (!gk) { memset(&plain, 0, sizeof(plain)); for (int g = 0; g < 32; g++) { plain.v[g] = 1.0; plain.a[g] = 1.0; } plain.scale[0] = plain.scale[1] = plain.scale[2] = 1.0; gk = &plain; }
   int n = ctx->nloc, nblk = cdiv(n, DDCMI_BLOCK * KE_PER);
   ENSURE(ctx, ctx->kpartials, (size_t)(nblk + 8) * 8);
   if (then_drift)
   {
      if (gk->scale[0] != 1.0 || gk->scale[1] != 1.0 || gk->scale[2] != 1.0) ctx->shell_skip = false;      /* (scaled positions: not a plain drift) */
      hipLaunchKernelGGL(k_kick_ke_drift, dim3(std::max(nblk, 1)), dim3(DDCMI_BLOCK), 0, ctx->stream, n, dt, ctx->d_invmass.p, ctx->d_mass.p, ctx->species.p, ctx->group.p, *gk,
                         ctx->fx.p, ctx->fy.p, ctx->fz.p, ctx->vx.p, ctx->vy.p, ctx->vz.p, ctx->pos.p, ctx->kpartials.p, ctx->gid.p);
   }
   else
   hipLaunchKernelGGL(k_kick_ke, dim3(std::max(nblk, 1)), dim3(DDCMI_BLOCK), 0, ctx->stream, n, dt, ctx->d_invmass.p, ctx->d_mass.p, ctx->species.p,
                      ctx->fx.p, ctx->fy.p, ctx->fz.p, ctx->vx.p, ctx->vy.p, ctx->vz.p, ctx->kpartials.p, do_kick, ctx->group.p, *gk, ctx->gid.p);
   RedJob jk = {ctx->kpartials.p, nblk, 7, ctx->d_results + R_RK, 0, then_drift ? dt : 0.0, ctx->d_results + R_DISP};      /* (+ the drift's share of the displacement bound) */
   if (with_forces)
   {
      const bool nb_on = (ctx->excludePotentialTerm & 128) == 0;
      const double self = nb_on ? ctx->self_ele : 0.0;
      RedJob jf = {ctx->partials.p, nb_on ? ctx->nitems : 0, 8, ctx->d_results + R_NB_LJ, 1};
      hipLaunchKernelGGL(k_reduce_jobs, dim3(RED_SPLIT, 2), dim3(1024), 0, ctx->stream, jf, jk, ctx->d_results, self, ctx->red_tmp.p);
   }
   else
      hipLaunchKernelGGL(k_reduce_jobs, dim3(RED_SPLIT, 1), dim3(1024), 0, ctx->stream, jk, jk, ctx->d_results, 0.0, ctx->red_tmp.p);
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

/* nglf.c:74-95: FRONT half kick + drift, clock advance */
static GroupLambda front_lambda(const ddcmi_ctx *ctx, double dt)
{
   GroupLambda lam;
   lam.lang_mask = 0; lam.seed = ctx->rng_seed; lam.vcm_mask = 0;
   lam.lcg = ctx->lcg_on ? ctx->lcg.p : nullptr;
   lam.scale[0] = lam.scale[1] = lam.scale[2] = 1.0;
   /* the FRONT update of a step sees the loop count before its increment, the BACK update the one after */
   lam.counter_front = 2ull * (unsigned long long)ctx->loop;
   lam.counter_back = 2ull * (unsigned long long)ctx->loop + 1ull;
   const double dt_half = 0.5 * dt;
   for (int g = 0; g < 32; g++)
   {
      /* lambda applies at the FRONT kick when doScaling is set (berendsen.c:74-80) */
      lam.v[g] = (g < ctx->ngroup && ctx->gtype[g] == DDCMI_BERENDSEN && ctx->gdoScaling[g]) ? ctx->glambda[g] : 1.0;
      lam.a[g] = 1.0; lam.dfac[g] = 0.0;
      if (g < ctx->ngroup && ctx->gtype[g] == DDCMI_LANGEVIN)
      {
         lam.lang_mask |= 1u << g;
         lam.a[g] = exp(-dt_half / ctx->gtau[g]);
         lam.dfac[g] = sqrt(2.0 * dt_half * ctx->gTeq[g] / ctx->gtau[g]);          /* kB = 1 in internal units */
         if ((size_t)(3 * g + 2) < ctx->gvcm.size() && (ctx->gvcm[3 * g] != 0.0 || ctx->gvcm[3 * g + 1] != 0.0 || ctx->gvcm[3 * g + 2] != 0.0))
         {
            lam.vcm_mask |= 1u << g;
            for (int k = 0; k < 3; k++) lam.vw[g][k] = (1.0 - lam.a[g]) * ctx->gvcm[3 * g + k];
         }
      }
   }
   return lam;
}
/* The FRONT half of a step in phases, so that a decomposed run can put its exchanges between them (one context: step_pre
 * calls them back to back, the transport's collectives in between; an in-process group: ddcmi_group_step_nglf calls each
 * phase for every domain and moves the data itself):
 *   a  barostat: this rank's sums of the last force evaluation -- virial diagonal and molecular term -- to the host
 *      (+ {P, F} of the split molecules on the device)                       -> all-reduce
 *   b  barostat: pressures, scale factors, box; FRONT half kick (+ drift unless constraints follow)
 *                                                                            -> velocity halo (constraints only)
 *   c  constraints: FRONT solve, drift; clock */
static int mg_allreduce_host_values(ddcmi_ctx *ctx, double *values, int n);
struct PackJob;
static bool ddcmi_mg_pack_job(ddcmi_ctx *ctx, PackJob *pk);
static int mg_allreduce_device(ddcmi_ctx *ctx, double *d, size_t n);
int ddcmi_mg_refresh_vel(ddcmi_ctx *ctx);
static inline bool decomposed(const ddcmi_ctx *ctx) { return ctx->nranks > 1 || ctx->loopback || ctx->group_ != nullptr; }
static int step_pre_a(ddcmi_ctx *ctx)
{
   if (ctx->drift_done || !(ctx->baro_beta > 0.0)) return DDCMI_OK;
   /* nglfconstraint.c:527-536 + changeVolume (:64-84): semi-isotropic Berendsen barostat from the molecular
    * pressure of the last force evaluation, at the TARGET temperature */
   int rcb;
   if (!ctx->molv_valid && (rcb = ddcmi_launch_mol_virial(ctx))) return rcb;     /* first step after ddcmi_eval_forces */
   if ((rcb = fetch_results(ctx))) return rcb;
   const double *mv = ctx->h_results + R_SCR_MOLV;       /* zero unless molecule lists are set */
   ctx->baro_sums[0] = ctx->h_results[R_VIR + DDCMI_XX]; ctx->baro_sums[1] = ctx->h_results[R_VIR + DDCMI_YY]; ctx->baro_sums[2] = ctx->h_results[R_VIR + DDCMI_ZZ];
   ctx->baro_sums[3] = mv[0]; ctx->baro_sums[4] = mv[1]; ctx->baro_sums[5] = mv[2];
   ctx->baro_sums[6] = (double)ctx->nloc;
   return DDCMI_OK;
}
static int step_pre_b(ddcmi_ctx *ctx, double dt)
{
   int n = ctx->nloc, nb = cdiv(n, 256);
   if (ctx->drift_done) return DDCMI_OK;
   GroupLambda lam = front_lambda(ctx, dt);
   if (ctx->baro_beta > 0.0)
   {
      int rcb;
      double split[3];
      if ((rcb = ddcmi_mol_split_term(ctx, split))) return rcb;      /* (P/M) o F of the molecules with atoms on several ranks, from the summed {P, F} */
      const double nmol = ctx->nmol_total > 0 ? (double)ctx->nmol_total : ctx->baro_sums[6];
      const double vol = ctx->h[0] * ctx->h[4] * ctx->h[8], NkT = nmol * ctx->baro_T;
      double pxx = (ctx->baro_sums[0] - (ctx->baro_sums[3] - split[0]) + NkT) / vol - ctx->baro_P0;
      double pyy = (ctx->baro_sums[1] - (ctx->baro_sums[4] - split[1]) + NkT) / vol - ctx->baro_P0;
      double pzz = (ctx->baro_sums[2] - (ctx->baro_sums[5] - split[2]) + NkT) / vol - ctx->baro_P0;
      ctx->pmol[0] = pxx + ctx->baro_P0; ctx->pmol[1] = pyy + ctx->baro_P0; ctx->pmol[2] = pzz + ctx->baro_P0;
      const double btt = ctx->baro_beta * dt / ctx->baro_tau;
      double pl = 0.5 * (pxx + pyy);
      if (ctx->baro_iso) pl = pzz = (1.0 / 3.0) * (pxx + pyy + pzz);          /* molecularPressureGPU.cu:211 */
      double l[3] = {cbrt(1.0 + pl * btt), cbrt(1.0 + pl * btt), cbrt(1.0 + pzz * btt)};
      for (int a = 0; a < 3; a++)
      {
         if (fabs(l[a] - 1.0) < 1e-14) l[a] = 1.0;          /* box.c:44 */
         lam.scale[a] = l[a];
         ctx->h[4 * a] *= l[a];
         ctx->gp.L[a] = ctx->h[4 * a];
      }
   }
   if (ctx->ncgroup > 0 && ctx->nhalo > 0 && (lam.scale[0] != 1.0 || lam.scale[1] != 1.0 || lam.scale[2] != 1.0))
      /* the FRONT solve reads the (scaled) positions of partners that are image / halo beads: adjustPosn for them too (an image
       * r + L goes to lambda r + lambda L, its place in the scaled box); the position halo after the drift replaces them */
      hipLaunchKernelGGL(k_scale_pos, dim3(cdiv(ctx->nhalo, 256)), dim3(256), 0, ctx->stream, ctx->nhalo, lam.scale[0], lam.scale[1], lam.scale[2], ctx->pos.p + n);
   /* the displacement bound of the shell-limited walk follows a plain kick + drift only: the barostat's scaling and the drift behind a
    * constraint solve move beads by more than dt |v| of this kernel */
   if (ctx->ncgroup > 0 || lam.scale[0] != 1.0 || lam.scale[1] != 1.0 || lam.scale[2] != 1.0) ctx->shell_skip = false;
   if (n > 0 && ctx->ncgroup > 0)
      /* nglfconstraint.c:538-553: FRONT kick, velocityConstraintOld(FRONT) at the (scaled) positions, drift */
      hipLaunchKernelGGL(k_kick_drift, dim3(nb), dim3(256), 0, ctx->stream, n, dt, ctx->d_invmass.p, ctx->species.p, ctx->group.p, lam, ctx->gid.p,
                         ctx->fx.p, ctx->fy.p, ctx->fz.p, ctx->vx.p, ctx->vy.p, ctx->vz.p, ctx->pos.p, 1, (double *)nullptr);
   else if (n > 0)
   {
      double *vpart = nullptr;
      if (ctx->shell_skip) { ENSURE(ctx, ctx->kpartials, (size_t)(nb + 8) * 8); vpart = ctx->kpartials.p; }
      hipLaunchKernelGGL(k_kick_drift, dim3(nb), dim3(256), 0, ctx->stream, n, dt, ctx->d_invmass.p, ctx->species.p, ctx->group.p, lam, ctx->gid.p,
                         ctx->fx.p, ctx->fy.p, ctx->fz.p, ctx->vx.p, ctx->vy.p, ctx->vz.p, ctx->pos.p, 3, vpart);
      if (vpart)
      {
         RedJob jd = {vpart, nb, 0, nullptr, 0, dt, ctx->d_results + R_DISP};      /* D += dt max |v| */
         hipLaunchKernelGGL(k_reduce_jobs, dim3(RED_SPLIT, 1), dim3(1024), 0, ctx->stream, jd, jd, ctx->d_results, 0.0, ctx->red_tmp.p);
      }
   }
   return DDCMI_OK;
}
static int step_pre_c(ddcmi_ctx *ctx, double dt)
{
   int n = ctx->nloc, nb = cdiv(n, 256);
   if (!ctx->drift_done && ctx->ncgroup > 0)
   {
      int rcc;
      GroupLambda lam = front_lambda(ctx, dt);
      if ((rcc = ddcmi_launch_constraints(ctx, dt, 0))) return rcc;
      if (n > 0)
         hipLaunchKernelGGL(k_kick_drift, dim3(nb), dim3(256), 0, ctx->stream, n, dt, ctx->d_invmass.p, ctx->species.p, ctx->group.p, lam, ctx->gid.p,
                            ctx->fx.p, ctx->fy.p, ctx->fz.p, ctx->vx.p, ctx->vy.p, ctx->vz.p, ctx->pos.p, 2, (double *)nullptr);
   }
   ctx->drift_done = false;             /* else: the previous step's last kernel already did this kick + drift */
   ctx->time += dt;
   ctx->loop += 1;
   ctx->halo_fresh = false;
   return DDCMI_OK;
}
/* do the constraint solves of this context need its neighbours' velocities? */
static inline bool cons_exchange(const ddcmi_ctx *ctx) { return ctx->ncgroup > 0 && ctx->cons_gid && decomposed(ctx); }
static int step_pre(ddcmi_ctx *ctx, double dt)
{
   int rc;
   if ((rc = step_pre_a(ctx))) return rc;
   if (!ctx->drift_done && ctx->baro_beta > 0.0 && decomposed(ctx))
   {
      if ((rc = mg_allreduce_host_values(ctx, ctx->baro_sums, 7))) return rc;
      if (ctx->nsplit > 0 && (rc = mg_allreduce_device(ctx, ctx->mol_red.p, 6 * (size_t)ctx->nsplit))) return rc;
   }
   const bool had_drift = ctx->drift_done;
   if ((rc = step_pre_b(ctx, dt))) return rc;
   if (!had_drift && cons_exchange(ctx) && (rc = ddcmi_mg_refresh_vel(ctx))) return rc;
   return step_pre_c(ctx, dt);
}
/* nglf.c:97-108: ddcenergy, BACK half kick, kinetic_terms, group Update */
static void graph_drop(ddcmi_ctx *ctx)
{
   if (ctx->graph_exec) (void)hipGraphExecDestroy(ctx->graph_exec);
   ctx->graph_exec = nullptr; ctx->graph_state = 0;
}
/* may the steady-state step be replayed as a graph?  Only when its kernel arguments are the same every
 * step: all groups FREE (no thermostat scalars), no barostat, no constraints, one domain, no event timing */
static bool graph_ok(const ddcmi_ctx *ctx, double dt)
{
   if (ctx->graph_max_beads <= 0 || ctx->nloc > ctx->graph_max_beads || ctx->nloc <= 0) return false;
   if (ctx->group_ || ctx->nranks > 1 || ctx->loopback || ctx->timing || ctx->baro_beta > 0.0 || ctx->ncgroup > 0) return false;
   for (int g = 0; g < ctx->ngroup; g++) if (ctx->gtype[g] != DDCMI_FREE) return false;
   return ctx->graph_state < 2 || ctx->graph_dt == dt;
}
/* may the integrator's pass ride in the pair kernel (k_nonbond<..., FUSE>)?  The force must be complete when the list walk ends
 * and the step must need nothing between the force and the drift */
static bool fuse_ok(const ddcmi_ctx *ctx)
{
   static const bool off = getenv("DDCMI_NO_FUSED_STEP") != nullptr;
   if (off || ctx->nloc <= 0 || ctx->group_) return false;
   if ((ctx->excludePotentialTerm & 128) != 0) return false;
   /* (bonded terms, restraints and charges are no obstacle: their kernels run in front of the pair kernel, the excluded-pair loop ends before the epilogue) */
   if (ctx->ncgroup > 0 || ctx->baro_beta > 0.0) return false;
   for (int g = 0; g < ctx->ngroup; g++) if (ctx->gtype[g] != DDCMI_FREE && ctx->gtype[g] != DDCMI_BERENDSEN) return false;
   return true;
}
static int step_post_cons_b(ddcmi_ctx *ctx, double dt)
{
   int rc;
   if ((rc = ddcmi_launch_constraints(ctx, dt, 1))) return rc;
   GroupLambda lam = front_lambda(ctx, dt);
   return launch_kinetic(ctx, dt, 0, true, &lam, false);
}
static int step_post(ddcmi_ctx *ctx, double dt, bool more_steps)
{
   int rc;
   if (more_steps && graph_ok(ctx, dt))
   {
      ctx->shell_skip = false;      /* (a recorded launch keeps the arguments of the step it was recorded on) */
      ctx->images_fresh = false;    /* (and its kernels: the recording must hold the image update) */
      GroupLambda lam = front_lambda(ctx, dt);
      if (ctx->graph_state == 1)
      {
         /* the previous plain step sized every buffer: record this one */
         hipGraph_t graph = nullptr;
         HIPCHK(ctx, hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeRelaxed));
         rc = launch_forces(ctx, true);
         if (!rc) rc = launch_kinetic(ctx, dt, 1, true, &lam, true);
         hipError_t e = hipStreamEndCapture(ctx->stream, &graph);
         if (rc) { if (graph) (void)hipGraphDestroy(graph); return rc; }
         if (e == hipSuccess && graph && hipGraphInstantiate(&ctx->graph_exec, graph, nullptr, nullptr, 0) == hipSuccess) { ctx->graph_state = 2; ctx->graph_dt = dt; }
         else { (void)hipGetLastError(); ctx->graph_exec = nullptr; ctx->graph_state = 0; ctx->graph_max_beads = 0; }      /* no graphs on this runtime: plain launches from now on */
         if (graph) (void)hipGraphDestroy(graph);
      }
      if (ctx->graph_state == 2)
      {
         HIPCHK(ctx, hipGraphLaunch(ctx->graph_exec, ctx->stream));
         ctx->drift_done = true;
         return DDCMI_OK;
      }
      if (ctx->graph_state == 0 && ctx->graph_max_beads > 0) ctx->graph_state = 1;      /* after the plain step below */
   }
   bool fuse = more_steps && fuse_ok(ctx);
   GroupLambda lam_f;
   if (fuse)
   {
      berendsen_update(ctx, 0.5 * dt);     /* host scalars only: nothing of this step's device results enters */
      lam_f = front_lambda(ctx, dt);
      for (int g = 1; g < ctx->ngroup; g++) fuse &= lam_f.v[g] == lam_f.v[0];      /* one scale factor for every bead */
      if (!fuse)
      {
         /* (groups with different Berendsen factors this step: the split kernels, with the factors just formed) */
         if ((rc = launch_forces(ctx, true))) return rc;
         if ((rc = launch_kinetic(ctx, dt, 1, true, &lam_f, true))) return rc;
         ctx->drift_done = true;
         return DDCMI_OK;
      }
   }
   if (fuse)
   {
      /* forces, BACK kick, kinetic terms, FRONT kick and drift in ONE pass: the pair kernel's epilogue is k_kick_ke_drift */
      const GroupLambda &lam = lam_f;
      FuseArgs fa;
      memset(&fa, 0, sizeof(fa));
      fa.dt = dt;
      fa.lam = ctx->ngroup > 0 ? lam.v[0] : 1.0;
      fa.invmass = ctx->d_invmass.p; fa.massv = ctx->d_mass.p;
      fa.vx = ctx->vx.p; fa.vy = ctx->vy.p; fa.vz = ctx->vz.p;
      ENSURE(ctx, ctx->kpartials, (size_t)(std::max(ctx->nitems, cdiv(ctx->nloc, DDCMI_BLOCK * KE_PER)) + 8) * 8);
      ENSURE(ctx, ctx->pos2, (size_t)ctx->nloc + ctx->nhalo);      /* (the size ddcmi_bl_reserve_halo gave both buffers: no reallocation here) */
      if (ctx->nhalo > 0 && (ctx->nranks > 1 || ctx->loopback) && ctx->fuse_tags_of != ctx->nrebuild)
      {
         /* received halo beads keep their tag word where they lie (k_halo_update rewrites x y z only): both buffers need it
          * (the periodic images of a single domain are whole copies of their owners' records) */
         HIPCHK(ctx, hipMemcpyAsync(ctx->pos2.p + ctx->nloc, ctx->pos.p + ctx->nloc, (size_t)ctx->nhalo * sizeof(double4), hipMemcpyDeviceToDevice, ctx->stream));
         ctx->fuse_tags_of = ctx->nrebuild;
      }
      fa.pos_new = ctx->pos2.p; fa.kpartials = ctx->kpartials.p;
      if ((rc = launch_forces(ctx, true, &fa))) return rc;
      if (fa.dt != 0.0)
      {
         RedJob jf = {ctx->partials.p, ctx->nitems, 8, ctx->d_results + R_NB_LJ, 1};
         RedJob jk = {ctx->kpartials.p, ctx->nitems, 7, ctx->d_results + R_RK, 0, dt, ctx->d_results + R_DISP};      /* + this drift's share of the displacement bound */
         std::swap(ctx->pos, ctx->pos2);
         PackJob pk;
         memset(&pk, 0, sizeof(pk));
         if (ctx->nhalo > 0 && ctx->nranks == 1 && !ctx->loopback)
         {
            /* + the periodic images at the drifted positions, in the same launch: the next force evaluation finds them fresh */
            ImageJob im = {ctx->nloc, ctx->nhalo, ctx->halo_src.p, ctx->halo_shift.p, ctx->gp.L[0], ctx->gp.L[1], ctx->gp.L[2], ctx->pos.p};
            hipLaunchKernelGGL(k_reduce_jobs_images, dim3(2 * RED_SPLIT + cdiv(ctx->nhalo, 1024)), dim3(1024), 0, ctx->stream, jf, jk, ctx->d_results, ctx->self_ele, ctx->red_tmp.p, im, pk);
            ctx->images_fresh = true;
         }
         else if ((ctx->nranks > 1 || ctx->loopback) && !ctx->halo_overlap && ddcmi_mg_pack_job(ctx, &pk) && pk.nsend > 0)
         {
            /* + this rank's halo messages packed from the drifted positions: the next step's exchange starts with the sends */
            ImageJob im;
            memset(&im, 0, sizeof(im));
            hipLaunchKernelGGL(k_reduce_jobs_images, dim3(2 * RED_SPLIT + cdiv(pk.nsend, 1024)), dim3(1024), 0, ctx->stream, jf, jk, ctx->d_results, ctx->self_ele, ctx->red_tmp.p, im, pk);
            ctx->pack_fresh = true;
         }
         else
            hipLaunchKernelGGL(k_reduce_jobs, dim3(RED_SPLIT, 2), dim3(1024), 0, ctx->stream, jf, jk, ctx->d_results, ctx->self_ele, ctx->red_tmp.p);
         ctx->drift_done = true;
         return DDCMI_OK;
      }
      /* (the launch kept the plain kernel: LDS layout; the kick follows as usual) */
      if ((rc = launch_kinetic(ctx, dt, 1, true, &lam, true))) return rc;
      ctx->drift_done = true;
      return DDCMI_OK;
   }
   if ((rc = launch_forces(ctx, true))) return rc;
   berendsen_update(ctx, 0.5 * dt);     /* host scalars only: nothing of this step's device results enters */
   GroupLambda lam = front_lambda(ctx, dt);
   if (ctx->baro_beta > 0.0)      /* from this step's forces, for the next step's barostat */
   {
      if ((rc = ddcmi_launch_mol_virial(ctx))) return rc;
      ctx->molv_valid = true;
   }
   if (ctx->ncgroup > 0)
   {
      /* nglfconstraint.c:567-571: BACK kick, velocityConstraintOld(BACK), then kinetic_terms (a decomposed run puts the
       * velocity halo between the kick and the solve: step_post_cons_b) */
      if (ctx->nloc > 0)
      {
         GroupLambda lb = lam;
         const int nblk = cdiv(ctx->nloc, DDCMI_BLOCK * KE_PER);
         ENSURE(ctx, ctx->kpartials, (size_t)(nblk + 8) * 8);
         hipLaunchKernelGGL(k_kick_ke, dim3(nblk), dim3(DDCMI_BLOCK), 0, ctx->stream, ctx->nloc, dt, ctx->d_invmass.p, ctx->d_mass.p, ctx->species.p,
                            ctx->fx.p, ctx->fy.p, ctx->fz.p, ctx->vx.p, ctx->vy.p, ctx->vz.p, ctx->kpartials.p, 1, ctx->group.p, lb, ctx->gid.p);
      }
      if (ctx->group_) return DDCMI_OK;      /* the group driver exchanges the velocities of all domains, then calls step_post_cons_b */
      if (cons_exchange(ctx) && (rc = ddcmi_mg_refresh_vel(ctx))) return rc;
      return step_post_cons_b(ctx, dt);
   }
   if (more_steps && ctx->nloc > 0 && !(ctx->baro_beta > 0.0))      /* the barostat needs this step's virial before the next drift */
   {
      if ((rc = launch_kinetic(ctx, dt, 1, true, &lam, true))) return rc;
      ctx->drift_done = true;
   }
   else if ((rc = launch_kinetic(ctx, dt, 1, true, &lam, false))) return rc;
   return DDCMI_OK;
}

/* updateRate == 0: does this domain's list need a rebuild?  (neighborCheck; one host
 * round trip per step, as the reference pays an MPI_Allreduce per step in this mode) */
int ddcmi_displacement_check(ddcmi_ctx *ctx, int *need)
{
   hipStream_t st = ctx->stream;
   const int n = ctx->nloc, nblk = cdiv(std::max(n, 1), DDCMI_BLOCK);
   *need = 1;
   if (!ctx->list_valid || ctx->pos0.cap < (size_t)n) return DDCMI_OK;
   *need = 0;
   if (n == 0)
   {
      if (decomposed(ctx) && !ctx->group_) { double z[4] = {0, 0, 0, 0}; return mg_allreduce_host_values(ctx, z, 4); }      /* an empty domain still takes part in the collective */
      return DDCMI_OK;
   }
   ENSURE(ctx, ctx->kpartials, (size_t)(nblk + 8) * 8);
   ENSURE(ctx, ctx->disp, 16);
   HIPCHK(ctx, hipMemsetAsync(ctx->disp.p, 0, 16 * sizeof(double), st));
   hipLaunchKernelGGL(k_disp_sum, dim3(nblk), dim3(DDCMI_BLOCK), 0, st, n, ctx->pos.p, ctx->pos0.p, ctx->kpartials.p);
   RedJob js = {ctx->kpartials.p, nblk, 3, ctx->disp.p, 0};
   hipLaunchKernelGGL(k_reduce_jobs, dim3(RED_SPLIT, 1), dim3(1024), 0, st, js, js, ctx->d_results, 0.0, ctx->red_tmp.p);
   if (decomposed(ctx))
   {
      /* neighborCheck measures every particle of a rank -- halo included -- against ONE mean displacement; here a rank
       * sees only its owned beads, and domains that each subtracted their own centroid would never count the drift of
       * one domain against its neighbour (shear, flow across a face).  With a transport the mean is the global one
       * (all-reduced sums); an in-process group subtracts nothing (conservative: a uniform drift then costs rebuilds,
       * never a stale list). */
      double hs[4] = {0, 0, 0, 0};
      if (!ctx->group_)
      {
         HIPCHK(ctx, hipMemcpyAsync(hs, ctx->disp.p, 3 * sizeof(double), hipMemcpyDeviceToHost, st));
         HIPCHK(ctx, hipStreamSynchronize(st));
         hs[3] = (double)n;
         int rca = mg_allreduce_host_values(ctx, hs, 4);
         if (rca) return rca;
         const double w = hs[3] > 0.0 ? (double)n / hs[3] : 0.0;      /* k_disp_max divides by this rank's bead count */
         for (int k = 0; k < 3; k++) hs[k] *= w;
      }
      HIPCHK(ctx, hipMemcpyAsync(ctx->disp.p, hs, 3 * sizeof(double), hipMemcpyHostToDevice, st));
   }
   hipLaunchKernelGGL(k_disp_max, dim3(nblk), dim3(DDCMI_BLOCK), 0, st, n, ctx->pos.p, ctx->pos0.p, ctx->disp.p, (unsigned long long *)(ctx->disp.p + 4));
   double d2max = 0.0;
   HIPCHK(ctx, hipMemcpyAsync(&d2max, ctx->disp.p + 4, sizeof(double), hipMemcpyDeviceToHost, st));
   HIPCHK(ctx, hipStreamSynchronize(st));
   *need = (2.0 * sqrt(d2max) < ctx->deltaR) ? 0 : 1;
   return DDCMI_OK;
}
static int mg_check_one_domain_features(ddcmi_ctx *ctx);
static int rebuild_due(ddcmi_ctx *ctx, bool *due)
{
   if (!ctx->list_valid) { *due = true; return DDCMI_OK; }
   if (ctx->updateRate > 0) { *due = (ctx->loop % ctx->updateRate == 0); return DDCMI_OK; }
   int need = 0, rc = ddcmi_displacement_check(ctx, &need);
   if (rc) return rc;
   if ((ctx->nranks > 1 || ctx->loopback) && (ctx->comm || ctx->hcomm))
   {
      /* check4updateNeighbor (ddcUpdateAll.c:56): anyone needs a rebuild -> everyone rebuilds */
      double v = (double)need;
      if ((rc = ddcmi_comm_allreduce_sum(ctx, &v, 1))) return rc;
      need = v > 0.0;
   }
   *due = need != 0;
   return DDCMI_OK;
}

extern "C" int ddcmi_step_nglf(ddcmi_ctx *ctx, double dt, int nsteps)
{
   if (!ctx || nsteps < 0) return DDCMI_EINVAL;
   if (!ctx->forces_valid) SETERR(ctx, DDCMI_EINVAL, "ddcmi_step_nglf needs forces: call ddcmi_eval_forces first (firstEnergyCall, masters.c:579)");
   if (ctx->group_) SETERR(ctx, DDCMI_EINVAL, "contexts of an in-process group are stepped with ddcmi_group_step_nglf");
   (void)hipSetDevice(ctx->device);
   int rc;
   if ((rc = mg_check_one_domain_features(ctx))) return rc;
   if (ctx->baro_beta > 0.0 && ctx->nmol_total == 0 && decomposed(ctx)) SETERR(ctx, DDCMI_EINVAL, "the barostat of a decomposed run needs the molecule count: ddcmi_set_molecule_lists_gid");
   if (ctx->baro_beta > 0.0 && ctx->nmol_total == 0)
      for (int m = 0; m < ctx->nmoltype; m++)
         if (ctx->mol_nspecies[m] > 1) SETERR(ctx, DDCMI_EINVAL, "the barostat acts on the molecular pressure: molecule type %d has %d beads, call ddcmi_set_molecule_lists first", m, ctx->mol_nspecies[m]);
   for (int s = 0; s < nsteps; s++)
   {
      if ((rc = step_pre(ctx, dt))) return rc;
      /* ddcUpdateAll.c:64-71: rebuild when loop % updateRate == 0, or (updateRate == 0) when neighborCheck asks */
      bool due = false;
      if ((rc = rebuild_due(ctx, &due))) return rc;
      if (due) { if ((rc = ddcmi_build_list(ctx))) return rc; ctx->images_fresh = true; }      /* (nothing moves between here and this step's forces) */
      if ((rc = step_post(ctx, dt, s + 1 < nsteps))) return rc;
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

/* per-group {kinetic energy, bead count} of this rank's beads -> h_results[R_GROUP..];
 * over RCCL the sums are all-reduced first (energyInfo.c:75-112 allreduce) */
int ddcmi_group_ke_sums(ddcmi_ctx *ctx)
{
   (void)hipSetDevice(ctx->device);
   hipStream_t st = ctx->stream;
   int n = ctx->nloc, ng = ctx->ngroup;
   ENSURE(ctx, ctx->kpartials, (size_t)GKE_BLOCKS * 2 * std::max(ng, 1) + 64);
   hipLaunchKernelGGL(k_group_ke, dim3(GKE_BLOCKS), dim3(DDCMI_BLOCK), 0, st, n, ng, ctx->d_mass.p, ctx->species.p, ctx->group.p,
                      ctx->vx.p, ctx->vy.p, ctx->vz.p, ctx->kpartials.p);
   hipLaunchKernelGGL(k_group_ke_sum, dim3(1), dim3(64), 0, st, ng, ctx->kpartials.p, ctx->d_results + R_GROUP);
   if ((ctx->nranks > 1 || ctx->loopback) && ctx->comm && !ctx->group_ && ng > 0)
      if (ncclAllReduce(ctx->d_results + R_GROUP, ctx->d_results + R_GROUP, 2 * ng, ncclDouble, ncclSum, (ncclComm_t)ctx->comm, st) != ncclSuccess)
         SETERR(ctx, DDCMI_ECOMM, "ncclAllReduce of the group kinetic energies failed");
   int rc = fetch_results(ctx);
   if (rc) return rc;
   if ((ctx->nranks > 1 || ctx->loopback) && ctx->hcomm && !ctx->group_ && ng > 0)      /* host transport: summed on the host copy */
      if (ddcmi_rdzv_allreduce_f64(ctx->hcomm, ctx->h_results + R_GROUP, 2 * ng, 0) != DDCMI_OK)
         SETERR(ctx, DDCMI_ECOMM, "all-reduce of the group kinetic energies failed: %s", ddcmi_rdzv_last_error(ctx->hcomm));
   return DDCMI_OK;
}
extern "C" int ddcmi_group_temperatures(ddcmi_ctx *ctx, double *Tgroup)
{
   if (!ctx || (ctx->nloc <= 0 && ctx->nranks == 1)) return DDCMI_EINVAL;
   if (ctx->group_) SETERR(ctx, DDCMI_EINVAL, "contexts of an in-process group: use ddcmi_group_temperatures_all");
   int rc = ddcmi_group_ke_sums(ctx);
   if (rc) return rc;
   for (int g = 0; g < ctx->ngroup; g++)
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

extern "C" int ddcmi_kinetic_detail(ddcmi_ctx *ctx, int by_species, int nclass, double *out)
{
   if (!ctx || !out || nclass <= 0) return DDCMI_EINVAL;
   if (nclass != (by_species ? ctx->nspecies : ctx->ngroup))
      SETERR(ctx, DDCMI_EINVAL, "ddcmi_kinetic_detail: %d classes asked, the context has %d %s", nclass, by_species ? ctx->nspecies : ctx->ngroup, by_species ? "species" : "groups");
   (void)hipSetDevice(ctx->device);
   hipStream_t st = ctx->stream;
   dbuf<double> part, res;
   if (part.ensure((size_t)GKE_BLOCKS * nclass * 16) || res.ensure((size_t)nclass * KD_NV)) SETERR(ctx, DDCMI_ENOMEM, "kinetic detail scratch");
   hipLaunchKernelGGL(k_class_kinetic, dim3(GKE_BLOCKS), dim3(DDCMI_BLOCK), 0, st, ctx->nloc, nclass, by_species, ctx->d_mass.p, ctx->species.p, ctx->group.p,
                      ctx->vx.p, ctx->vy.p, ctx->vz.p, part.p);
   hipLaunchKernelGGL(k_class_kinetic_sum, dim3(cdiv(nclass * KD_NV, 64)), dim3(64), 0, st, nclass, part.p, res.p);
   HIPCHK(ctx, hipMemcpyAsync(out, res.p, (size_t)nclass * KD_NV * sizeof(double), hipMemcpyDeviceToHost, st));
   HIPCHK(ctx, hipStreamSynchronize(st));
   part.release(); res.release();
   return DDCMI_OK;
}

extern "C" int ddcmi_comm_stats(const ddcmi_ctx *ctx, int64_t stats[8])
{
   if (!ctx || !stats) return DDCMI_EINVAL;
   int v = 0;
   (void)ncclGetVersion(&v);
   stats[0] = ctx->nsend; stats[1] = ctx->nrecv; stats[2] = ctx->hmsg_s.n; stats[3] = ctx->hmsg_r.n;
   stats[4] = v; stats[5] = ctx->comm ? (ctx->loopback ? 3 : 1) : ctx->hcomm ? 2 : 0;
   stats[6] = ctx->nranks; stats[7] = ctx->rank;
   return DDCMI_OK;
}

extern "C" int ddcmi_get_list(ddcmi_ctx *ctx, int which, int *start, int *j, int64_t *nentries)
{
   if (!ctx || !ctx->list_valid || which < 0 || which > 1) return DDCMI_EINVAL;
   (void)hipSetDevice(ctx->device);
   hipStream_t st = ctx->stream;
   int n = ctx->nloc, nb = cdiv(n, 256);
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
      {
         NbTileArgs na;
         na.ntile = ctx->ntile; na.stage_stride = ctx->stage_cap; na.cap = ctx->stage_cap; na.nlj = ctx->nnb;
         na.cell_start_o = ctx->cell_start_o.p; na.stage_idx = ctx->stage_idx.p; na.tile_nstage = ctx->tile_nstage.p;
         na.tile_base = ctx->tile_base.p; na.tile_width = ctx->tile_width.p; na.tile_rows = ctx->tile_rows.p;
         na.nbr16 = ctx->nbr16.p; na.nbr_cnt = ctx->nbr_cnt.p;
         if (which == 0)
            hipLaunchKernelGGL(k_tilelist_to_csr, dim3(ctx->ntile), dim3(256), 0, st, na, ctx->pack_type ? 1 : 0, n, ctx->orig.p, ctx->halo_src.p, d_start.p, d_j.p);
         else
            hipLaunchKernelGGL(k_tileexcl_to_csr, dim3(ctx->ntile), dim3(256), 0, st, na, ctx->pack_type ? 1 : 0, n, ctx->npad, ctx->excl16.p, ctx->excl_cnt.p,
                               ctx->orig.p, ctx->halo_src.p, d_start.p, d_j.p);
      }
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
      if (k / 2 < ctx->ev_fused.size() && ctx->ev_fused[k / 2]) ctx->t_ms_fused += ms;
   }
   ctx->ev_used = 0;
   if (launches) *launches = ctx->t_launches;
   if (total_ms) *total_ms = ctx->t_ms;
   ctx->t_last_fused[0] = (double)ctx->t_launches_fused; ctx->t_last_fused[1] = ctx->t_ms_fused;
   if (reset) { ctx->t_launches = 0; ctx->t_ms = 0; ctx->t_launches_fused = 0; ctx->t_ms_fused = 0; }
   return DDCMI_OK;
}
/* of the launches and milliseconds the last ddcmi_timing_read returned: the share of k_nonbond<..., FUSE> (the pair kernel
 * whose epilogue is the integrator's pass, ddcmi_step_nglf's steps between print steps of systems without bonded terms) */
extern "C" int ddcmi_timing_fused(ddcmi_ctx *ctx, int64_t *launches, double *total_ms)
{
   if (!ctx) return DDCMI_EINVAL;
   if (launches) *launches = (int64_t)ctx->t_last_fused[0];
   if (total_ms) *total_ms = ctx->t_last_fused[1];
   return DDCMI_OK;
}

#include "ddcmi_multigpu.inl"
