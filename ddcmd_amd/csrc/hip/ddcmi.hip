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
#include <atomic>
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

/* roctx ranges on the regions ddcMD's own profile() calls mark (ptiming.h:10-37: MDSTEP, DDCENERGY, P_FORCE, CHARMM_NONBOND, CHARMM_COVALENT,
 * KINETIC_TERMS, UPDATEALL, PAIRLIST, UPDATE, EVAL_ETYPE), named like them, so that a `rocprofv3 --marker-trace` timeline of a run reads like a
 * ddcMD timing report.  Off unless DDCMI_ROCTX=1: the marker library (librocprofiler-sdk-roctx.so, else libroctx64.so) is looked up with
 * dlopen the first time a range opens -- libddcmi.so does not link it -- and a run without the variable pays one predictable branch per range. */
std::atomic<long> g_roctx_ranges{0};
extern "C" long ddcmi_debug_roctx_ranges(void) { return g_roctx_ranges.load(); }

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
/* Which periodic images does an owned bead have?  By CELL (round 5): the beads of the two outermost layers of cells on a periodic axis -- cells are at
 * least half a list radius wide, so that is everything within the list radius of the face and at most a few per cent more.  An image cell then holds
 * exactly the beads of ONE owned cell, in the same order (k_fill_images places an image by its owner's cell, k_sort_cells orders a cell's images like
 * their owners), so the pair kernel can stage a single domain's images from their owners by cell arithmetic alone (NbTileArgs::self_img). */
#define IMG_LAYERS 2
__device__ __forceinline__ void image_dirs(const GridParams &gp, const double4 &p, int d[3])
{
   double r[3] = {p.x, p.y, p.z};
#pragma unroll
   for (int a = 0; a < 3; a++)
   {
      d[a] = 0;
      if (gp.m[a] == 0) continue;               /* no image margin on this axis */
      const int ic = min(max((int)floor((r[a] - gp.lo[a]) * gp.cinv[a]), 0), gp.n[a] - 1);      /* (the bead's cell: cell_coords) */
      if (ic < IMG_LAYERS) d[a] = +1;                                           /* image at r+L */
      else if (ic >= gp.n[a] - IMG_LAYERS) d[a] = -1;                           /* image at r-L */
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
   int ocx, ocy, ocz;
   cell_coords(gp, p.x, p.y, p.z, true, ocx, ocy, ocz);
   int k = img_off[i];
   for (int sz = 0; sz <= (d[2] != 0); sz++)
      for (int sy = 0; sy <= (d[1] != 0); sy++)
         for (int sx = 0; sx <= (d[0] != 0); sx++)
         {
            if (!(sx | sy | sz)) continue;
            int ix = sx * d[0], iy = sy * d[1], iz = sz * d[2];
            /* the image's cell: its owner's cell moved by whole boxes (from the shifted position a rounding could put it next door, and the image
             * cell would no longer hold its owner cell's beads one for one) */
            int c = cell_linear(gp, ocx + ix * gp.n[0], ocy + iy * gp.n[1], ocz + iz * gp.n[2]);
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

#include "ddcmi_listbuild.inl"
#include "ddcmi_nonbond.inl"
#include "ddcmi_integrator.inl"
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
   ctx->no_lean = getenv("DDCMI_NO_LEAN_STEP") != nullptr;
   ctx->no_self_img = getenv("DDCMI_NO_SELF_IMAGES") != nullptr;
   ctx->no_direct_halo = getenv("DDCMI_NO_DIRECT_HALO") != nullptr;
   ctx->force_lvl = getenv("DDCMI_FORCE_LEVEL_TABLE") != nullptr;
   /* test hook, armed only together with DDCMI_DEBUG_HOOKS=1 (a stray value alone does nothing; read per context: a test sets it between two of them) */
   ctx->debug_image_bound = (getenv("DDCMI_DEBUG_HOOKS") && getenv("DDCMI_DEBUG_IMAGE_BOUND")) ? atoi(getenv("DDCMI_DEBUG_IMAGE_BOUND")) : 0;
   /* (Rounds 3-5 registered the context itself with hipHostRegister, so that the two small count arrays inside it -- mig_scnt, hs_cnt: host
    * transport and in-process groups only -- were DMA targets.  A heap object registered, unregistered and freed, its address handed out again by
    * the allocator: later copies from or to pageable memory in those pages ended with "Memory access fault by GPU node" -- 5 of 12 processes that ran
    * eight workloads one after another died that way, in a different workload every time (tools/rowsloop.sh; 0 of 14 without the registration).
    * The two copies are followed by a stream synchronisation anyway: pageable destinations cost them nothing.) */
   if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess ||
       hipMalloc((void **)&ctx->d_results, R_SIZE * sizeof(double)) != hipSuccess ||
       hipHostMalloc((void **)&ctx->h_results, R_SIZE * sizeof(double), hipHostMallocDefault) != hipSuccess ||
       hipMalloc((void **)&ctx->d_flags, DDCMI_NFLAGS * sizeof(int)) != hipSuccess ||
       hipHostMalloc((void **)&ctx->h_flags, 64 * sizeof(int), hipHostMallocDefault) != hipSuccess)
   {
      g_create_err = "context allocation failed";
      delete ctx;
      return DDCMI_ENOMEM;
   }
   (void)hipMemset(ctx->d_results, 0, R_SIZE * sizeof(double));
   (void)hipMemset(ctx->d_flags, 0, DDCMI_NFLAGS * sizeof(int));
   if (ctx->red_tmp.ensure(2 * RED_SPLIT * 8 + 8)) { g_create_err = "context allocation failed"; delete ctx; return DDCMI_ENOMEM; }
   (void)hipMemset(ctx->red_tmp.p, 0, (2 * RED_SPLIT * 8 + 8) * sizeof(double));      /* incl. the two ticket counters */
   (void)hipDeviceSynchronize();      /* null-stream memsets are not ordered with the context's non-blocking stream */
   memset(ctx->h_results, 0, R_SIZE * sizeof(double));
   { const char *ov = getenv("DDCMI_HALO_OVERLAP"); ctx->halo_overlap = (ov && atoi(ov) != 0); }
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
   if (ctx->stream2) (void)hipStreamSynchronize(ctx->stream2);      /* (an interior search a failed rebuild left behind reads the buffers released below) */
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
   ctx->d_lvltab.release(); ctx->d_lvlidx.release();
   /* (audit of every dbuf member against this function, round 6: these were never released either) */
   ctx->hkey.release(); ctx->lcg.release(); ctx->lcg2.release(); ctx->nbr_cum.release(); ctx->cg_atom_gid.release(); ctx->mol_atom_gid.release(); ctx->cg_slot.release(); ctx->mol_slot.release();
   ctx->mol_mtot.release(); ctx->mol_info.release(); ctx->mol_red.release(); ctx->mol_split.release(); ctx->hs_idx.release(); ctx->dir_cnt.release(); ctx->send_map.release();
   ctx->sendbuf.release(); ctx->hrecv3.release(); ctx->hrecv5.release(); ctx->mig_out.release(); ctx->mig_in.release(); ctx->keep.release(); ctx->cnt_xchg.release();
   ctx->lean_part.release(); ctx->lean_kpart.release(); ctx->lean_hist.release(); ctx->lean_tmp.release(); ctx->lean_bpart.release(); ctx->d_vring.release();      /* (~80 MB per context at 1 M beads) */
   ctx->pos.release(); ctx->pos2.release(); ctx->d_ljtab.release(); ctx->gid.release(); ctx->gid2.release();
   ctx->d_exmask.release(); ctx->rest_gid.release(); ctx->rest_fc.release(); ctx->rest_slot.release(); ctx->rest_r0.release(); ctx->rest_kb.release(); ctx->pos0.release(); ctx->disp.release(); ctx->atom_gid.release(); ctx->hkeys.release();
   for (auto b : {&ctx->cg_dist, &ctx->inc_bpar, &ctx->inc_apar, &ctx->inc_tpar}) b->release();
   for (auto b : {&ctx->cg_atom_off, &ctx->cg_atoms, &ctx->cg_pair_off, &ctx->cons_status, &ctx->mol_off, &ctx->mol_atoms}) b->release();
   ctx->cg_pa.release(); ctx->cg_pb.release();
   for (auto b : {&ctx->inc_boff, &ctx->inc_aoff, &ctx->inc_toff, &ctx->inc_brow, &ctx->inc_arow, &ctx->inc_trow, &ctx->inc_haoff, &ctx->inc_harow, &ctx->inc_hatoms, &ctx->inc_latoms, &ctx->inc_ldesc, &ctx->inc_hdesc, &ctx->inc_tab, &ctx->inc_htab, &ctx->slot_of_atom, &ctx->hvals}) b->release();
   ctx->tile_nib.release();
   ctx->tile_base.release(); ctx->nbr16.release(); ctx->excl16.release(); ctx->kpartials.release(); ctx->red_tmp.release(); ctx->fb.release(); ctx->tmp32.release();
   for (auto &e : ctx->ev) (void)hipEventDestroy(e);
   if (ctx->ev_drift) (void)hipEventDestroy(ctx->ev_drift);
   if (ctx->ev_halo) (void)hipEventDestroy(ctx->ev_halo);
   if (ctx->ev_build) (void)hipEventDestroy(ctx->ev_build);
   if (ctx->ev_sorted) (void)hipEventDestroy(ctx->ev_sorted);
   if (ctx->ev_interior) (void)hipEventDestroy(ctx->ev_interior);
   if (ctx->stream2) { (void)hipStreamSynchronize(ctx->stream2); (void)hipStreamDestroy(ctx->stream2); }
   if (ctx->stream_post) { (void)hipStreamSynchronize(ctx->stream_post); (void)hipStreamDestroy(ctx->stream_post); }
   if (ctx->d_results) (void)hipFree(ctx->d_results);
   if (ctx->h_results) (void)hipHostFree(ctx->h_results);
   if (ctx->d_flags) (void)hipFree(ctx->d_flags);
   if (ctx->h_flags) (void)hipHostFree(ctx->h_flags);
   for (int k = 0; k < 3; k++) if (ctx->h_pin[k]) (void)hipHostFree(ctx->h_pin[k]);
   if (ctx->mbox_h) (void)hipHostFree(ctx->mbox_h);
   if (ctx->agree_h) (void)hipHostFree(ctx->agree_h);
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
   ARGCHK(ctx, !h, "ddcmi_set_box: h is NULL");
   ARGCHK(ctx, pbc < 0 || pbc > 7, "ddcmi_set_box: pbc = %d is not a mask of the three axes (0..7)", pbc);
   const int off[6] = {1, 2, 3, 5, 6, 7};
   for (int k = 0; k < 6; k++)
      if (fabs(h[off[k]]) > 1e-10) SETERR(ctx, DDCMI_EUNSUPPORTED, "only orthorhombic boxes are supported (h[%d]=%g)", off[k], h[off[k]]);
   if (!(h[0] > 0 && h[4] > 0 && h[8] > 0) || !std::isfinite(h[0]) || !std::isfinite(h[4]) || !std::isfinite(h[8])) SETERR(ctx, DDCMI_EINVAL, "box lengths must be positive and finite (%g %g %g)", h[0], h[4], h[8]);
   memcpy(ctx->h, h, sizeof(double) * 9);
   ctx->pbc = pbc;
   ctx->have_box = true;
   ctx->list_valid = false; ctx->forces_valid = false;      /* (the forces on the device are those of the old box) */
   return DDCMI_OK;
}

extern "C" int ddcmi_set_species(ddcmi_ctx *ctx, int nspecies, const double *mass, const double *charge, const int *ljtype, const int *moltype)
{
   ARGCHK(ctx, nspecies <= 0 || !mass || !ljtype, "ddcmi_set_species: %d species%s%s", nspecies, mass ? "" : ", mass is NULL", ljtype ? "" : ", ljtype is NULL");
   for (int s = 0; s < nspecies; s++)
   {
      if (!(mass[s] > 0)) SETERR(ctx, DDCMI_EINVAL, "species %d has non-positive mass", s);      /* (before anything is kept: a refused call changes nothing) */
      if (charge && !std::isfinite(charge[s])) SETERR(ctx, DDCMI_EINVAL, "species %d: charge %g is not finite", s, charge[s]);
   }
   (void)hipSetDevice(ctx->device);
   ctx->nspecies = nspecies;
   ctx->mass.assign(mass, mass + nspecies);
   ctx->charge.assign(nspecies, 0.0);
   if (charge) ctx->charge.assign(charge, charge + nspecies);
   ctx->ljtype.assign(ljtype, ljtype + nspecies);
   ctx->moltype.assign(nspecies, 0);
   if (moltype) ctx->moltype.assign(moltype, moltype + nspecies);
   ctx->has_charge = false;
   ctx->tables_dirty = true; ctx->forces_valid = false;      /* (forces of the old charges: ddcmi_step_nglf asks for a new evaluation, which rebuilds the tables) */
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
   ARGCHK(ctx, nlj <= 0 || nlj > 4096 || !sigma || !eps || !shift, "ddcmi_set_nonbonded: %d LJ types%s", nlj, (sigma && eps && shift) ? "" : ", a table is NULL");
   ARGCHK(ctx, !(rmax > 0) || !std::isfinite(rmax), "ddcmi_set_nonbonded: the cut-off rmax = %g must be positive and finite", rmax);
   ARGCHK(ctx, !std::isfinite(keR) || !std::isfinite(krf) || !std::isfinite(crf), "ddcmi_set_nonbonded: the reaction-field constants keR = %g, krf = %g, crf = %g must be finite", keR, krf, crf);
   for (int k = 0; k < nlj * nlj; k++)
      if (!std::isfinite(sigma[k]) || !std::isfinite(eps[k]) || !std::isfinite(shift[k]))
         SETERR(ctx, DDCMI_EINVAL, "ddcmi_set_nonbonded: entry (%d,%d) of the LJ table is not finite (sigma %g, eps %g, shift %g)", k / nlj, k % nlj, sigma[k], eps[k], shift[k]);
   (void)hipSetDevice(ctx->device);
   ctx->nlj = nlj;
   ctx->sigma.assign(sigma, sigma + nlj * nlj); ctx->eps.assign(eps, eps + nlj * nlj); ctx->shift.assign(shift, shift + nlj * nlj);
   ctx->rmax = rmax; ctx->keR = keR; ctx->krf = krf; ctx->crf = crf;
   ctx->list_valid = false; ctx->forces_valid = false;
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
/* self electrostatic term -1/2 sum q_i^2 keR crf over the local beads (bioMartini.c:1030-1035), one domain: from the species counts of the upload, so that new
 * charges (ddcmi_set_species) or new reaction-field constants (ddcmi_set_nonbonded) under an uploaded state take effect -- until round 6 the term of the
 * upload stayed (tools/fuzz_sequence.py).  Decomposed ranks sum it on the device at every rebuild (ddcmi_multigpu.inl). */
static void update_self_ele(ddcmi_ctx *ctx)
{
   if (ctx->nranks > 1 || ctx->loopback || ctx->group_ || ctx->sp_count.empty()) return;
   double q2 = 0.0;
   for (size_t sp = 0; sp < ctx->sp_count.size() && sp < ctx->charge.size(); sp++) q2 += (double)ctx->sp_count[sp] * ctx->charge[sp] * ctx->charge[sp];
   ctx->self_ele = -0.5 * q2 * ctx->keR * ctx->crf;
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
   {
      /* the distinct entries (ddcmi_ctx::d_lvltab): compared bit for bit, in order of first appearance */
      bool anyq = false;
      for (double q : cls_q) anyq |= q != 0.0;
      std::vector<double4> lv;
      std::vector<unsigned char> idx((size_t)nnb * nnb);
      ctx->nlvl = 0;
      for (size_t k = 0; k < tab.size(); k++)
      {
         double4 e = tab[k];
         if (anyq) e.w = kq[k];      /* (the charged kernel forms 24 eps = 6 x 4 eps itself: its fourth word is ke/eps_r q_a q_b) */
         size_t f = 0;
         for (; f < lv.size(); f++) if (memcmp(&lv[f], &e, sizeof(e)) == 0) break;
         if (f == lv.size()) lv.push_back(e);
         if (lv.size() > 256) break;
         idx[k] = (unsigned char)f;
      }
      if (lv.size() <= 256)
      {
         if ((rc = upload_vec(ctx, ctx->d_lvltab, lv.data(), lv.size())) || (rc = upload_vec(ctx, ctx->d_lvlidx, idx.data(), idx.size()))) return rc;
         ctx->nlvl = (int)lv.size();
      }
   }
   if ((rc = upload_vec(ctx, ctx->d_ljtab, tab.data(), tab.size())) || (rc = upload_vec(ctx, ctx->d_kqtab, kq.data(), kq.size())) ||
       (rc = upload_vec(ctx, ctx->d_ljtype_sp, nb.data(), nb.size()))) return rc;
   ctx->nnb = nnb;
   ctx->tables_dirty = false;
   update_self_ele(ctx);
   if (ctx->nloc > 0 && ctx->pos.p)      /* parameters changed under an uploaded state: refresh the tags */
      hipLaunchKernelGGL(k_retag, dim3(cdiv(ctx->nloc, 256)), dim3(256), 0, ctx->stream, ctx->nloc, ctx->species.p, ctx->d_ljtype_sp.p, ctx->pos.p);
   ctx->list_valid = false;
   return DDCMI_OK;
}

extern "C" int ddcmi_set_molecules(ddcmi_ctx *ctx, int nmoltype, const int *mol_nspecies, const int *bpair_off, const int *bpairI, const int *bpairJ)
{
   ARGCHK(ctx, nmoltype < 0, "ddcmi_set_molecules: %d molecule types", nmoltype);
   ARGCHK(ctx, nmoltype > 0 && (!mol_nspecies || !bpair_off), "ddcmi_set_molecules: mol_nspecies or bpair_off is NULL");
   for (int m = 0; m < nmoltype; m++)
   {
      if (mol_nspecies[m] < 1) SETERR(ctx, DDCMI_EINVAL, "ddcmi_set_molecules: molecule type %d has %d species", m, mol_nspecies[m]);
      if (bpair_off[m] < 0 || bpair_off[m + 1] < bpair_off[m] || (m == 0 && bpair_off[0] != 0))
         SETERR(ctx, DDCMI_EINVAL, "ddcmi_set_molecules: bpair_off must start at 0 and never decrease (type %d: %d .. %d)", m, bpair_off[m], bpair_off[m + 1]);
   }
   ARGCHK(ctx, nmoltype > 0 && bpair_off[nmoltype] > 0 && (!bpairI || !bpairJ), "ddcmi_set_molecules: %d bonded pairs but bpairI or bpairJ is NULL", bpair_off[nmoltype]);
   (void)hipSetDevice(ctx->device);
   ctx->nmoltype = nmoltype;
   ctx->list_valid = false; ctx->forces_valid = false;      /* (... of the old exclusions) */
   if (nmoltype == 0) return DDCMI_OK;
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
   ARGCHK(ctx, !(deltaR >= 0) || !std::isfinite(deltaR), "ddcmi_set_neighbor: the skin deltaR = %g must be >= 0 and finite", deltaR);
   if (updateRate < 0) SETERR(ctx, DDCMI_EINVAL, "updateRate must be >= 0 (0 = rebuild when neighborCheck says so, ddcUpdateAll.c:64-71)");
   ctx->deltaR = deltaR; ctx->updateRate = updateRate; ctx->list_valid = false;
   return DDCMI_OK;
}

extern "C" int ddcmi_set_groups(ddcmi_ctx *ctx, int ngroup, const int *type, const double *Teq, const double *tau, const int *interval)
{
   ARGCHK(ctx, ngroup <= 0 || ngroup > 32 || !type, "ddcmi_set_groups: %d groups (1..32 are supported)%s", ngroup, type ? "" : ", type is NULL");
   for (int g = 0; g < ngroup; g++)
   {
      /* berendsen.c:30-62 forms sqrt(1 + dt/tau (Teq/T - 1)), langevin.c sqrt(2 dt Teq / tau): a negative or NaN tau or Teq is a NaN in every velocity ten steps later */
      if (type[g] == DDCMI_BERENDSEN && ((tau && !(tau[g] >= 0.0 && std::isfinite(tau[g]))) || (Teq && !(Teq[g] >= 0.0 && std::isfinite(Teq[g])))))
         SETERR(ctx, DDCMI_EINVAL, "group %d: BERENDSEN needs Teq >= 0 and tau >= 0 (tau = 0: rescale to Teq every step), both finite", g);
      if (type[g] == DDCMI_LANGEVIN && Teq && !(Teq[g] >= 0.0 && std::isfinite(Teq[g]))) SETERR(ctx, DDCMI_EINVAL, "group %d: LANGEVIN needs a finite Teq >= 0", g);
      if (type[g] == DDCMI_LANGEVIN && tau && !std::isfinite(tau[g])) SETERR(ctx, DDCMI_EINVAL, "group %d: LANGEVIN needs a finite tau > 0", g);
   }
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

extern "C" int ddcmi_set_group_vcm(ddcmi_ctx *ctx, int ngroup, const double *vcm)
{
   ARGCHK(ctx, ngroup < 0 || ngroup > 32 || (ngroup > 0 && !vcm), "ddcmi_set_group_vcm: %d groups%s", ngroup, vcm ? "" : ", vcm is NULL");
   if (ngroup != ctx->ngroup) SETERR(ctx, DDCMI_EINVAL, "ddcmi_set_group_vcm: %d groups, ddcmi_set_groups gave %d", ngroup, ctx->ngroup);
   ctx->gvcm.assign(vcm, vcm + 3 * (size_t)ngroup);
   return DDCMI_OK;
}
extern "C" int ddcmi_set_group_temperature(ddcmi_ctx *ctx, int group, double Teq)
{
   ARGCHK(ctx, group < 0 || group >= ctx->ngroup || !(Teq >= 0.0) || !std::isfinite(Teq), "ddcmi_set_group_temperature: group %d of %d, Teq = %g (must be finite and >= 0)", group, ctx->ngroup, Teq);
   ctx->gTeq[group] = Teq;      /* (host scalars: the next step's factors are formed from them) */
   return DDCMI_OK;
}
extern "C" int ddcmi_set_barostat(ddcmi_ctx *ctx, double T, double P0, double beta, double tau)
{
   ARGCHK(ctx, !(beta >= 0.0) || !std::isfinite(beta) || (beta > 0.0 && (!(tau > 0.0) || !std::isfinite(tau) || !(T >= 0.0) || !std::isfinite(T) || !std::isfinite(P0))),
          "ddcmi_set_barostat: beta = %g must be >= 0 (0 = off) and, when on, tau = %g > 0 with finite T = %g >= 0 and P0 = %g", beta, tau, T, P0);
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
   ARGCHK(ctx, nlocal < 0 || (nlocal > 0 && (!rx || !ry || !rz || !species)), "ddcmi_upload_state: %d beads%s", nlocal, nlocal < 0 ? "" : ", a coordinate array or species is NULL");
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
      ctx->nloc = 0; ctx->nhalo = 0; ctx->npad = DDCMI_BLOCK; ctx->self_ele = 0.0; ctx->sp_count.assign(ctx->nspecies, 0);
      ctx->list_valid = false; ctx->forces_valid = false;
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
   ctx->sp_count.assign(ctx->nspecies, 0);
   for (int i = 0; i < n; i++) ctx->sp_count[species[i]]++;
   {
      double q2 = 0.0;      /* (in bead order, as every round summed it: decomposed ranks keep this value until their first rebuild) */
      for (int i = 0; i < n; i++) { double q = ctx->charge[species[i]]; q2 += q * q; }
      ctx->self_ele = -0.5 * q2 * ctx->keR * ctx->crf;
   }
   ctx->list_valid = false; ctx->forces_valid = false;
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
   ctx->forces_valid = false; ctx->halo_fresh = false; ctx->drift_done = false;
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

#include "ddcmi_rebuild.inl"
#include "ddcmi_step.inl"
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

extern "C" int ddcmi_comm_peer_stats(const ddcmi_ctx *ctx, int cap, int *peer, int64_t *send_beads, int64_t *recv_beads)
{
   if (!ctx || cap < 0) return -1;
   const HaloMsgs &ms = ctx->hmsg_s, &mr = ctx->hmsg_r;
   if (!ctx->comm && !ctx->hcomm) return 0;
   for (int k = 0; k < ms.n && k < cap; k++)
   {
      if (peer) peer[k] = ms.peer[k];
      if (send_beads) send_beads[k] = ms.cnt[k];
      if (recv_beads) { recv_beads[k] = 0; for (int q = 0; q < mr.n; q++) if (mr.peer[q] == ms.peer[k]) recv_beads[k] += mr.cnt[q]; }
   }
   return ms.n;
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
