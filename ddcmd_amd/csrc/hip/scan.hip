/*
 * scan.hip -- hand-written exclusive prefix sum over int arrays (cell counts,
 * image counts).  Replaces the reference's three-stage generateBinPermutation*
 * kernels and cub::DeviceScan (pairProcessGPU.cu:763-1058, prefixScan.cu) --
 * no cub/hipcub/rocPRIM.  Two launches: 2048-element tiles scanned by one
 * 256-thread block each (wave64 shuffles + LDS); then every tile's block sums
 * the totals of the tiles before it (at most a few thousand ints out of L2) and
 * adds that to its elements.  src -> dst (may be the same array); optional
 * grand total written to d_total.  The rebuild is bound by its number of small
 * launches: the earlier copy + three-kernel form cost four of them per scan.
 */
#include "ddcmi_internal.h"
#include <algorithm>

#define SCAN_TILE 2048   /* 256 threads x 8 items */

__device__ __forceinline__ int wave_incl_scan(int v)
{
   int lane = threadIdx.x & 63;
#pragma unroll
   for (int off = 1; off < 64; off <<= 1)
   {
      int t = __shfl_up(v, off, 64);
      if (lane >= off) v += t;
   }
   return v;
}
/* exclusive scan of one value per thread across a 256-thread block; returns block total via *tot */
__device__ __forceinline__ int block_excl_scan(int v, int *tot)
{
   __shared__ int s_w[4];
   int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
   int inc = wave_incl_scan(v);
   if (lane == 63) s_w[w] = inc;
   __syncthreads();
   int base = 0;
   for (int k = 0; k < w; k++) base += s_w[k];
   *tot = s_w[0] + s_w[1] + s_w[2] + s_w[3];
   __syncthreads();
   return base + inc - v;
}
__global__ __launch_bounds__(256) void k_scan_tiles(const int *src, int *dst, int n, int *tile_sums, int *d_total)
{
   int base = blockIdx.x * SCAN_TILE + threadIdx.x * 8;
   int v[8], s = 0;
#pragma unroll
   for (int k = 0; k < 8; k++) { v[k] = (base + k < n) ? src[base + k] : 0; s += v[k]; }
   int tot;
   int ex = block_excl_scan(s, &tot);
#pragma unroll
   for (int k = 0; k < 8; k++) { if (base + k < n) dst[base + k] = ex; ex += v[k]; }
   if (threadIdx.x == 0) { tile_sums[blockIdx.x] = tot; if (d_total && gridDim.x == 1) *d_total = tot; }
}
/* tile b: + the totals of tiles 0..b-1 (fixed summation order: thread-strided partial sums, then the block tree) */
__global__ __launch_bounds__(256) void k_scan_finish(int *data, int n, const int *__restrict__ tile_sums, int *d_total)
{
   const int b = blockIdx.x;
   int p = 0;
   for (int k = threadIdx.x; k < b; k += 256) p += tile_sums[k];
   int tot;
   (void)block_excl_scan(p, &tot);
   if (b > 0)
   {
#pragma unroll
      for (int k = 0; k < 8; k++)
      {
         int i = b * SCAN_TILE + k * 256 + threadIdx.x;
         if (i < n) data[i] += tot;
      }
   }
   if (d_total && b == (int)gridDim.x - 1 && threadIdx.x == 0) *d_total = tot + tile_sums[b];
}

int ddcmi_scan_exclusive(ddcmi_ctx *ctx, const int *src, int *dst, int n, int *d_total)
{
   if (n <= 0)
   {
      if (d_total) HIPCHK(ctx, hipMemsetAsync(d_total, 0, sizeof(int), ctx->stream));
      return DDCMI_OK;
   }
   int nt = cdiv(n, SCAN_TILE);
   ENSURE(ctx, ctx->scan_tmp, nt + 1);
   hipLaunchKernelGGL(k_scan_tiles, dim3(nt), dim3(256), 0, ctx->stream, src, dst, n, ctx->scan_tmp.p, d_total);
   if (nt > 1) hipLaunchKernelGGL(k_scan_finish, dim3(nt), dim3(256), 0, ctx->stream, dst, n, ctx->scan_tmp.p, d_total);
   return DDCMI_OK;
}

/* several small arrays zeroed by ONE launch (a hipMemsetAsync each is a launch each, and an odd byte count two) */
__global__ void k_zero_multi(ZeroJobs z)
{
   const int j = blockIdx.y;
   int *p = z.p[j];
   const int n = z.n[j];
   for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) p[i] = 0;
}
int ddcmi_zero_ints(ddcmi_ctx *ctx, hipStream_t st, const ZeroJobs &z)
{
   int mx = 0;
   for (int j = 0; j < z.cnt; j++) mx = std::max(mx, z.n[j]);
   if (z.cnt <= 0 || mx <= 0) return DDCMI_OK;
   hipLaunchKernelGGL(k_zero_multi, dim3(std::min(cdiv(mx, 256), 1024), z.cnt), dim3(256), 0, st, z);
   return DDCMI_OK;
}
