/*
 * scan.hip -- hand-written exclusive prefix sum over int arrays (cell counts,
 * image counts).  Replaces the reference's three-stage generateBinPermutation*
 * kernels and cub::DeviceScan (pairProcessGPU.cu:763-1058, prefixScan.cu) --
 * no cub/hipcub/rocPRIM.  Two levels: 2048-element tiles scanned by one
 * 256-thread block each (wave64 shuffles + LDS), tile sums scanned by a single
 * block, then added back.  In place; optional grand total written to d_total.
 */
#include "ddcmi_internal.h"

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
__global__ __launch_bounds__(256) void k_scan_tiles(int *data, int n, int *tile_sums)
{
   int base = blockIdx.x * SCAN_TILE + threadIdx.x * 8;
   int v[8], s = 0;
#pragma unroll
   for (int k = 0; k < 8; k++) { v[k] = (base + k < n) ? data[base + k] : 0; s += v[k]; }
   int tot;
   int ex = block_excl_scan(s, &tot);
#pragma unroll
   for (int k = 0; k < 8; k++) { if (base + k < n) data[base + k] = ex; ex += v[k]; }
   if (threadIdx.x == 0) tile_sums[blockIdx.x] = tot;
}
__global__ __launch_bounds__(256) void k_scan_sums(int *sums, int nt, int *d_total)
{
   int carry = 0;
   for (int base = 0; base < nt; base += 256)
   {
      int i = base + threadIdx.x;
      int v = (i < nt) ? sums[i] : 0;
      int tot;
      int ex = block_excl_scan(v, &tot);
      if (i < nt) sums[i] = carry + ex;
      carry += tot;
   }
   if (threadIdx.x == 0 && d_total) *d_total = carry;
}
__global__ void k_scan_add(int *data, int n, const int *tile_sums)
{
   int i = blockIdx.x * blockDim.x + threadIdx.x;
   if (i < n) data[i] += tile_sums[i / SCAN_TILE];
}

int ddcmi_scan_exclusive(ddcmi_ctx *ctx, int *data, int n, int *d_total)
{
   if (n <= 0)
   {
      if (d_total) HIPCHK(ctx, hipMemsetAsync(d_total, 0, sizeof(int), ctx->stream));
      return DDCMI_OK;
   }
   int nt = cdiv(n, SCAN_TILE);
   ENSURE(ctx, ctx->scan_tmp, nt + 1);
   hipLaunchKernelGGL(k_scan_tiles, dim3(nt), dim3(256), 0, ctx->stream, data, n, ctx->scan_tmp.p);
   hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(256), 0, ctx->stream, ctx->scan_tmp.p, nt, d_total);
   hipLaunchKernelGGL(k_scan_add, dim3(cdiv(n, 256)), dim3(256), 0, ctx->stream, data, n, ctx->scan_tmp.p);
   return DDCMI_OK;
}
