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
#include <sched.h>
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

/* The mailbox: small results a rebuild decides on (flags, counts, tile costs) are written by a one-workgroup kernel straight
 * into mapped, coherent host memory, followed by a sequence word the host spins on.  hipMemcpyAsync + hipStreamSynchronize
 * cost more than the kernels they wait for: a device-to-host copy of more than a few KB blocks the HOST until the kernel in
 * front of it has finished (the 40 KB of tile costs of a 500 k-bead rank: 283 us inside hipMemcpyAsync, the duration of
 * k_tile_build), and a blocked stream or event wait wakes up 20-90 us late. */
__global__ __launch_bounds__(1024) void k_post(PostJobs j, int *dst, int seq, unsigned *ticket)
{
   /* several workgroups share a large payload (235 KB of tile costs at 4 M beads took one workgroup 35 us, between the search
    * and the transposition); the last one to finish -- a ticket -- writes the sequence word */
   for (int q = 0; q < j.cnt; q++)
   {
      const int *src = j.src[q];
      int *d = dst + j.off[q];
      for (int i = blockIdx.x * 1024 + threadIdx.x; i < j.n[q]; i += gridDim.x * 1024) d[i] = src[i];
   }
   __threadfence_system();
   __syncthreads();
   if (threadIdx.x == 0)
   {
      bool last = true;
      if (gridDim.x > 1)
      {
         __threadfence();
         last = atomicAdd(ticket, 1u) == gridDim.x - 1;
         if (last) { *ticket = 0u; __threadfence_system(); }
      }
      if (last) __hip_atomic_store(dst, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
   }
}
int ddcmi_post(ddcmi_ctx *ctx, hipStream_t st, PostJobs &j)
{
   size_t need = 16;
   for (int q = 0; q < j.cnt; q++) { j.off[q] = (int)need; need += (size_t)((j.n[q] + 3) & ~3); }
   if (need > ctx->mbox_cap)
   {
      /* (every post is waited for before the next one: the old mailbox is idle) */
      if (ctx->mbox_h) (void)hipHostFree(ctx->mbox_h);
      ctx->mbox_h = nullptr; ctx->mbox_d = nullptr; ctx->mbox_cap = 0;
      const size_t cap = need + need / 4 + 1024;
      if (hipHostMalloc((void **)&ctx->mbox_h, cap * sizeof(int), hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess) SETERR(ctx, DDCMI_ENOMEM, "mapped host memory for the mailbox");
      if (hipHostGetDevicePointer((void **)&ctx->mbox_d, ctx->mbox_h, 0) != hipSuccess) SETERR(ctx, DDCMI_ENODEVICE, "device address of the mailbox");
      ctx->mbox_cap = cap;
      ctx->mbox_h[0] = 0; ctx->mbox_seq = 0;
   }
   ctx->mbox_seq++;
   unsigned *ticket = (unsigned *)(ctx->d_flags + DDCMI_FLAG_TICKET);      /* the mailbox's own zeroed word (left at zero by the last workgroup) */
   hipLaunchKernelGGL(k_post, dim3((unsigned)std::min<size_t>(16, (need + 16383) / 16384)), dim3(1024), 0, st, j, ctx->mbox_d, ctx->mbox_seq, ticket);
   HIPCHK(ctx, hipGetLastError());
   return DDCMI_OK;
}
/* wait until the last post has landed; job q's data is at mbox_h + off[q].  The first ~100 us are a hot spin (the single-GPU
 * case the mailbox was built for: the post is a few us away); after that the thread yields between looks -- in a decomposed run
 * the wait covers a collective, i.e. the slowest peer, and this thread may share its cores with the transport's progress thread.
 * A post that never lands is an error after 20 s: with a communicator attached it is reported as such (a peer that is gone),
 * never turned into a blocking hipStreamSynchronize behind a collective that cannot finish. */
int ddcmi_post_wait(ddcmi_ctx *ctx, hipStream_t st)
{
   volatile int *flag = ctx->mbox_h;
   struct timespec t0; clock_gettime(CLOCK_MONOTONIC, &t0);
   bool hot = true;
   for (unsigned long spin = 0;; spin++)
   {
      if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) == ctx->mbox_seq) return DDCMI_OK;
      if (!hot || (spin & 0xff) == 0xff)
      {
         struct timespec t1; clock_gettime(CLOCK_MONOTONIC, &t1);
         const double el = (t1.tv_sec - t0.tv_sec) + 1e-9 * (t1.tv_nsec - t0.tv_nsec);
         if (el > 100e-6) hot = false;
         if (el > 20.0)
         {
            if (ctx->comm || ctx->hcomm)
               SETERR(ctx, DDCMI_ECOMM, "the device posted no results within 20 s (mailbox sequence %d): a collective in front of the post did not finish -- is a peer rank gone?", ctx->mbox_seq);
            /* one rank: the kernels in front of the post did not finish -- let the runtime say why */
            HIPCHK(ctx, hipStreamSynchronize(st));
            if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) == ctx->mbox_seq) return DDCMI_OK;
            SETERR(ctx, DDCMI_ENODEVICE, "the device never posted its results (mailbox sequence %d)", ctx->mbox_seq);
         }
      }
      if (hot) __builtin_ia32_pause(); else sched_yield();
   }
}
/* the other direction: a small table the host prepared in mapped memory, fetched by a kernel (stream-ordered, no host wait) */
__global__ void k_fetch(int *dst, const int *src, int n)
{
   const int i = blockIdx.x * blockDim.x + threadIdx.x;
   if (i < n) dst[i] = src[i];
}
/* the tail of a rebuild in ONE launch (two fetch launches, two memsets and -- decomposed runs -- a host-to-device copy before): the tile
 * order and the XCD ranges come in from mapped host memory, and everything the NEXT rebuild expects to find zero is zeroed now
 * (TailJobs::zero): its counters are dead from here on, and a rebuild that finds them clean starts without a clearing launch */
__global__ void k_rebuild_tail(TailJobs j)
{
   const int stride = gridDim.x * blockDim.x, i0 = blockIdx.x * blockDim.x + threadIdx.x;
   for (int q = 0; q < j.nfetch; q++) for (int i = i0; i < j.fn[q]; i += stride) j.fdst[q][i] = j.fsrc[q][i];
   for (int q = 0; q < j.zero.cnt; q++) { int *p = j.zero.p[q]; for (int i = i0; i < j.zero.n[q]; i += stride) p[i] = 0; }
}
int ddcmi_rebuild_tail(ddcmi_ctx *ctx, hipStream_t st, TailJobs &j)
{
   size_t mx = 1;
   for (int q = 0; q < j.nfetch; q++)
   {
      if (hipHostGetDevicePointer((void **)&j.fsrc[q], (void *)j.fsrc[q], 0) != hipSuccess) SETERR(ctx, DDCMI_ENODEVICE, "device address of a pinned table");
      mx = std::max(mx, (size_t)j.fn[q]);
   }
   for (int q = 0; q < j.zero.cnt; q++) mx = std::max(mx, (size_t)j.zero.n[q]);
   hipLaunchKernelGGL(k_rebuild_tail, dim3((unsigned)std::min<size_t>(256, (mx + 255) / 256)), dim3(256), 0, st, j);
   return DDCMI_OK;
}
int ddcmi_fetch(ddcmi_ctx *ctx, hipStream_t st, int *dst, const int *src_host_mapped, int n)
{
   if (n <= 0) return DDCMI_OK;
   const int *src_d = nullptr;
   if (hipHostGetDevicePointer((void **)&src_d, (void *)src_host_mapped, 0) != hipSuccess) SETERR(ctx, DDCMI_ENODEVICE, "device address of a pinned table");
   hipLaunchKernelGGL(k_fetch, dim3(cdiv(n, 256)), dim3(256), 0, st, dst, src_d, n);
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
