/*
 * comm.hip -- multi-GPU layer (RCCL over xGMI).  Replaces the MPI side of ddc
 * (ddcSendRecv.c, ddcUpdate.c, energyInfo.c allreduce).  Round-1 state: the
 * communicator bootstrap and the 24-double energy all-reduce are implemented;
 * the spatial decomposition itself (halo exchange of image slots) is the next
 * step and until then N ranks run replicas.
 */
#include "ddcmi_internal.h"
#include <rccl/rccl.h>

#define NCCLCHK(ctx, call) do { ncclResult_t _r = (call); if (_r != ncclSuccess) SETERR(ctx, DDCMI_ECOMM, "%s failed: %s", #call, ncclGetErrorString(_r)); } while (0)

extern "C" int ddcmi_comm_unique_id(char id[128])
{
   if (!id) return DDCMI_EINVAL;
   ncclUniqueId uid;
   static_assert(sizeof(ncclUniqueId) <= 128, "ncclUniqueId larger than the 128-byte ABI slot");
   if (ncclGetUniqueId(&uid) != ncclSuccess) return DDCMI_ECOMM;
   memset(id, 0, 128);
   memcpy(id, &uid, sizeof(uid));
   return DDCMI_OK;
}

extern "C" int ddcmi_comm_init(ddcmi_ctx *ctx, int rank, int nranks, const char id[128], int px, int py, int pz)
{
   if (!ctx || !id || nranks < 1 || rank < 0 || rank >= nranks) return DDCMI_EINVAL;
   if (px * py * pz != nranks) SETERR(ctx, DDCMI_EINVAL, "process grid %dx%dx%d does not match %d ranks", px, py, pz, nranks);
   (void)hipSetDevice(ctx->device);
   ncclUniqueId uid;
   memcpy(&uid, id, sizeof(uid));
   ncclComm_t comm;
   NCCLCHK(ctx, ncclCommInitRank(&comm, nranks, uid, rank));
   ctx->comm = (void *)comm;
   ctx->rank = rank; ctx->nranks = nranks;
   ctx->pgrid[0] = px; ctx->pgrid[1] = py; ctx->pgrid[2] = pz;
   return DDCMI_OK;
}

extern "C" int ddcmi_comm_allreduce_sum(ddcmi_ctx *ctx, double *values, int n)
{
   if (!ctx || !values || n <= 0 || n > 64) return DDCMI_EINVAL;
   if (ctx->nranks == 1 || !ctx->comm) return DDCMI_OK;
   (void)hipSetDevice(ctx->device);
   double *d = ctx->d_results + R_GROUP;   /* scratch */
   HIPCHK(ctx, hipMemcpyAsync(d, values, n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
   NCCLCHK(ctx, ncclAllReduce(d, d, n, ncclDouble, ncclSum, (ncclComm_t)ctx->comm, ctx->stream));
   HIPCHK(ctx, hipMemcpyAsync(values, d, n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
   HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
   return DDCMI_OK;
}

void ddcmi_comm_destroy(ddcmi_ctx *ctx)
{
   if (ctx->comm) { (void)ncclCommDestroy((ncclComm_t)ctx->comm); ctx->comm = nullptr; }
}
