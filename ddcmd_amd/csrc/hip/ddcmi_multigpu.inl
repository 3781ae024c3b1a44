/*
 * ddcmi_multigpu.inl -- spatial domain decomposition (compiled into ddcmi.hip).
 *
 * Replaces ddcMD's DDC layer for this path: ddcAssignment (particle ownership and
 * migration, ddcAssignment.c:64-150), ddcSendRecvTables / ddcFindPairs (halo
 * tables, ddcSendRecv.c:41-264), ddcUpdate (per-step position halo,
 * ddcUpdate.c:40-85).  ddcUpdateForce (ddcUpdate.c:140-223) has no counterpart:
 * every rank evaluates the full list of its own beads, so no force returns.
 *
 * Domains are the bricks of a px*py*pz grid (the Voronoi cells of a cubic lattice
 * of domain centres, domain.c:191-208).  Each of the 26 neighbour directions has a
 * destination rank and a periodic shift; a direction whose destination is the
 * rank itself (undivided periodic axis) yields local image beads, exactly like
 * the single-domain path.  Halo beads live in the image slots behind the owned
 * beads; the sender applies the shift.  Transport is RCCL point-to-point
 * (grouped ncclSend/ncclRecv, one message per direction: 7 distinct peers at
 * 2x2x2 = the 7 xGMI links) or, for tests on one GPU, direct copies between
 * contexts of one process (ddcmi_group_*).
 */

struct DirTab
{
   int dest[27];        /* destination rank, -1 = none (open boundary) */
   int shift[27][3];    /* periodic shift the RECEIVER sees, in box lengths */
   int me;
};

extern "C" int ddcmi_plan_directions(int px, int py, int pz, int rank, int pbc, int *dest, int *shift)
{
   if (px < 1 || py < 1 || pz < 1 || rank < 0 || rank >= px * py * pz || !dest || !shift) return DDCMI_EINVAL;
   int P[3] = {px, py, pz};
   int pc[3] = {rank % px, (rank / px) % py, rank / (px * py)};
   for (int code = 0; code < 27; code++)
   {
      int d[3] = {code % 3 - 1, (code / 3) % 3 - 1, code / 9 - 1};
      int c[3], ok = 1;
      for (int a = 0; a < 3; a++)
      {
         c[a] = pc[a] + d[a];
         shift[3 * code + a] = 0;
         if (c[a] < 0)
         {
            if ((pbc >> a) & 1) { c[a] += P[a]; shift[3 * code + a] = +1; } else ok = 0;
         }
         else if (c[a] >= P[a])
         {
            if ((pbc >> a) & 1) { c[a] -= P[a]; shift[3 * code + a] = -1; } else ok = 0;
         }
      }
      dest[code] = (ok && code != 13) ? (c[2] * py + c[1]) * px + c[0] : -1;
   }
   return DDCMI_OK;
}

static void mg_set_topology(ddcmi_ctx *ctx, int rank, int nranks, int px, int py, int pz)
{
   ctx->rank = rank; ctx->nranks = nranks;
   ctx->pgrid[0] = px; ctx->pgrid[1] = py; ctx->pgrid[2] = pz;
   ctx->pcoord[0] = rank % px; ctx->pcoord[1] = (rank / px) % py; ctx->pcoord[2] = rank / (px * py);
   ddcmi_plan_directions(px, py, pz, rank, ctx->pbc, ctx->dir_dest, &ctx->dir_shift[0][0]);
   ctx->list_valid = false;
}

static DirTab mg_dirtab(const ddcmi_ctx *ctx)
{
   DirTab t;
   for (int c = 0; c < 27; c++) { t.dest[c] = ctx->dir_dest[c]; for (int a = 0; a < 3; a++) t.shift[c][a] = ctx->dir_shift[c][a]; }
   t.me = ctx->rank;
   return t;
}

/* ------------------------------------------------------------------------- */
__global__ void k_iota(int n, int *a)
{
   int i = blockIdx.x * blockDim.x + threadIdx.x;
   if (i < n) a[i] = i;
}

struct MigGeom { double L[3], W[3]; int P[3], pc[3], pbc; };

/* ownership (voronoiCalcParticleDestinations for a cubic lattice of centres =
 * brick index) + packing of the beads that leave: record = x y z tag vx vy vz gid {group, LCG64 multID, prime} {LCG64 state} */
__global__ void k_mig_classify(MigGeom mg, int nloc, int mig_cap, double4 *pos, const double *vx, const double *vy, const double *vz,
                               const uint64_t *gid, const int *group, int *keep, int *dir_cnt, double *mig_out, int *flags, const ulonglong2 *lcg, int *orig)
{
   int i = blockIdx.x * blockDim.x + threadIdx.x;
   if (i == 0) dir_cnt[27] = mig_cap;      /* travels with the counts: every rank sees whether any rank's segments overflowed */
   if (i >= nloc) return;
   orig[i] = i;      /* (the beads' numbers of this rebuild: was a launch of its own) */
   double4 p = pos[i];
   if (mg.pbc & 1) { if (p.x > 0.5 * mg.L[0]) p.x -= mg.L[0]; if (p.x < -0.5 * mg.L[0]) p.x += mg.L[0]; }
   if (mg.pbc & 2) { if (p.y > 0.5 * mg.L[1]) p.y -= mg.L[1]; if (p.y < -0.5 * mg.L[1]) p.y += mg.L[1]; }
   if (mg.pbc & 4) { if (p.z > 0.5 * mg.L[2]) p.z -= mg.L[2]; if (p.z < -0.5 * mg.L[2]) p.z += mg.L[2]; }
   pos[i] = p;
   double r[3] = {p.x, p.y, p.z};
   int d[3];
#pragma unroll
   for (int a = 0; a < 3; a++)
   {
      int b = (int)floor((r[a] + 0.5 * mg.L[a]) / mg.W[a]);
      b = min(max(b, 0), mg.P[a] - 1);
      int dd = b - mg.pc[a];
      if (dd > 1) dd -= mg.P[a];
      if (dd < -1) dd += mg.P[a];
      if (dd > 1 || dd < -1) { atomicMax(&flags[6], 1); atomicMax(&dir_cnt[29], 1); dd = 0; }      /* [29] travels with the counts: every rank learns of it in the same round */
      d[a] = dd;
   }
   int code = (d[0] + 1) + 3 * (d[1] + 1) + 9 * (d[2] + 1);
   keep[i] = (code == 13);
   if (code != 13)
   {
      int slot = atomicAdd(&dir_cnt[code], 1);
      if (slot < mig_cap)
      {
         double *rec = mig_out + ((size_t)code * mig_cap + slot) * 10;
         rec[0] = p.x; rec[1] = p.y; rec[2] = p.z; rec[3] = p.w;
         rec[4] = vx[i]; rec[5] = vy[i]; rec[6] = vz[i];
         rec[7] = __longlong_as_double((long long)gid[i]);
         /* [8] group | multID << 8 | prime << 32, [9] the LCG64 state (bit patterns; both zero beyond the group without streams) */
         const ulonglong2 q = lcg ? lcg[i] : make_ulonglong2(0ull, 0ull);
         rec[8] = __longlong_as_double((long long)((unsigned long long)(group[i] & 0xff) | (q.y & 3ull) << 8 | (q.y >> 32) << 32));
         rec[9] = __longlong_as_double((long long)q.x);
      }
   }
}
__global__ void k_compact_order(int n, const int *keep, const int *scan, int *order)
{
   int i = blockIdx.x * blockDim.x + threadIdx.x;
   if (i < n && keep[i]) order[scan[i]] = i;
}
__global__ void k_unpack_mig(int narr, int nkeep, const double *mig_in, double4 *pos, double *vx, double *vy, double *vz,
                             uint64_t *gid, int *species, int *group, int *orig, ulonglong2 *lcg)
{
   int k = blockIdx.x * blockDim.x + threadIdx.x;
   if (k >= narr) return;
   const double *rec = mig_in + (size_t)k * 10;
   int i = nkeep + k;
   pos[i] = make_double4(rec[0], rec[1], rec[2], rec[3]);
   vx[i] = rec[4]; vy[i] = rec[5]; vz[i] = rec[6];
   gid[i] = (uint64_t)__double_as_longlong(rec[7]);
   species[i] = (int)((__double_as_longlong(rec[3]) >> 16) & 0xffff);
   const unsigned long long w = (unsigned long long)__double_as_longlong(rec[8]);
   group[i] = (int)(w & 0xffull);
   if (lcg) lcg[i] = make_ulonglong2((unsigned long long)__double_as_longlong(rec[9]), (w >> 8 & 3ull) | (w >> 32) << 32);
   orig[i] = i;
}

/* which owned beads does each neighbour direction need?  (ddcSendRecvTables:
 * every local particle within rcut of the neighbouring domain) */
__global__ void k_halo_select(GridParams gp, DirTab dt, int nloc, int hs_cap, const double4 *pos, int *dir_cnt, int *hs_idx)
{
   /* per-direction counters of the workgroup in LDS, ONE global atomic per direction and workgroup: 26 global
    * counters shared by every bead near a face serialised (42 us for a 500 k-bead brick) */
   __shared__ int s_cnt[27], s_base[27];
   if (threadIdx.x < 27) s_cnt[threadIdx.x] = 0;
   __syncthreads();
   int i = blockIdx.x * blockDim.x + threadIdx.x;
   if (i == 0) dir_cnt[27] = hs_cap;
   unsigned mask = 0;          /* directions this bead goes to */
   if (i < nloc)
   {
      double4 p = pos[i];
      double r[3] = {p.x, p.y, p.z};
      int nlo[3], nhi[3];
#pragma unroll
      for (int a = 0; a < 3; a++)
      {
         double W = gp.n[a] / gp.cinv[a];
         nlo[a] = (gp.m[a] > 0) && (r[a] < gp.lo[a] + gp.rlist);
         nhi[a] = (gp.m[a] > 0) && (r[a] >= gp.lo[a] + W - gp.rlist);
      }
      for (int dz = -1; dz <= 1; dz++)
      {
         if ((dz < 0 && !nlo[2]) || (dz > 0 && !nhi[2])) continue;
         for (int dy = -1; dy <= 1; dy++)
         {
            if ((dy < 0 && !nlo[1]) || (dy > 0 && !nhi[1])) continue;
            for (int dx = -1; dx <= 1; dx++)
            {
               if ((dx < 0 && !nlo[0]) || (dx > 0 && !nhi[0])) continue;
               int code = (dx + 1) + 3 * (dy + 1) + 9 * (dz + 1);
               if (code == 13 || dt.dest[code] < 0) continue;
               mask |= 1u << code;
            }
         }
      }
   }
   /* a bead near a corner goes to 7 directions -- in a domain narrower than two list radii to up to 26: its place
    * inside the workgroup's share of each */
   int loc[27], nl = 0;
   for (unsigned m = mask; m; m &= m - 1) loc[nl++] = atomicAdd(&s_cnt[__ffs((int)m) - 1], 1);
   __syncthreads();
   if (threadIdx.x < 27 && s_cnt[threadIdx.x] > 0) s_base[threadIdx.x] = atomicAdd(&dir_cnt[threadIdx.x], s_cnt[threadIdx.x]);
   __syncthreads();
   nl = 0;
   for (unsigned m = mask; m; m &= m - 1)
   {
      const int code = __ffs((int)m) - 1;
      const int slot = s_base[code] + loc[nl++];
      if (slot < hs_cap) hs_idx[(size_t)code * hs_cap + slot] = i;
   }
}
struct OffTab { int off[28]; };   /* exclusive offsets of the flattened per-direction segments */


/* rebuild: the send map AND the 5-wide records of the beads it names, one launch (a map launch and a pack launch before) */
__global__ void k_send_map_pack5(int nsend, SegTab so, DirTab dt, int hs_cap, const int *__restrict__ hs_idx, unsigned *__restrict__ send_map, double L0, double L1, double L2,
                                 const double4 *__restrict__ pos, const uint64_t *__restrict__ gid, double *__restrict__ out)
{
   int k = blockIdx.x * blockDim.x + threadIdx.x;
   if (k >= nsend) return;
   int q = 0;
   while (k >= so.off[q + 1]) q++;
   const int code = so.code[q];
   const int i = hs_idx[(size_t)code * hs_cap + (k - so.off[q])];
   send_map[k] = (unsigned)i | ((unsigned)code << 27);
   const double4 p = pos[i];
   double *o = out + (size_t)k * 5;
   o[0] = p.x + dt.shift[code][0] * L0;
   o[1] = p.y + dt.shift[code][1] * L1;
   o[2] = p.z + dt.shift[code][2] * L2;
   o[3] = p.w; o[4] = __longlong_as_double((long long)gid[i]);
}
/* pack the beads a remote neighbour needs, shift applied: width 3 (x y z, every
 * step) or 5 (+ tag, gid: at rebuilds) */
/* rebuild: send slot k of the halo buffer <- owned bead and direction, flattened once so that the per-step pack is a plain
 * gather (a search through the segment table and two dependent index loads per bead and step before) */
__global__ void k_send_map(int nsend, SegTab so, int hs_cap, const int *hs_idx, unsigned *send_map)
{
   int k = blockIdx.x * blockDim.x + threadIdx.x;
   if (k >= nsend) return;
   int q = 0;
   while (k >= so.off[q + 1]) q++;
   const int code = so.code[q];
   send_map[k] = (unsigned)hs_idx[(size_t)code * hs_cap + (k - so.off[q])] | ((unsigned)code << 27);
}
/* pack the beads a remote neighbour needs, shift applied: width 3 (x y z, every
 * step) or 5 (+ tag, gid: at rebuilds) */
__global__ void k_pack_halo(int nsend, DirTab dt, const unsigned *__restrict__ send_map, double L0, double L1, double L2,
                            const double4 *__restrict__ pos, const uint64_t *__restrict__ gid, double *__restrict__ out, int width)
{
   int k = blockIdx.x * blockDim.x + threadIdx.x;
   if (k >= nsend) return;
   const unsigned m = send_map[k];
   const int i = (int)(m & 0x7ffffffu), code = (int)(m >> 27);
   double4 p = pos[i];
   double *o = out + (size_t)k * width;
   o[0] = p.x + dt.shift[code][0] * L0;
   o[1] = p.y + dt.shift[code][1] * L1;
   o[2] = p.z + dt.shift[code][2] * L2;
   if (width == 5) { o[3] = p.w; o[4] = __longlong_as_double((long long)gid[i]); }
}
/* halo descriptors: first the local images (directions that wrap onto this rank),
 * then the beads received from other ranks */
__global__ void k_halo_assemble(GridParams gp, int nself, int nrecv, OffTab selfo, SegTab recvo, DirTab dt, int hs_cap, const int *hs_idx,
                                const double4 *pos, const double *hrecv5, int *hsrc, int *hshift, int *hcid, int *hrank, int *cell_cnt_h,
                                const uint64_t *__restrict__ gid, uint64_t *hkey)
{
   int h = blockIdx.x * blockDim.x + threadIdx.x;
   if (h >= nself + nrecv) return;
   double x, y, z;
   int side[3];
   if (h < nself)
   {
      int code = 0;
      while (h >= selfo.off[code + 1]) code++;
      int i = hs_idx[(size_t)code * hs_cap + (h - selfo.off[code])];
      double4 p = pos[i];
      x = p.x + dt.shift[code][0] * gp.L[0]; y = p.y + dt.shift[code][1] * gp.L[1]; z = p.z + dt.shift[code][2] * gp.L[2];
      hsrc[h] = i;
      hkey[h] = gid[i];
      hshift[h] = (dt.shift[code][0] + 1) + 3 * (dt.shift[code][1] + 1) + 9 * (dt.shift[code][2] + 1);
      /* an image sent in direction d appears on the opposite side of this (same) domain */
      side[0] = -(code % 3 - 1); side[1] = -((code / 3) % 3 - 1); side[2] = -(code / 9 - 1);
   }
   else
   {
      int k = h - nself;
      x = hrecv5[5 * k]; y = hrecv5[5 * k + 1]; z = hrecv5[5 * k + 2];
      hsrc[h] = -1 - k;
      hkey[h] = (uint64_t)__double_as_longlong(hrecv5[5 * k + 4]);
      hshift[h] = 13;          /* fixed up below: 27 when the sender applied a periodic shift */
      /* received along the SENDER's direction `code`: it lies on my opposite side */
      int q = 0;
      while (k >= recvo.off[q + 1]) q++;
      const int code = recvo.code[q];
      /* the sender's shift for its direction `code` is minus mine for the opposite direction */
      if (dt.shift[26 - code][0] | dt.shift[26 - code][1] | dt.shift[26 - code][2]) hshift[h] = 27;
      side[0] = -(code % 3 - 1); side[1] = -((code / 3) % 3 - 1); side[2] = -(code / 9 - 1);
   }
   int c = halo_cell(gp, x, y, z, side);
   hcid[h] = c;
   hrank[h] = atomicAdd(&cell_cnt_h[c], 1);
}
__global__ void k_rec5to3(int n, const double *r5, double *r3)
{
   int k = blockIdx.x * blockDim.x + threadIdx.x;
   if (k < n) { r3[3 * k] = r5[5 * k]; r3[3 * k + 1] = r5[5 * k + 1]; r3[3 * k + 2] = r5[5 * k + 2]; }
}
__global__ void k_sum_q2(int n, const double4 *pos, const double *charge_sp, double *out)
{
   int i = blockIdx.x * blockDim.x + threadIdx.x;
   double q = (i < n) ? charge_sp[(int)((__double_as_longlong(pos[i].w) >> 16) & 0xffff)] : 0.0;      /* species sits in the record tag */
   double v = q * q;
   v = wave_sum(v);
   if ((threadIdx.x & 63) == 0 && v != 0.0) atomicAdd(out, v);
}

/* ------------------------------------------------------------------------- */
/* capacity of the owned-bead arrays (beads arrive by migration) */
static int mg_ensure_owned(ddcmi_ctx *ctx, size_t need)
{
   hipStream_t st = ctx->stream;
   size_t want = need + need / 4 + 4096;
   const bool lcg_fits = !ctx->lcg_on || (ctx->lcg.cap >= need + 1 && ctx->lcg2.cap >= need + 1);
   if (ctx->vx.cap >= need && ctx->species.cap >= need + 1 && ctx->pos.cap >= need && lcg_fits) return DDCMI_OK;
   dbuf<double> *d3[] = {&ctx->vx, &ctx->vy, &ctx->vz, &ctx->fx, &ctx->fy, &ctx->fz};
   for (auto b : d3) if (b->ensure(want, true, st)) SETERR(ctx, DDCMI_ENOMEM, "growing bead arrays to %zu failed", want);
   dbuf<double> *d3b[] = {&ctx->vx2, &ctx->vy2, &ctx->vz2};
   for (auto b : d3b) if (b->ensure(want)) SETERR(ctx, DDCMI_ENOMEM, "growing bead arrays to %zu failed", want);
   if (ctx->lcg_on && (ctx->lcg.ensure(want + 1, true, st) || ctx->lcg2.ensure(want + 1))) SETERR(ctx, DDCMI_ENOMEM, "growing the LCG64 records to %zu failed", want);
   dbuf<int> *i1[] = {&ctx->species, &ctx->group, &ctx->orig};
   for (auto b : i1) if (b->ensure(want + 1, true, st)) SETERR(ctx, DDCMI_ENOMEM, "growing bead arrays to %zu failed", want);
   dbuf<int> *i2[] = {&ctx->species2, &ctx->group2, &ctx->orig2, &ctx->slot_of_orig, &ctx->cid, &ctx->crank, &ctx->order, &ctx->nimg, &ctx->img_off};
   for (auto b : i2) if (b->ensure(want + 1)) SETERR(ctx, DDCMI_ENOMEM, "growing bead arrays to %zu failed", want);
   /* keep[]: written by the migration's first phase, read by its second -- which is where a rank that receives more beads than it has room for grows
    * its arrays.  Grown without its contents (rounds 2-6) the second phase compacted by whatever the new allocation held: a memory fault or lost beads
    * on the first rank whose bead count outgrew its first rebuild's headroom (25 % + 4096) between two rebuilds (tools/fuzz_abi.py, round 6) */
   if (ctx->keep.ensure(want + 1, true, st)) SETERR(ctx, DDCMI_ENOMEM, "growing bead arrays to %zu failed", want);
   if (ctx->pos.ensure(want, true, st) || ctx->pos2.ensure(want) || ctx->gid.ensure(want, true, st) || ctx->gid2.ensure(want))
      SETERR(ctx, DDCMI_ENOMEM, "growing bead arrays to %zu failed", want);
   return DDCMI_OK;
}

/* ---- transport ------------------------------------------------------------ */
struct ddcmi_group { std::vector<ddcmi_ctx *> ranks; };

/* loopback: a single rank sends its periodic images to itself through the transport, so the whole
 * exchange (counts, grouped send/recv, message matching, all-reduce) runs on one GPU */
static inline bool dir_remote(const int *dest, int rank, bool loopback, int code) { return code != 13 && dest[code] >= 0 && (dest[code] != rank || loopback); }
static inline bool mg_remote(const ddcmi_ctx *ctx, int code) { return dir_remote(ctx->dir_dest, ctx->rank, ctx->loopback, code); }
static inline bool mg_selfdir(const ddcmi_ctx *ctx, int code) { return code != 13 && ctx->dir_dest[code] == ctx->rank && !ctx->loopback; }
static inline int mg_opp(int code) { return 26 - code; }
static inline bool mg_transport(const ddcmi_ctx *ctx) { return ctx->comm != nullptr || ctx->hcomm != nullptr; }

#define NCCLCHK2(ctx, call) do { ncclResult_t _r = (call); if (_r != ncclSuccess) SETERR(ctx, DDCMI_ECOMM, "%s failed: %s", #call, ncclGetErrorString(_r)); } while (0)
#define HOSTCHK(ctx, call) do { int _r = (call); if (_r != DDCMI_OK) SETERR(ctx, DDCMI_ECOMM, "%s failed: %s", #call, ddcmi_rdzv_last_error((ctx)->hcomm)); } while (0)

/* ---- host logic shared by every transport (and callable without a GPU) ------ */
/* what this rank receives, from the all-gathered send counts: the message a neighbour sends along ITS
 * direction `code` comes from the rank in my direction opp(code) */
static void plan_recv_counts(const int *dest, int rank, bool loopback, const int *all_counts, int *rcnt)
{
   for (int code = 0; code < 27; code++)
      rcnt[code] = dir_remote(dest, rank, loopback, mg_opp(code)) ? all_counts[27 * dest[mg_opp(code)] + code] : 0;
}
/* buffer layout of the halo exchange from the per-direction counts: remote segments ordered by
 * (peer rank, direction code) on both sides, so the segments of one peer are contiguous and travel as
 * ONE message (7 at 2x2x2 instead of 26); send_off/recv_off[code] index the same layout */
static void plan_halo_layout(const int *dest, int rank, int nranks, bool loopback, const int *scnt, const int *rcnt,
                             SegTab &ss, SegTab &rs, int *send_off, int *recv_off, int *nsend, int *nrecv, HaloMsgs &ms, HaloMsgs &mr)
{
   ss.nseg = rs.nseg = 0;
   int ns = 0, nr = 0;
   for (int code = 0; code < 28; code++) { send_off[code] = 0; recv_off[code] = 0; }
   for (int peer = 0; peer < nranks; peer++)
      for (int code = 0; code < 27; code++)
      {
         if (dir_remote(dest, rank, loopback, code) && dest[code] == peer)
         { ss.code[ss.nseg] = (signed char)code; ss.off[ss.nseg++] = ns; send_off[code] = ns; ns += scnt[code]; }
         if (dir_remote(dest, rank, loopback, mg_opp(code)) && dest[mg_opp(code)] == peer)
         { rs.code[rs.nseg] = (signed char)code; rs.off[rs.nseg++] = nr; recv_off[code] = nr; nr += rcnt[code]; }
      }
   ss.off[ss.nseg] = ns; rs.off[rs.nseg] = nr;
   for (int q = ss.nseg + 1; q < 28; q++) ss.off[q] = ns;
   for (int q = rs.nseg + 1; q < 28; q++) rs.off[q] = nr;
   send_off[27] = ns; recv_off[27] = nr;
   *nsend = ns; *nrecv = nr;
   ms.n = mr.n = 0;
   for (int q = 0; q < ss.nseg;)
   {
      int peer = dest[(int)ss.code[q]], q1 = q;
      while (q1 < ss.nseg && dest[(int)ss.code[q1]] == peer) q1++;
      int cnt = ss.off[q1] - ss.off[q];
      if (cnt > 0) { ms.peer[ms.n] = peer; ms.off[ms.n] = ss.off[q]; ms.cnt[ms.n] = cnt; ms.n++; }
      q = q1;
   }
   for (int q = 0; q < rs.nseg;)
   {
      /* rseg.code = the SENDER's direction: it arrives from the rank in my opposite direction */
      int peer = dest[mg_opp((int)rs.code[q])], q1 = q;
      while (q1 < rs.nseg && dest[mg_opp((int)rs.code[q1])] == peer) q1++;
      int cnt = rs.off[q1] - rs.off[q];
      if (cnt > 0) { mr.peer[mr.n] = peer; mr.off[mr.n] = rs.off[q]; mr.cnt[mr.n] = cnt; mr.n++; }
      q = q1;
   }
}
extern "C" int ddcmi_plan_recv_counts(int px, int py, int pz, int rank, int pbc, int loopback, const int *all_counts, int *recv_cnt)
{
   int dest[27], shift[81];
   if (!all_counts || !recv_cnt) return DDCMI_EINVAL;
   int rc = ddcmi_plan_directions(px, py, pz, rank, pbc, dest, shift);
   if (rc) return rc;
   plan_recv_counts(dest, rank, loopback != 0, all_counts, recv_cnt);
   return DDCMI_OK;
}
extern "C" int ddcmi_plan_halo_layout(int px, int py, int pz, int rank, int pbc, int loopback, const int send_cnt[27], const int recv_cnt[27],
                                      int send_off[28], int recv_off[28], int *msgs, int *msgr)
{
   int dest[27], shift[81];
   if (!send_cnt || !recv_cnt || !send_off || !recv_off || !msgs || !msgr) return DDCMI_EINVAL;
   int rc = ddcmi_plan_directions(px, py, pz, rank, pbc, dest, shift);
   if (rc) return rc;
   SegTab ss, rs; HaloMsgs ms, mr;
   int ns, nr;
   plan_halo_layout(dest, rank, px * py * pz, loopback != 0, send_cnt, recv_cnt, ss, rs, send_off, recv_off, &ns, &nr, ms, mr);
   msgs[0] = ms.n; msgr[0] = mr.n;
   for (int k = 0; k < ms.n; k++) { msgs[1 + 3 * k] = ms.peer[k]; msgs[2 + 3 * k] = ms.off[k]; msgs[3 + 3 * k] = ms.cnt[k]; }
   for (int k = 0; k < mr.n; k++) { msgr[1 + 3 * k] = mr.peer[k]; msgr[2 + 3 * k] = mr.off[k]; msgr[3 + 3 * k] = mr.cnt[k]; }
   return DDCMI_OK;
}

/* ---- transports ------------------------------------------------------------- */
/* Every rank learns every rank's 27 per-direction counts (+ the capacity its segments had, slot 27) with ONE all-gather
 * straight from the device counters (26 four-byte point-to-point messages took 77 us; this is rebuild-time control
 * traffic) and looks up what its neighbours send to it.  One host synchronisation per round.
 * The block of a rank is 32 ints: slot 28 carries the rank's pending error code (local_err: an allocation or a check that
 * failed on the way to this round), slot 29 the device's "a bead moved further than one domain" flag -- so a hard error
 * of ONE rank ends the rebuild on EVERY rank in the same round, instead of leaving the others inside the next exchange.
 * RCCL: the all-gather reads dir_cnt on the device; host transport: the rendezvous' all-gather after the download. */
#define MG_BLK 32
static int mg_counts_round(ddcmi_ctx *ctx, int *scnt, int *rcnt, bool *any_over, int *my_max, bool with_flags, int local_err)
{
   hipStream_t st = ctx->stream;
   const int nr = std::max(ctx->nranks, 1);
   { int rca = ddcmi_agree_poll(ctx); if (rca) return rca; }      /* the last rebuild's agreement, before this rank waits on a collective again */
   const std::string local_msg = ctx->err;
   int *h = ctx->pinned(2, 64 + (MG_BLK + 27) * (size_t)nr);
   if (!h) SETERR(ctx, DDCMI_ENOMEM, "pinned staging for the count exchange");
   int *all = h + 64, *packed = h + 64 + MG_BLK * (size_t)nr;
   if (with_flags && ctx->hcomm) HIPCHK(ctx, hipMemcpyAsync(ctx->h_flags, ctx->d_flags, 8 * sizeof(int), hipMemcpyDeviceToHost, st));
   if (ctx->dir_cnt.cap < MG_BLK && ctx->dir_cnt.ensure(MG_BLK)) SETERR(ctx, DDCMI_ENOMEM, "direction counters");
   if (local_err)
   {
      /* this rank takes part with empty segments and its error code */
      memset(h + 32, 0, MG_BLK * sizeof(int));
      h[32 + 28] = local_err;
      HIPCHK(ctx, hipMemcpyAsync(ctx->dir_cnt.p, h + 32, MG_BLK * sizeof(int), hipMemcpyHostToDevice, st));
   }
   if (ctx->hcomm)
   {
      HIPCHK(ctx, hipMemcpyAsync(h, ctx->dir_cnt.p, MG_BLK * sizeof(int), hipMemcpyDeviceToHost, st));
      HIPCHK(ctx, hipStreamSynchronize(st));
      HOSTCHK(ctx, ddcmi_rdzv_allgather(ctx->hcomm, h, all, MG_BLK * sizeof(int)));
   }
   else
   {
      ENSURE(ctx, ctx->cnt_xchg, 32 + MG_BLK * (size_t)nr);
      NCCLCHK2(ctx, ncclAllGather(ctx->dir_cnt.p, ctx->cnt_xchg.p + 32, MG_BLK, ncclInt, (ncclComm_t)ctx->comm, st));
      /* every rank's block (and this rank's flags) through the mailbox: no copy call that waits, no late wake-up (scan.hip) */
      PostJobs pj;
      pj.add(ctx->cnt_xchg.p + 32, MG_BLK * (size_t)nr).add(ctx->d_flags, 8);
      int rcp;
      if ((rcp = ddcmi_post(ctx, st, pj)) || (rcp = ddcmi_post_wait(ctx, st))) return rcp;
      memcpy(all, ctx->mbox_h + pj.off[0], MG_BLK * (size_t)nr * sizeof(int));
      if (with_flags) memcpy(ctx->h_flags, ctx->mbox_h + pj.off[1], 8 * sizeof(int));
   }
   *any_over = false;
   for (int r = 0; r < nr; r++)
   {
      int mx = 0;
      for (int c = 0; c < 27; c++) { mx = std::max(mx, all[MG_BLK * r + c]); packed[27 * r + c] = all[MG_BLK * r + c]; }
      if (mx > all[MG_BLK * r + 27]) *any_over = true;
      if (r == ctx->rank) *my_max = mx;
   }
   /* errors first, the lowest rank's: every rank returns from the same round */
   for (int r = 0; r < nr; r++)
   {
      if (all[MG_BLK * r + 28])
      {
         if (r == ctx->rank) { ctx->err = local_msg; return all[MG_BLK * r + 28]; }
         SETERR(ctx, DDCMI_ECOMM, "rank %d failed during the list rebuild (error %d): its own message says why", r, all[MG_BLK * r + 28]);
      }
      if (all[MG_BLK * r + 29]) SETERR(ctx, DDCMI_EINVAL, "a bead moved further than one domain between rebuilds (rank %d)", r);
      if (all[MG_BLK * r + 30])
         SETERR(ctx, DDCMI_EINVAL, "%d beads of rank %d have non-finite coordinates or lie more than a box length outside the box at loop %lld: the run is unstable (time step, overlapping start, singular bonded term?)",
                all[MG_BLK * r + 30], r, (long long)ctx->loop);
   }
   for (int c = 0; c < 27; c++) scnt[c] = all[MG_BLK * (size_t)ctx->rank + c];
   plan_recv_counts(ctx->dir_dest, ctx->rank, ctx->loopback, packed, rcnt);
   return DDCMI_OK;
}
static int mg_fatal(ddcmi_ctx *ctx, int code, const char *msg);
/* the ranks agree on the outcome of a phase that has no count round behind it: max of the error codes (one small collective) */
static int mg_agree(ddcmi_ctx *ctx, int local_rc)
{
   if ((ctx->nranks == 1 && !ctx->loopback) || !mg_transport(ctx)) return local_rc;
   const std::string local_msg = ctx->err;
   hipStream_t st = ctx->stream;
   int worst = local_rc < 0 ? -local_rc : local_rc;
   if (ctx->hcomm)
   {
      double v = (double)worst;
      HOSTCHK(ctx, ddcmi_rdzv_allreduce_f64(ctx->hcomm, &v, 1, 1));
      worst = (int)v;
   }
   else
   {
      int *h = ctx->pinned(2, 64);
      if (!h) SETERR(ctx, DDCMI_ENOMEM, "pinned staging for the count exchange");
      ENSURE(ctx, ctx->cnt_xchg, 32);
      h[0] = worst;
      HIPCHK(ctx, hipMemcpyAsync(ctx->cnt_xchg.p, h, sizeof(int), hipMemcpyHostToDevice, st));
      NCCLCHK2(ctx, ncclAllReduce(ctx->cnt_xchg.p, ctx->cnt_xchg.p + 1, 1, ncclInt, ncclMax, (ncclComm_t)ctx->comm, st));
      PostJobs pj;
      pj.add(ctx->cnt_xchg.p + 1, 1);
      int rcp;
      if ((rcp = ddcmi_post(ctx, st, pj)) || (rcp = ddcmi_post_wait(ctx, st))) return rcp;
      worst = ctx->mbox_h[pj.off[0]];
   }
   if (local_rc) { ctx->err = local_msg; return local_rc; }
   if (worst) SETERR(ctx, DDCMI_ECOMM, "another rank failed during the list rebuild (error %d): its own message says why", worst);
   return DDCMI_OK;
}
/* The same agreement without anybody waiting for it (RCCL transport, rebuilds after the first two).  ADVICE r3: from the third
 * rebuild on a rank whose local phase failed (a bond stretched beyond the halo in a system going unstable, a capacity, an
 * allocation) used to abort its communicator and return, and its peers sat in the next halo kernel until something outside
 * killed the job.  Now every rank contributes its error code to an all-reduce (max) queued behind the phase's kernels, a post
 * kernel leaves the result in mapped host memory, and nobody waits: the failing rank returns its own error once the
 * collective has run; a healthy rank goes on queueing steps and looks at the word in front of its next host wait (ddcmi_agree_poll:
 * the next rebuild's count round, a read of energies, a download, ddcmi_sync) -- the word lands BEFORE the exchange the failed
 * peer never joins, so the look cannot hang, and a reported failure aborts the communicator (which releases the waiting
 * kernels) and returns DDCMI_ECOMM naming the rebuild.  Cost: one small collective per rebuild on the stream, no host wait. */
__global__ void k_agree_post(const int *src, int *dst, int seq)
{
   if (threadIdx.x == 0)
   {
      dst[1] = *src;
      __threadfence_system();
      __hip_atomic_store(dst, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
   }
}
static int mg_agree_async(ddcmi_ctx *ctx, int local_rc, int pretend_peer_code = 0 /* tests: the code a failed peer would have contributed */)
{
   hipStream_t st = ctx->stream;
   if (!ctx->agree_h)
   {
      if (hipHostMalloc((void **)&ctx->agree_h, 16 * sizeof(int), hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess) { ctx->agree_h = nullptr; return mg_agree(ctx, local_rc); }
      if (hipHostGetDevicePointer((void **)&ctx->agree_d, ctx->agree_h, 0) != hipSuccess) { (void)hipHostFree(ctx->agree_h); ctx->agree_h = nullptr; return mg_agree(ctx, local_rc); }
      memset(ctx->agree_h, 0, 16 * sizeof(int));
   }
   { int rcp = ddcmi_agree_poll(ctx); if (rcp && !local_rc) return rcp; }      /* (never two agreements in flight) */
   const std::string local_msg = ctx->err;
   int *h = ctx->pinned(2, 64);
   if (!h) return mg_agree(ctx, local_rc);
   h[0] = std::max(local_rc < 0 ? -local_rc : local_rc, pretend_peer_code);
   int *d = ctx->d_flags + DDCMI_FLAG_AGREE;
   /* (a rebuild that went well contributes the zero its tail launch left there: no copy) */
   if (h[0] != 0 || !ctx->list_valid) HIPCHK(ctx, hipMemcpyAsync(d, h, sizeof(int), hipMemcpyHostToDevice, st));
   NCCLCHK2(ctx, ncclAllReduce(d, d + 1, 1, ncclInt, ncclMax, (ncclComm_t)ctx->comm, st));
   ctx->agree_seq++;
   hipLaunchKernelGGL(k_agree_post, dim3(1), dim3(64), 0, st, d + 1, ctx->agree_d, ctx->agree_seq);
   ctx->agree_pending = true; ctx->agree_loop = ctx->loop;
   if (local_rc)
   {
      /* this rank's own failure: it has told the others; it returns once the collective has run (its peers all join it) */
      (void)hipStreamSynchronize(st);
      ctx->agree_pending = false;
      ctx->err = local_msg;
      return local_rc;
   }
   return DDCMI_OK;
}
int ddcmi_agree_poll(ddcmi_ctx *ctx)
{
   if (!ctx->agree_pending) return DDCMI_OK;
   volatile int *flag = ctx->agree_h;
   struct timespec t0; clock_gettime(CLOCK_MONOTONIC, &t0);
   for (unsigned long spin = 0;; spin++)
   {
      if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) == ctx->agree_seq) break;
      if ((spin & 0xff) == 0xff)
      {
         struct timespec t1; clock_gettime(CLOCK_MONOTONIC, &t1);
         if ((t1.tv_sec - t0.tv_sec) + 1e-9 * (t1.tv_nsec - t0.tv_nsec) > 20.0)
         {
            ctx->agree_pending = false;
            return mg_fatal(ctx, DDCMI_ECOMM, "the ranks' agreement on the last list rebuild did not arrive within 20 s: a peer rank is gone");
         }
         sched_yield();
      }
   }
   ctx->agree_pending = false;
   const int worst = ctx->agree_h[1];
   if (worst)
   {
      char b[256];
      snprintf(b, sizeof(b), "another rank failed during the list rebuild at loop %lld (error %d): its own message says why", (long long)ctx->agree_loop, worst);
      return mg_fatal(ctx, DDCMI_ECOMM, b);      /* (aborting the communicator releases this rank's kernels that wait for the peer) */
   }
   return DDCMI_OK;
}
/* a rank that cannot go on at a point where its peers already wait for its data leaves the job: the host transport's
 * streams are closed (the peers' receives fail at once with "peer closed"), an RCCL communicator is aborted (peers inside
 * a kernel are released when the launcher tears the job down on this rank's non-zero exit) */
static int mg_fatal(ddcmi_ctx *ctx, int code, const char *msg)
{
   if (msg) ctx->err = msg;
   if (ctx->hcomm) ddcmi_rdzv_abort(ctx->hcomm);
   else if (ctx->comm) { (void)ncclCommAbort((ncclComm_t)ctx->comm); ctx->comm = nullptr; }
   return code;
}
/* host transport: device segments -> host staging -> TCP streams -> host staging -> device segments.
 * nm* messages; offsets and counts in doubles */
static int mg_host_exchange(ddcmi_ctx *ctx, hipStream_t st, int nms, const int *speer, const double *const *sdev, const size_t *scount,
                            int nmr, const int *rpeer, double *const *rdev, const size_t *rcount)
{
   size_t stot = 0, rtot = 0;
   for (int k = 0; k < nms; k++) stot += scount[k];
   for (int k = 0; k < nmr; k++) rtot += rcount[k];
   ctx->hstage_s.resize(stot + 1); ctx->hstage_r.resize(rtot + 1);
   const void *sb[27]; void *rb[27]; size_t sbytes[27], rbytes[27];
   size_t o = 0;
   for (int k = 0; k < nms; k++)
   {
      HIPCHK(ctx, hipMemcpyAsync(ctx->hstage_s.data() + o, sdev[k], scount[k] * sizeof(double), hipMemcpyDeviceToHost, st));
      sb[k] = ctx->hstage_s.data() + o; sbytes[k] = scount[k] * sizeof(double); o += scount[k];
   }
   HIPCHK(ctx, hipStreamSynchronize(st));
   o = 0;
   for (int k = 0; k < nmr; k++) { rb[k] = ctx->hstage_r.data() + o; rbytes[k] = rcount[k] * sizeof(double); o += rcount[k]; }
   HOSTCHK(ctx, ddcmi_rdzv_exchange(ctx->hcomm, nms, speer, sb, sbytes, nmr, rpeer, rb, rbytes));
   o = 0;
   for (int k = 0; k < nmr; k++)
   {
      HIPCHK(ctx, hipMemcpyAsync(rdev[k], ctx->hstage_r.data() + o, rcount[k] * sizeof(double), hipMemcpyHostToDevice, st));
      o += rcount[k];
   }
   HIPCHK(ctx, hipStreamSynchronize(st));      /* the staging vector is reused by the next exchange */
   return DDCMI_OK;
}
/* migration records: one message per direction (rebuilds only); segments are flattened
 * with the given offsets (send: by my direction code; recv: by the SENDER's direction code) */
static int mg_xchg_data(ddcmi_ctx *ctx, const double *sbase, const int *soff, const int *scnt, size_t sstride_items,
                        double *rbase, const int *roff, const int *rcnt, int width)
{
   hipStream_t st = ctx->stream;
   if (ctx->hcomm)
   {
      int speer[27], rpeer[27], nms = 0, nmr = 0;
      const double *sdev[27]; double *rdev[27]; size_t sc[27], rc_[27];
      for (int code = 0; code < 27; code++)
      {
         if (mg_remote(ctx, code) && scnt[code] > 0)
         {
            sdev[nms] = sstride_items ? sbase + (size_t)code * sstride_items * width : sbase + (size_t)soff[code] * width;
            sc[nms] = (size_t)scnt[code] * width; speer[nms++] = ctx->dir_dest[code];
         }
         if (mg_remote(ctx, mg_opp(code)) && rcnt[code] > 0)
         { rdev[nmr] = rbase + (size_t)roff[code] * width; rc_[nmr] = (size_t)rcnt[code] * width; rpeer[nmr++] = ctx->dir_dest[mg_opp(code)]; }
      }
      return mg_host_exchange(ctx, st, nms, speer, sdev, sc, nmr, rpeer, rdev, rc_);
   }
   ncclComm_t comm = (ncclComm_t)ctx->comm;
   NCCLCHK2(ctx, ncclGroupStart());
   for (int code = 0; code < 27; code++)
   {
      if (mg_remote(ctx, code) && scnt[code] > 0)
      {
         const double *src = sstride_items ? sbase + (size_t)code * sstride_items * width : sbase + (size_t)soff[code] * width;
         NCCLCHK2(ctx, ncclSend(src, (size_t)scnt[code] * width, ncclDouble, ctx->dir_dest[code], comm, st));
      }
      if (mg_remote(ctx, mg_opp(code)) && rcnt[code] > 0)
         NCCLCHK2(ctx, ncclRecv(rbase + (size_t)roff[code] * width, (size_t)rcnt[code] * width, ncclDouble, ctx->dir_dest[mg_opp(code)], comm, st));
   }
   NCCLCHK2(ctx, ncclGroupEnd());
   return DDCMI_OK;
}
/* halo beads (every step): each peer pair exchanges ONE message (mg_layout_halo), which is what
 * the grouped point-to-point kernel's time follows */
static int mg_xchg_halo(ddcmi_ctx *ctx, const double *sbase, double *rbase, int width, hipStream_t st)
{
   const HaloMsgs &ms = ctx->hmsg_s, &mr = ctx->hmsg_r;
   if (ctx->hcomm)
   {
      const double *sdev[27]; double *rdev[27]; size_t sc[27], rc_[27];
      for (int k = 0; k < ms.n; k++) { sdev[k] = sbase + (size_t)ms.off[k] * width; sc[k] = (size_t)ms.cnt[k] * width; }
      for (int k = 0; k < mr.n; k++) { rdev[k] = rbase + (size_t)mr.off[k] * width; rc_[k] = (size_t)mr.cnt[k] * width; }
      return mg_host_exchange(ctx, st, ms.n, ms.peer, sdev, sc, mr.n, mr.peer, rdev, rc_);
   }
   ncclComm_t comm = (ncclComm_t)ctx->comm;
   /* DDCMI_DEBUG_SPLIT_MSGS=<k> (with DDCMI_DEBUG_HOOKS=1, loopback only): every message travels as k send/recv pairs -- what a rank with k peers hands
    * RCCL per step (2x2x2: seven), on the one GPU there is to measure on: the same bytes, k times the operations of the group */
   static const int split = (getenv("DDCMI_DEBUG_HOOKS") && getenv("DDCMI_DEBUG_SPLIT_MSGS")) ? std::max(1, atoi(getenv("DDCMI_DEBUG_SPLIT_MSGS"))) : 1;
   NCCLCHK2(ctx, ncclGroupStart());
   if (split > 1 && ctx->loopback)
   {
      for (int k = 0; k < ms.n; k++)
         for (int q = 0; q < split; q++)
         {
            const size_t a = (size_t)ms.cnt[k] * q / split, b = (size_t)ms.cnt[k] * (q + 1) / split;
            if (b > a) NCCLCHK2(ctx, ncclSend(sbase + ((size_t)ms.off[k] + a) * width, (b - a) * width, ncclDouble, ms.peer[k], comm, st));
         }
      for (int k = 0; k < mr.n; k++)
         for (int q = 0; q < split; q++)
         {
            const size_t a = (size_t)mr.cnt[k] * q / split, b = (size_t)mr.cnt[k] * (q + 1) / split;
            if (b > a) NCCLCHK2(ctx, ncclRecv(rbase + ((size_t)mr.off[k] + a) * width, (b - a) * width, ncclDouble, mr.peer[k], comm, st));
         }
      NCCLCHK2(ctx, ncclGroupEnd());
      return DDCMI_OK;
   }
   for (int k = 0; k < ms.n; k++) NCCLCHK2(ctx, ncclSend(sbase + (size_t)ms.off[k] * width, (size_t)ms.cnt[k] * width, ncclDouble, ms.peer[k], comm, st));
   for (int k = 0; k < mr.n; k++) NCCLCHK2(ctx, ncclRecv(rbase + (size_t)mr.off[k] * width, (size_t)mr.cnt[k] * width, ncclDouble, mr.peer[k], comm, st));
   NCCLCHK2(ctx, ncclGroupEnd());
   return DDCMI_OK;
}
static void mg_layout_halo(ddcmi_ctx *ctx)
{
   plan_halo_layout(ctx->dir_dest, ctx->rank, ctx->nranks, ctx->loopback, ctx->hs_cnt, ctx->hr_cnt, ctx->sseg, ctx->rseg,
                    ctx->send_off, ctx->recv_off, &ctx->nsend, &ctx->nrecv, ctx->hmsg_s, ctx->hmsg_r);
}
/* in-process emulation: same matching rule, direct device copies */
static int mg_xchg_data_local(ddcmi_group *g, int which /*0 migration, 1 halo5, 2 halo3*/)
{
   for (ddcmi_ctx *A : g->ranks) HIPCHK(A, hipStreamSynchronize(A->stream));
   for (ddcmi_ctx *A : g->ranks)
      for (int code = 0; code < 27; code++)
      {
         if (!mg_remote(A, code)) continue;
         ddcmi_ctx *B = g->ranks[A->dir_dest[code]];
         if (which == 0)
         {
            int n = A->mig_scnt[code];
            if (n <= 0) continue;
            int roff = 0;
            for (int c = 0; c < code; c++) roff += B->mig_rcnt[c];
            HIPCHK(A, hipMemcpyAsync(B->mig_in.p + (size_t)roff * 10, A->mig_out.p + (size_t)code * A->mig_cap * 10, (size_t)n * 10 * sizeof(double), hipMemcpyDeviceToDevice, A->stream));
         }
         else
         {
            int n = A->hs_cnt[code];
            if (n <= 0) continue;
            int w = (which == 1) ? 5 : 3;
            double *dst = (which == 1 ? B->hrecv5.p : B->hrecv3.p) + (size_t)B->recv_off[code] * w;
            HIPCHK(A, hipMemcpyAsync(dst, A->sendbuf.p + (size_t)A->send_off[code] * w, (size_t)n * w * sizeof(double), hipMemcpyDeviceToDevice, A->stream));
         }
      }
   /* a device-to-device hipMemcpy on the null stream is ordered neither with the host nor with the ranks'
    * non-blocking streams: the copies go on the sender's stream and every stream is drained before the
    * receivers' kernels are queued */
   for (ddcmi_ctx *A : g->ranks) HIPCHK(A, hipStreamSynchronize(A->stream));
   return DDCMI_OK;
}

/* ---- rebuild phases --------------------------------------------------------- */
/* enqueue the classification of the owned beads (stay / leave towards one of 26 directions); no host round trip */
static int mg_phase1_launch(ddcmi_ctx *ctx)
{
   hipStream_t st = ctx->stream;
   int n = ctx->nloc, rc;
   if (!ctx->bonded_gid && ctx->nbond + ctx->nangle + ctx->ntors > 0)
      SETERR(ctx, DDCMI_EUNSUPPORTED, "with domain decomposition bonded terms must be given by gid (ddcmi_set_bonded_gid): caller-order indices do not survive migration");
   if ((rc = mg_ensure_owned(ctx, (size_t)n + 1))) return rc;
   if (ctx->mig_cap == 0) ctx->mig_cap = std::max(1024, n / 16);
   ENSURE(ctx, ctx->mig_out, (size_t)27 * ctx->mig_cap * 10);
   ENSURE(ctx, ctx->dir_cnt, 32);
   ENSURE(ctx, ctx->keep, (size_t)n + 1);
   if (!ctx->dircnt_clean) ddcmi_zero_ints(ctx, st, ZeroJobs().add(ctx->dir_cnt.p, 32).add(ctx->d_flags, 8));      /* (else: the last rebuild's tail launch left them zeroed) */
   ctx->dircnt_clean = false;
   MigGeom mg;
   for (int a = 0; a < 3; a++) { mg.L[a] = ctx->h[4 * a]; mg.P[a] = ctx->pgrid[a]; mg.W[a] = mg.L[a] / mg.P[a]; mg.pc[a] = ctx->pcoord[a]; }
   mg.pbc = ctx->pbc;
   if (n > 0)
   {
      hipLaunchKernelGGL(k_mig_classify, dim3(cdiv(n, 256)), dim3(256), 0, st, mg, n, ctx->mig_cap, ctx->pos.p, ctx->vx.p, ctx->vy.p, ctx->vz.p,
                         ctx->gid.p, ctx->group.p, ctx->keep.p, ctx->dir_cnt.p, ctx->mig_out.p, ctx->d_flags, ctx->lcg_on ? ctx->lcg.p : (const ulonglong2 *)nullptr, ctx->orig.p);
   }
   return DDCMI_OK;
}
/* in-process groups: the counts come to the host per context */
static int mg_phase1_migrate_out(ddcmi_ctx *ctx)
{
   bl_drop_interior(ctx);      /* (in-process groups enter the rebuild here: see ddcmi_build_list) */
   hipStream_t st = ctx->stream;
   int rc;
   if ((rc = bl_validate_tables(ctx))) return rc;
   for (;;)
   {
      if ((rc = mg_phase1_launch(ctx))) return rc;
      HIPCHK(ctx, hipMemcpyAsync(ctx->mig_scnt, ctx->dir_cnt.p, 27 * sizeof(int), hipMemcpyDeviceToHost, st));
      HIPCHK(ctx, hipMemcpyAsync(ctx->h_flags, ctx->d_flags, 8 * sizeof(int), hipMemcpyDeviceToHost, st));
      HIPCHK(ctx, hipStreamSynchronize(st));
      if (ctx->h_flags[6]) SETERR(ctx, DDCMI_EINVAL, "a bead moved further than one domain between rebuilds");
      int mx = 0;
      for (int c = 0; c < 27; c++) mx = std::max(mx, ctx->mig_scnt[c]);
      if (mx <= ctx->mig_cap) break;
      ctx->mig_cap = mx + mx / 4 + 64;     /* positions were only wrapped (idempotent): simply redo */
   }
   return DDCMI_OK;
}
static int mg_phase2_migrate_in(ddcmi_ctx *ctx)
{
   /* mig_rcnt is known and mig_in holds the arrivals (flattened by sender direction) */
   hipStream_t st = ctx->stream;
   int n = ctx->nloc, rc;
   int narr = 0;
   for (int c = 0; c < 27; c++) narr += ctx->mig_rcnt[c];
   int nleave = 0;
   for (int c = 0; c < 27; c++) nleave += ctx->mig_scnt[c];
   int nkeep = n - nleave;
   if ((rc = mg_ensure_owned(ctx, (size_t)nkeep + narr + 1))) return rc;
   if (nleave > 0)
   {
      if ((rc = ddcmi_scan_exclusive(ctx, ctx->keep.p, ctx->img_off.p, n, nullptr))) return rc;
      hipLaunchKernelGGL(k_compact_order, dim3(cdiv(n, 256)), dim3(256), 0, st, n, ctx->keep.p, ctx->img_off.p, ctx->order.p);
      if (nkeep > 0)
      {
         hipLaunchKernelGGL(k_gather_state, dim3(cdiv(nkeep, 256)), dim3(256), 0, st, nkeep, ctx->order.p, ctx->pos.p, ctx->vx.p, ctx->vy.p, ctx->vz.p,
                            ctx->species.p, ctx->group.p, ctx->gid.p, ctx->orig.p,
                            ctx->pos2.p, ctx->vx2.p, ctx->vy2.p, ctx->vz2.p, ctx->species2.p, ctx->group2.p, ctx->gid2.p, ctx->orig2.p, ctx->slot_of_orig.p, ctx->gp, (int *)nullptr, 0,
                            ctx->lcg_on ? ctx->lcg.p : (const ulonglong2 *)nullptr, ctx->lcg2.p);
      }
      if (ctx->lcg_on) std::swap(ctx->lcg, ctx->lcg2);
      std::swap(ctx->pos, ctx->pos2); std::swap(ctx->vx, ctx->vx2); std::swap(ctx->vy, ctx->vy2); std::swap(ctx->vz, ctx->vz2);
      std::swap(ctx->species, ctx->species2); std::swap(ctx->group, ctx->group2); std::swap(ctx->gid, ctx->gid2); std::swap(ctx->orig, ctx->orig2);
   }
   if (narr > 0)
      hipLaunchKernelGGL(k_unpack_mig, dim3(cdiv(narr, 256)), dim3(256), 0, st, narr, nkeep, ctx->mig_in.p, ctx->pos.p, ctx->vx.p, ctx->vy.p, ctx->vz.p,
                         ctx->gid.p, ctx->species.p, ctx->group.p, ctx->orig.p, ctx->lcg_on ? ctx->lcg.p : (ulonglong2 *)nullptr);
   ctx->nloc = nkeep + narr;
   n = ctx->nloc;
   ctx->sort_renumbers = true;      /* (the sort's first kernel numbers the beads: was a launch of its own) */
   /* sort the owned beads, then pick what each neighbour direction needs */
   if ((rc = ddcmi_bl_sort_owned(ctx))) return rc;
   return bl_launch_interior(ctx);      /* the tiles that need nothing of the halo: searched while the halo is exchanged and sorted */
}
/* which owned beads does each neighbour direction need?  enqueue only */
static int mg_halo_select_launch(ddcmi_ctx *ctx)
{
   hipStream_t st = ctx->stream;
   const int n = ctx->nloc;
   if (ctx->hs_cap == 0) ctx->hs_cap = std::max(4096, n / 4);
   ENSURE(ctx, ctx->hs_idx, (size_t)27 * ctx->hs_cap);
   if (!ctx->dir28_clean) ddcmi_zero_ints(ctx, st, ZeroJobs().add(ctx->dir_cnt.p, 28));      /* (slots 28-31 keep what the rebuild has flagged so far: k_wrap_cell's slot 30; else: the sort's first kernel zeroed them) */
   ctx->dir28_clean = false;
   if (n > 0)
      hipLaunchKernelGGL(k_halo_select, dim3(cdiv(n, 256)), dim3(256), 0, st, ctx->gp, mg_dirtab(ctx), n, ctx->hs_cap, ctx->pos.p, ctx->dir_cnt.p, ctx->hs_idx.p);
   return DDCMI_OK;
}
/* in-process groups: counts to the host per context */
static int mg_halo_select(ddcmi_ctx *ctx)
{
   hipStream_t st = ctx->stream;
   int rc;
   for (;;)
   {
      if ((rc = mg_halo_select_launch(ctx))) return rc;
      int h[32];
      HIPCHK(ctx, hipMemcpyAsync(h, ctx->dir_cnt.p, 32 * sizeof(int), hipMemcpyDeviceToHost, st));
      HIPCHK(ctx, hipStreamSynchronize(st));
      memcpy(ctx->hs_cnt, h, 27 * sizeof(int));
      /* the sort's "these beads are not numbers" count (slot 30, k_wrap_cell): a transport's ranks learn of it in the halo count round; the domains of a
       * group stop here too -- before a neighbour files such a bead into a halo cell (tools/fuzz_abi.py: an LJ table entry of 1e30 on 2x2x2 bricks) */
      if (h[30] > 0)
         SETERR(ctx, DDCMI_EINVAL, "%d beads have non-finite coordinates or lie more than a box length outside the box at loop %lld: the run is unstable (time step, overlapping start, singular bonded term?)",
                h[30], (long long)ctx->loop);
      int mx = 0;
      for (int c = 0; c < 27; c++) mx = std::max(mx, ctx->hs_cnt[c]);
      if (mx <= ctx->hs_cap) break;
      ctx->hs_cap = mx + mx / 4 + 64;
   }
   return DDCMI_OK;
}
/* after the halo counts are exchanged: offsets, buffers, pack the rebuild records */
static int mg_phase3_pack(ddcmi_ctx *ctx, int width)
{
   hipStream_t st = ctx->stream;
   mg_layout_halo(ctx);
   const int ns = ctx->nsend, nr = ctx->nrecv;
   ENSURE(ctx, ctx->sendbuf, (size_t)ns * 5 + 8); ENSURE(ctx, ctx->hrecv5, (size_t)nr * 5 + 8); ENSURE(ctx, ctx->hrecv3, (size_t)nr * 3 + 8);
   if (ns > 0)
   {
      if (ctx->nloc >= (1 << 27)) SETERR(ctx, DDCMI_EUNSUPPORTED, "%d beads on one rank: more than the halo send map's 27 bits name", ctx->nloc);
      ENSURE(ctx, ctx->send_map, (size_t)ns + 8);
      if (width == 5)
         hipLaunchKernelGGL(k_send_map_pack5, dim3(cdiv(ns, 256)), dim3(256), 0, st, ns, ctx->sseg, mg_dirtab(ctx), ctx->hs_cap, ctx->hs_idx.p, ctx->send_map.p,
                            ctx->gp.L[0], ctx->gp.L[1], ctx->gp.L[2], ctx->pos.p, ctx->gid.p, ctx->sendbuf.p);
      else
      {
         hipLaunchKernelGGL(k_send_map, dim3(cdiv(ns, 256)), dim3(256), 0, st, ns, ctx->sseg, ctx->hs_cap, ctx->hs_idx.p, ctx->send_map.p);
         hipLaunchKernelGGL(k_pack_halo, dim3(cdiv(ns, 256)), dim3(256), 0, st, ns, mg_dirtab(ctx), ctx->send_map.p,
                            ctx->gp.L[0], ctx->gp.L[1], ctx->gp.L[2], ctx->pos.p, ctx->gid.p, ctx->sendbuf.p, width);
      }
   }
   return DDCMI_OK;
}
static int mg_phase4_finish(ddcmi_ctx *ctx)
{
   hipStream_t st = ctx->stream;
   int rc;
   OffTab selfo;
   int nself = 0;
   for (int code = 0; code < 27; code++) { selfo.off[code] = nself; if (mg_selfdir(ctx, code)) nself += ctx->hs_cnt[code]; }
   selfo.off[27] = nself;
   int nh = nself + ctx->nrecv;
   ctx->nhalo = nh;
   ctx->nself_images = nself;
   ctx->halo_in_recv = false;      /* (the rebuild places every halo bead in pos[] from the 5-wide records; the 3-wide receive buffer is stale) */
   ctx->hkey_valid = false;
   if (nh > 0)
   {
      if ((rc = ddcmi_bl_reserve_halo(ctx, nh))) return rc;
      ENSURE(ctx, ctx->hkey, (size_t)nh + 1);
      hipLaunchKernelGGL(k_halo_assemble, dim3(cdiv(nh, 256)), dim3(256), 0, st, ctx->gp, nself, ctx->nrecv, selfo, ctx->rseg, mg_dirtab(ctx), ctx->hs_cap, ctx->hs_idx.p,
                         ctx->pos.p, ctx->hrecv5.p, ctx->hsrc_t.p, ctx->hshift_t.p, ctx->hcid.p, ctx->hrank.p, ctx->cell_cnt_h.p, ctx->gid.p, ctx->hkey.p);
      ctx->hkey_valid = true;
      /* (no 5 -> 3 copy of the received records: the tagged halo update of the rebuild reads x y z out of the 5-wide ones) */
   }
   ctx->phase(6, "mg halo assemble launched");
   if ((rc = ddcmi_bl_halo_sort(ctx))) return rc;
   ctx->phase(7, "mg halo sort launched");
   if (ctx->has_charge)
   {
      /* sum of q^2 over this rank's beads: reaches the host with ddcmi_bl_finish's own round trip (pinned h_results) */
      double *d = ctx->d_results + R_GROUP;
      HIPCHK(ctx, hipMemsetAsync(d, 0, sizeof(double), st));
      if (ctx->nloc > 0) hipLaunchKernelGGL(k_sum_q2, dim3(cdiv(ctx->nloc, 256)), dim3(256), 0, st, ctx->nloc, ctx->pos.p, ctx->d_charge_sp.p, d);
      HIPCHK(ctx, hipMemcpyAsync(ctx->h_results + R_GROUP, d, sizeof(double), hipMemcpyDeviceToHost, st));
   }
   if ((rc = ddcmi_bl_finish(ctx))) return rc;
   ctx->self_ele = ctx->has_charge ? -0.5 * ctx->h_results[R_GROUP] * ctx->keR * ctx->crf : 0.0;      /* (: 0 -- charges switched off under an uploaded state) */     /* bioMartini.c:1030-1035 over this rank's local beads */
   ctx->halo_fresh = true;
   return DDCMI_OK;
}

/* one rank, RCCL transport */
int ddcmi_mg_rebuild(ddcmi_ctx *ctx)
{
   if (ctx->group_) SETERR(ctx, DDCMI_EINVAL, "contexts of an in-process group are rebuilt with ddcmi_group_* calls");
   if (!mg_transport(ctx)) SETERR(ctx, DDCMI_EINVAL, "ddcmi_comm_init has not been called");
   int rc;
   /* the per-direction counts never wait on the host between the kernel that counts and the exchange: the all-gather reads the
    * device counters, and ONE synchronisation brings every rank's counts (and capacities: a rank whose segments overflowed makes
    * all ranks repeat the round together) to the host.  A rank on which something failed on the way still joins the round, with
    * its error code in its block, and every rank returns from the rebuild together (mg_counts_round, mg_agree): none is left
    * waiting inside the next exchange for a peer that has gone. */
   ctx->phase(-1, nullptr);
   ctx->mg_rebuilds++;
   for (;;)
   {
      bool over = false;
      int mx = 0;
      const int lrc = mg_phase1_launch(ctx);
      ctx->phase(0, "mg phase1 launched");
      if ((rc = mg_counts_round(ctx, ctx->mig_scnt, ctx->mig_rcnt, &over, &mx, true, lrc))) return rc;
      ctx->phase(1, "mg migration count round");
      if (!over) break;
      if (mx > ctx->mig_cap) ctx->mig_cap = mx + mx / 4 + 64;     /* positions were only wrapped (idempotent): simply redo */
   }
   int lrc = DDCMI_OK;
   {
      int roff[27], acc = 0;
      for (int c = 0; c < 27; c++) { roff[c] = acc; acc += ctx->mig_rcnt[c]; }
      /* (between a count round and its data exchange only an allocation of a few MB can fail: that rank leaves the job, mg_fatal) */
      if (ctx->mig_in.ensure((size_t)acc * 10 + 16)) return mg_fatal(ctx, DDCMI_ENOMEM, "device allocation for the arriving beads failed");
      if ((rc = mg_xchg_data(ctx, ctx->mig_out.p, nullptr, ctx->mig_scnt, ctx->mig_cap, ctx->mig_in.p, roff, ctx->mig_rcnt, 10))) return rc;
   }
   ctx->phase(2, "mg migration exchange");
   lrc = mg_phase2_migrate_in(ctx);
   for (;;)
   {
      bool over = false;
      int mx = 0;
      if (!lrc) lrc = mg_halo_select_launch(ctx);
      ctx->phase(3, "mg phase2+select launched");
      if ((rc = mg_counts_round(ctx, ctx->hs_cnt, ctx->hr_cnt, &over, &mx, false, lrc))) return rc;
      ctx->phase(4, "mg halo count round");
      if (!over) break;
      if (mx > ctx->hs_cap) ctx->hs_cap = mx + mx / 4 + 64;
   }
   if ((lrc = mg_phase3_pack(ctx, 5))) return mg_fatal(ctx, lrc, nullptr);
   if ((rc = mg_xchg_halo(ctx, ctx->sendbuf.p, ctx->hrecv5.p, 5, ctx->stream))) return rc;
   /* the list build, the bonded terms' and the constraint groups' partners (beads that are not numbers, a partner beyond the
    * halo ...): local checks, agreed on before anybody enters the next collective -- the one collective of a rebuild that
    * exists for errors only (one small all-reduce: 2 us per step at a 20-step period) */
   ctx->phase(5, "mg pack+exchange launched");
   lrc = mg_phase4_finish(ctx);
   ctx->phase(15, "mg phase4 rest");
   int pretend = 0;
   if (getenv("DDCMI_DEBUG_HOOKS"))
   {
      /* test hooks: DDCMI_DEBUG_FAIL_REBUILD="<rank>:<n>" fails this rank's local phase of its n-th rebuild;
       * DDCMI_DEBUG_PEER_FAILS_REBUILD="<n>" contributes an error code to the n-th rebuild's agreement as a failed peer would */
       int fr = -1, fn = -1;
       const char *e1 = getenv("DDCMI_DEBUG_FAIL_REBUILD"), *e2 = getenv("DDCMI_DEBUG_PEER_FAILS_REBUILD");
       if (e1 && sscanf(e1, "%d:%d", &fr, &fn) == 2 && fr == ctx->rank && fn == (int)ctx->mg_rebuilds && !lrc) { ctx->err = "injected failure of the rebuild's local phase (DDCMI_DEBUG_FAIL_REBUILD)"; lrc = DDCMI_ENOMEM; }
       if (e2 && atoi(e2) == (int)ctx->mg_rebuilds) pretend = 3;
   }
   /* The first rebuilds of a run can fail on the set-up (a bonded partner or a constraint partner beyond the halo, capacities):
    * the ranks agree on the outcome with one small all-reduce and its host wait (mg_agree), so that a set-up error comes back at
    * once from every rank.  Later rebuilds agree as well -- a bond that stretches beyond the halo in a system going unstable, a
    * capacity or an allocation are found only locally and can appear mid-run -- but without the host wait (mg_agree_async: it
    * cost a healthy run ~4 us per step at a 20-step period on a 500 k-bead rank): the result is looked at in front of the
    * next host wait. */
   if (ctx->mg_rebuilds <= 2 || ctx->hcomm) { if ((rc = mg_agree(ctx, lrc))) return rc; }      /* (a count every rank keeps alike, whatever failed where) */
   else if ((rc = mg_agree_async(ctx, lrc, pretend))) return rc;
   ctx->phase(16, "mg agree");
   /* which molecules have atoms on several ranks, and where is their anchor? (one all-reduce of 4 doubles per multi-bead molecule) */
   if (ctx->mol_gid && ctx->nmol_multi > 0 && (rc = mg_allreduce_device(ctx, ctx->mol_info.p, 4 * (size_t)ctx->nmol_multi))) return rc;
   return ddcmi_mol_split_finish(ctx);
}

/* per-step halo refresh: pack x y z of the send lists, exchange, (k_halo_update places them) */
static int mg_pack3(ddcmi_ctx *ctx, hipStream_t st)
{
   if (ctx->nsend > 0)
   {
      hipLaunchKernelGGL(k_pack_halo, dim3(cdiv(ctx->nsend, 256)), dim3(256), 0, st, ctx->nsend, mg_dirtab(ctx), ctx->send_map.p,
                         ctx->gp.L[0], ctx->gp.L[1], ctx->gp.L[2], ctx->pos.p, ctx->gid.p, ctx->sendbuf.p, 3);
   }
   return DDCMI_OK;
}
/* the pack of the next halo refresh as a job of the fused step's reduction launch (k_reduce_jobs_images) */
static bool ddcmi_mg_pack_job(ddcmi_ctx *ctx, PackJob *pk)
{
   if (ctx->group_ || !ctx->list_valid || ctx->nsend <= 0 || !ctx->send_map.p || !ctx->sendbuf.p) return false;
   const DirTab dt = mg_dirtab(ctx);
   pk->nsend = ctx->nsend; pk->send_map = ctx->send_map.p;
   for (int c = 0; c < 27; c++) for (int a = 0; a < 3; a++) pk->shift[c][a] = dt.shift[c][a];
   pk->L0 = ctx->gp.L[0]; pk->L1 = ctx->gp.L[1]; pk->L2 = ctx->gp.L[2];
   pk->pos = ctx->pos.p; pk->out = ctx->sendbuf.p;
   return true;
}
int ddcmi_mg_refresh_halo(ddcmi_ctx *ctx, hipStream_t st)
{
   RoctxRange rng_upd("UPDATE");      /* ddcUpdate.c:40-85 */
   if (ctx->group_) SETERR(ctx, DDCMI_EINVAL, "in-process group: halos are refreshed by ddcmi_group_step_nglf / ddcmi_group_eval_forces");
   int rc;
   const bool packed = ctx->pack_fresh;      /* (the fused step before this one left the messages packed) */
   ctx->pack_fresh = false;
   if (!packed && (rc = mg_pack3(ctx, st))) return rc;
   if ((rc = mg_xchg_halo(ctx, ctx->sendbuf.p, ctx->hrecv3.p, 3, st))) return rc;
   ctx->halo_fresh = true;
   return DDCMI_OK;
}

/* velocity halo (constraint groups named by gid): the velocities of the beads on the halo send lists, same messages
 * and layout as the positions; received beads land in the velocity slots behind the owned ones */
__global__ void k_pack_vel(int nsend, const unsigned *__restrict__ send_map, const double *vx, const double *vy, const double *vz, double *out)
{
   int k = blockIdx.x * blockDim.x + threadIdx.x;
   if (k >= nsend) return;
   const int i = (int)(send_map[k] & 0x7ffffffu);
   out[3 * k] = vx[i]; out[3 * k + 1] = vy[i]; out[3 * k + 2] = vz[i];
}
__global__ void k_unpack_vel(int nloc, int nhalo, const int *halo_src, const double *vrecv, double *vx, double *vy, double *vz)
{
   int h = blockIdx.x * blockDim.x + threadIdx.x;
   if (h >= nhalo) return;
   const int s = halo_src[h];
   if (s >= 0) { vx[nloc + h] = vx[s]; vy[nloc + h] = vy[s]; vz[nloc + h] = vz[s]; }      /* periodic self-image */
   else { const int k = -1 - s; vx[nloc + h] = vrecv[3 * k]; vy[nloc + h] = vrecv[3 * k + 1]; vz[nloc + h] = vrecv[3 * k + 2]; }
}
static int mg_pack_vel(ddcmi_ctx *ctx, hipStream_t st)
{
   if (ctx->nsend > 0)
      hipLaunchKernelGGL(k_pack_vel, dim3(cdiv(ctx->nsend, 256)), dim3(256), 0, st, ctx->nsend, ctx->send_map.p, ctx->vx.p, ctx->vy.p, ctx->vz.p, ctx->sendbuf.p);
   return DDCMI_OK;
}
static int mg_unpack_vel(ddcmi_ctx *ctx, hipStream_t st)
{
   if (ctx->nhalo > 0)
      hipLaunchKernelGGL(k_unpack_vel, dim3(cdiv(ctx->nhalo, 256)), dim3(256), 0, st, ctx->nloc, ctx->nhalo, ctx->halo_src.p, ctx->hrecv3.p, ctx->vx.p, ctx->vy.p, ctx->vz.p);
   return DDCMI_OK;
}
int ddcmi_mg_refresh_vel(ddcmi_ctx *ctx)
{
   if (ctx->group_) SETERR(ctx, DDCMI_EINVAL, "in-process group: velocities are exchanged by ddcmi_group_step_nglf");
   int rc;
   hipStream_t st = ctx->stream;
   if (mg_transport(ctx))
   {
      if ((rc = mg_pack_vel(ctx, st))) return rc;
      if ((rc = mg_xchg_halo(ctx, ctx->sendbuf.p, ctx->hrecv3.p, 3, st))) return rc;
   }
   return mg_unpack_vel(ctx, st);
}
/* sum of n doubles on the device over the ranks, in place */
static int mg_allreduce_device(ddcmi_ctx *ctx, double *d, size_t n)
{
   if (n == 0 || (ctx->nranks == 1 && !ctx->loopback) || !mg_transport(ctx)) return DDCMI_OK;
   if (ctx->hcomm)
   {
      ctx->hstage_s.resize(n + 1);
      HIPCHK(ctx, hipMemcpyAsync(ctx->hstage_s.data(), d, n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
      HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
      HOSTCHK(ctx, ddcmi_rdzv_allreduce_f64(ctx->hcomm, ctx->hstage_s.data(), (int)n, 0));
      HIPCHK(ctx, hipMemcpyAsync(d, ctx->hstage_s.data(), n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
      HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
      return DDCMI_OK;
   }
   NCCLCHK2(ctx, ncclAllReduce(d, d, n, ncclDouble, ncclSum, (ncclComm_t)ctx->comm, ctx->stream));
   return DDCMI_OK;
}

/* ---- in-process group (tests on one GPU) ----------------------------------- */
/* the group's all-reduce: element-wise sum of one device array per domain, written back to all of them */
static int group_sum_device(ddcmi_group *g, const std::vector<double *> &bufs, size_t n)
{
   if (n == 0) return DDCMI_OK;
   std::vector<double> sum(n, 0.0), tmp(n);
   for (size_t r = 0; r < g->ranks.size(); r++)
   {
      ddcmi_ctx *c = g->ranks[r];
      HIPCHK(c, hipMemcpyAsync(tmp.data(), bufs[r], n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
      HIPCHK(c, hipStreamSynchronize(c->stream));
      for (size_t k = 0; k < n; k++) sum[k] += tmp[k];
   }
   for (size_t r = 0; r < g->ranks.size(); r++)
   {
      ddcmi_ctx *c = g->ranks[r];
      HIPCHK(c, hipMemcpyAsync(bufs[r], sum.data(), n * sizeof(double), hipMemcpyHostToDevice, c->stream));
      HIPCHK(c, hipStreamSynchronize(c->stream));
   }
   return DDCMI_OK;
}
static int group_refresh_vel(ddcmi_group *g)
{
   int rc;
   for (ddcmi_ctx *c : g->ranks) if ((rc = mg_pack_vel(c, c->stream))) return rc;
   if ((rc = mg_xchg_data_local(g, 2))) return rc;
   for (ddcmi_ctx *c : g->ranks) if ((rc = mg_unpack_vel(c, c->stream))) return rc;
   return DDCMI_OK;
}
static int group_rebuild(ddcmi_group *g)
{
   int rc;
   for (ddcmi_ctx *c : g->ranks) { (void)hipSetDevice(c->device); if ((rc = mg_phase1_migrate_out(c))) return rc; }
   for (ddcmi_ctx *B : g->ranks)
   {
      int acc = 0;
      for (int code = 0; code < 27; code++)
      {
         B->mig_rcnt[code] = mg_remote(B, mg_opp(code)) ? g->ranks[B->dir_dest[mg_opp(code)]]->mig_scnt[code] : 0;
         acc += B->mig_rcnt[code];
      }
      if (B->mig_in.ensure((size_t)acc * 10 + 16)) SETERR(B, DDCMI_ENOMEM, "migration buffer");
   }
   if ((rc = mg_xchg_data_local(g, 0))) return rc;
   for (ddcmi_ctx *c : g->ranks) if ((rc = mg_phase2_migrate_in(c))) return rc;
   for (ddcmi_ctx *c : g->ranks) if ((rc = mg_halo_select(c))) return rc;
   for (ddcmi_ctx *B : g->ranks)
      for (int code = 0; code < 27; code++)
         B->hr_cnt[code] = mg_remote(B, mg_opp(code)) ? g->ranks[B->dir_dest[mg_opp(code)]]->hs_cnt[code] : 0;
   for (ddcmi_ctx *c : g->ranks) if ((rc = mg_phase3_pack(c, 5))) return rc;
   if ((rc = mg_xchg_data_local(g, 1))) return rc;
   for (ddcmi_ctx *c : g->ranks) if ((rc = mg_phase4_finish(c))) return rc;
   {
      ddcmi_ctx *c0 = g->ranks[0];
      if (c0->mol_gid && c0->nmol_multi > 0)
      {
         std::vector<double *> bufs;
         for (ddcmi_ctx *c : g->ranks) bufs.push_back(c->mol_info.p);
         if ((rc = group_sum_device(g, bufs, 4 * (size_t)c0->nmol_multi))) return rc;
      }
      for (ddcmi_ctx *c : g->ranks) if ((rc = ddcmi_mol_split_finish(c))) return rc;
   }
   /* (the rebuild's tagged halo update placed every image and halo bead from these very positions: the force evaluation that follows
    * needs no update -- and has no 3-wide records to take one from: only the steps' exchanges fill them) */
   for (ddcmi_ctx *c : g->ranks) c->images_fresh = true;
   return DDCMI_OK;
}
static int group_refresh(ddcmi_group *g)
{
   int rc;
   for (ddcmi_ctx *c : g->ranks) if ((rc = mg_pack3(c, c->stream))) return rc;
   if ((rc = mg_xchg_data_local(g, 2))) return rc;
   for (ddcmi_ctx *c : g->ranks) c->halo_fresh = true;
   return DDCMI_OK;
}

extern "C" int ddcmi_group_create(ddcmi_ctx **ctxs, int n, int px, int py, int pz)
{
   if (!ctxs || n < 1 || px * py * pz != n) return DDCMI_EINVAL;
   ddcmi_group *g = new ddcmi_group();
   for (int r = 0; r < n; r++)
   {
      if (!ctxs[r] || ctxs[r]->group_ || ctxs[r]->comm) { delete g; return DDCMI_EINVAL; }
      g->ranks.push_back(ctxs[r]);
   }
   for (int r = 0; r < n; r++) { mg_set_topology(ctxs[r], r, n, px, py, pz); ctxs[r]->group_ = g; }
   for (int r = 0; r < n; r++) { int rc = mg_check_one_domain_features(ctxs[r]); if (rc) return rc; }
   return DDCMI_OK;
}
extern "C" int ddcmi_group_destroy(ddcmi_ctx **ctxs, int n)
{
   if (!ctxs || n < 1 || !ctxs[0] || !ctxs[0]->group_) return DDCMI_EINVAL;
   ddcmi_group *g = ctxs[0]->group_;
   for (ddcmi_ctx *c : g->ranks) { c->group_ = nullptr; c->nranks = 1; c->rank = 0; }
   delete g;
   return DDCMI_OK;
}
extern "C" int ddcmi_group_eval_forces(ddcmi_ctx **ctxs, int n)
{
   if (!ctxs || n < 1 || !ctxs[0] || !ctxs[0]->group_) return DDCMI_EINVAL;
   ddcmi_group *g = ctxs[0]->group_;
   int rc;
   /* new species or nonbonded parameters under an uploaded state: class tables, tags and with them the lists first (as ddcmi_eval_forces does for one context;
    * tools/fuzz_sequence.py's bricks: a group evaluated with the tables of the old parameters) */
   for (ddcmi_ctx *c : g->ranks) if (c->tables_dirty && (rc = nb_tables(c))) return rc;
   bool valid = true;
   for (ddcmi_ctx *c : g->ranks) valid = valid && c->list_valid;
   if (!valid) { if ((rc = group_rebuild(g))) return rc; }
   else if ((rc = group_refresh(g))) return rc;
   for (ddcmi_ctx *c : g->ranks) if ((rc = launch_forces(c))) return rc;
   for (ddcmi_ctx *c : g->ranks) HIPCHK(c, hipStreamSynchronize(c->stream));
   return DDCMI_OK;
}
extern "C" int ddcmi_group_step_nglf(ddcmi_ctx **ctxs, int n, double dt, int nsteps)
{
   if (!ctxs || n < 1 || !ctxs[0] || !ctxs[0]->group_) return DDCMI_EINVAL;
   ddcmi_group *g = ctxs[0]->group_;
   int rc;
   ARGCHK(ctxs[0], nsteps < 0 || !std::isfinite(dt), "ddcmi_group_step_nglf: %d steps of dt = %g", nsteps, dt);
   for (ddcmi_ctx *c : g->ranks)
      if (!c->forces_valid) SETERR(ctxs[0], DDCMI_EINVAL, "ddcmi_group_step_nglf needs forces: call ddcmi_group_eval_forces first (firstEnergyCall, masters.c:579)");
   for (ddcmi_ctx *c : g->ranks) if ((rc = mg_check_one_domain_features(c))) return rc;
   for (int s = 0; s < nsteps; s++)
   {
      bool rebuild = false;
      ddcmi_ctx *c0 = g->ranks[0];
      const bool cons = c0->ncgroup > 0, baro = c0->baro_beta > 0.0;
      std::vector<bool> had_drift;
      for (ddcmi_ctx *c : g->ranks) { had_drift.push_back(c->drift_done); if ((rc = step_pre_a(c))) return rc; }
      if (baro && !c0->drift_done)
      {
         /* energyInfo.c allreduce() of the barostat's inputs */
         double sum[7] = {0, 0, 0, 0, 0, 0, 0};
         for (ddcmi_ctx *c : g->ranks) for (int k = 0; k < 7; k++) sum[k] += c->baro_sums[k];
         for (ddcmi_ctx *c : g->ranks) for (int k = 0; k < 7; k++) c->baro_sums[k] = sum[k];
         if (c0->nsplit > 0)
         {
            std::vector<double *> bufs;
            for (ddcmi_ctx *c : g->ranks) bufs.push_back(c->mol_red.p);
            if ((rc = group_sum_device(g, bufs, 6 * (size_t)c0->nsplit))) return rc;
         }
      }
      for (ddcmi_ctx *c : g->ranks) if ((rc = step_pre_b(c, dt))) return rc;
      if (cons && !had_drift[0] && (rc = group_refresh_vel(g))) return rc;
      for (ddcmi_ctx *c : g->ranks)
      {
         if ((rc = step_pre_c(c, dt))) return rc;
         if (!c->list_valid) rebuild = true;
         else if (c->updateRate > 0) { if (c->loop % c->updateRate == 0) rebuild = true; }
         else { int need = 0; if ((rc = ddcmi_displacement_check(c, &need))) return rc; if (need) rebuild = true; }      /* check4updateNeighbor: any domain */
      }
      if (rebuild) { if ((rc = group_rebuild(g))) return rc; }
      else if ((rc = group_refresh(g))) return rc;
      for (ddcmi_ctx *c : g->ranks) if ((rc = step_post(c, dt, s + 1 < nsteps))) return rc;
      if (cons)
      {
         /* BACK kick done by step_post: velocity halo, BACK solve, kinetic terms */
         if ((rc = group_refresh_vel(g))) return rc;
         for (ddcmi_ctx *c : g->ranks) if ((rc = step_post_cons_b(c, dt))) return rc;
      }
   }
   return DDCMI_OK;
}

/* group temperatures of an in-process group: the per-rank sums are added on the host */
extern "C" int ddcmi_group_temperatures_all(ddcmi_ctx **ctxs, int n, double *Tgroup)
{
   if (!ctxs || n < 1 || !ctxs[0] || !ctxs[0]->group_) return DDCMI_EINVAL;
   ddcmi_group *g = ctxs[0]->group_;
   int ng = ctxs[0]->ngroup, rc;
   std::vector<double> sum(2 * (size_t)std::max(ng, 1), 0.0);
   for (ddcmi_ctx *c : g->ranks)
   {
      if ((rc = ddcmi_group_ke_sums(c))) return rc;
      for (int k = 0; k < 2 * ng; k++) sum[k] += c->h_results[R_GROUP + k];
   }
   for (int q = 0; q < ng; q++)
   {
      double T = (sum[2 * q + 1] > 0.0) ? 2.0 * sum[2 * q] / (3.0 * sum[2 * q + 1]) : 0.0;
      for (ddcmi_ctx *c : g->ranks) if (sum[2 * q + 1] > 0.0) c->gT[q] = T;
      if (Tgroup) Tgroup[q] = T;
   }
   return DDCMI_OK;
}

/* current local beads in device order, identified by gid (ddcMD identifies
 * particles by label); r wrapped into the box */
__global__ void k_export_particles(GridParams gp, int n, const double4 *pos, double *x, double *y, double *z, int *species)
{
   int i = blockIdx.x * blockDim.x + threadIdx.x;
   if (i >= n) return;
   double4 p = pos[i];
   if (gp.pbc & 1) { if (p.x > 0.5 * gp.L[0]) p.x -= gp.L[0]; if (p.x < -0.5 * gp.L[0]) p.x += gp.L[0]; }
   if (gp.pbc & 2) { if (p.y > 0.5 * gp.L[1]) p.y -= gp.L[1]; if (p.y < -0.5 * gp.L[1]) p.y += gp.L[1]; }
   if (gp.pbc & 4) { if (p.z > 0.5 * gp.L[2]) p.z -= gp.L[2]; if (p.z < -0.5 * gp.L[2]) p.z += gp.L[2]; }
   x[i] = p.x; y[i] = p.y; z[i] = p.z;
   species[i] = (int)((__double_as_longlong(p.w) >> 16) & 0xffff);
}
extern "C" int ddcmi_download_particles(ddcmi_ctx *ctx, int cap, int *nout, uint64_t *gid, int *species,
                                        double *rx, double *ry, double *rz, double *vx, double *vy, double *vz,
                                        double *fx, double *fy, double *fz)
{
   if (!ctx || !nout) return DDCMI_EINVAL;
   (void)hipSetDevice(ctx->device);
   { int rca = ddcmi_agree_poll(ctx); if (rca) return rca; }
   hipStream_t st = ctx->stream;
   int n = ctx->nloc;
   *nout = n;
   if (cap < n) SETERR(ctx, DDCMI_EINVAL, "ddcmi_download_particles: capacity %d < %d local beads", cap, n);
   if (n == 0) return DDCMI_OK;
   GridParams gp = ctx->gp;
   gp.pbc = ctx->pbc; gp.L[0] = ctx->h[0]; gp.L[1] = ctx->h[4]; gp.L[2] = ctx->h[8];
   ENSURE(ctx, ctx->species2, (size_t)n + 1);
   hipLaunchKernelGGL(k_export_particles, dim3(cdiv(n, 256)), dim3(256), 0, st, gp, n, ctx->pos.p, ctx->vx2.p, ctx->vy2.p, ctx->vz2.p, ctx->species2.p);
   if (rx) HIPCHK(ctx, hipMemcpyAsync(rx, ctx->vx2.p, n * sizeof(double), hipMemcpyDeviceToHost, st));
   if (ry) HIPCHK(ctx, hipMemcpyAsync(ry, ctx->vy2.p, n * sizeof(double), hipMemcpyDeviceToHost, st));
   if (rz) HIPCHK(ctx, hipMemcpyAsync(rz, ctx->vz2.p, n * sizeof(double), hipMemcpyDeviceToHost, st));
   if (species) HIPCHK(ctx, hipMemcpyAsync(species, ctx->species2.p, n * sizeof(int), hipMemcpyDeviceToHost, st));
   if (gid) HIPCHK(ctx, hipMemcpyAsync(gid, ctx->gid.p, n * sizeof(uint64_t), hipMemcpyDeviceToHost, st));
   if (vx) HIPCHK(ctx, hipMemcpyAsync(vx, ctx->vx.p, n * sizeof(double), hipMemcpyDeviceToHost, st));
   if (vy) HIPCHK(ctx, hipMemcpyAsync(vy, ctx->vy.p, n * sizeof(double), hipMemcpyDeviceToHost, st));
   if (vz) HIPCHK(ctx, hipMemcpyAsync(vz, ctx->vz.p, n * sizeof(double), hipMemcpyDeviceToHost, st));
   if (fx) HIPCHK(ctx, hipMemcpyAsync(fx, ctx->fx.p, n * sizeof(double), hipMemcpyDeviceToHost, st));
   if (fy) HIPCHK(ctx, hipMemcpyAsync(fy, ctx->fy.p, n * sizeof(double), hipMemcpyDeviceToHost, st));
   if (fz) HIPCHK(ctx, hipMemcpyAsync(fz, ctx->fz.p, n * sizeof(double), hipMemcpyDeviceToHost, st));
   HIPCHK(ctx, hipStreamSynchronize(st));
   return DDCMI_OK;
}
extern "C" int ddcmi_domain_bounds(const ddcmi_ctx *ctx, double lo[3], double hi[3])
{
   if (!ctx || !ctx->have_box) return DDCMI_EINVAL;
   for (int a = 0; a < 3; a++)
   {
      double L = ctx->h[4 * a], W = L / ctx->pgrid[a];
      lo[a] = -0.5 * L + ctx->pcoord[a] * W;
      hi[a] = lo[a] + W;
   }
   return DDCMI_OK;
}

/* nglfconstraint's barostat and velocity constraints given by caller-order indices are solved per domain: with several
 * domains each rank would scale its own box from its local virial and solve constraint groups without their halo
 * partners -- silently wrong physics.  Checked where the decomposition is set up AND at every step call, whatever the
 * call order.  The gid forms carry the cross-domain sums and the velocity halo. */
static int mg_check_one_domain_features(ddcmi_ctx *ctx)
{
   if (ctx->nranks > 1 && ((ctx->baro_beta > 0.0 && !ctx->mol_gid) || (ctx->ncgroup > 0 && !ctx->cons_gid)))
      SETERR(ctx, DDCMI_EUNSUPPORTED, "%d domains: the barostat and the velocity constraints (NGLFCONSTRAINT) work on a single domain unless the molecules and the "
             "constraint groups are named by gid (ddcmi_set_molecule_lists_gid, ddcmi_set_constraints_gid)", ctx->nranks);
   return DDCMI_OK;
}
/* RCCL bootstrap (the 128-byte id is distributed by the caller: MPI_Bcast in
 * ddcMD, torch.distributed in bench.py) */
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
   if (ctx->group_) SETERR(ctx, DDCMI_EINVAL, "context already belongs to an in-process group");
   if (mg_transport(ctx)) SETERR(ctx, DDCMI_EINVAL, "the context already has a communicator");
   (void)hipSetDevice(ctx->device);
   ncclUniqueId uid;
   memcpy(&uid, id, sizeof(uid));
   ncclComm_t comm;
   NCCLCHK2(ctx, ncclCommInitRank(&comm, nranks, uid, rank));
   ctx->comm = (void *)comm;
   { const char *lb = getenv("DDCMI_RCCL_LOOPBACK"); ctx->loopback = (nranks == 1 && lb && atoi(lb) != 0); }
   mg_set_topology(ctx, rank, nranks, px, py, pz);
   return mg_check_one_domain_features(ctx);
}
extern "C" int ddcmi_comm_init_host(ddcmi_ctx *ctx, ddcmi_rdzv *rdzv, int px, int py, int pz)
{
   if (!ctx || !rdzv) return DDCMI_EINVAL;
   const int rank = ddcmi_rdzv_rank(rdzv), nranks = ddcmi_rdzv_world(rdzv);
   if (px * py * pz != nranks) SETERR(ctx, DDCMI_EINVAL, "process grid %dx%dx%d does not match %d ranks", px, py, pz, nranks);
   if (ctx->group_) SETERR(ctx, DDCMI_EINVAL, "context already belongs to an in-process group");
   if (mg_transport(ctx)) SETERR(ctx, DDCMI_EINVAL, "the context already has a communicator");
   ctx->hcomm = rdzv;
   { const char *lb = getenv("DDCMI_RCCL_LOOPBACK"); ctx->loopback = (nranks == 1 && lb && atoi(lb) != 0); }
   mg_set_topology(ctx, rank, nranks, px, py, pz);
   return mg_check_one_domain_features(ctx);
}
/* sum of n doubles over the ranks, in place on the host */
static int mg_allreduce_host_values(ddcmi_ctx *ctx, double *values, int n)
{
   if ((ctx->nranks == 1 && !ctx->loopback) || !mg_transport(ctx)) return DDCMI_OK;
   if (ctx->hcomm) { HOSTCHK(ctx, ddcmi_rdzv_allreduce_f64(ctx->hcomm, values, n, 0)); return DDCMI_OK; }
   { int rca = ddcmi_agree_poll(ctx); if (rca) return rca; }
   double *d = ctx->d_results + R_GROUP;   /* scratch */
   HIPCHK(ctx, hipMemcpyAsync(d, values, n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
   NCCLCHK2(ctx, ncclAllReduce(d, d, n, ncclDouble, ncclSum, (ncclComm_t)ctx->comm, ctx->stream));
   HIPCHK(ctx, hipMemcpyAsync(values, d, n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
   HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
   return DDCMI_OK;
}
/* energyInfo.c:9-63 allreduce(): sum the ETYPE block across ranks */
extern "C" int ddcmi_comm_allreduce_sum(ddcmi_ctx *ctx, double *values, int n)
{
   if (!ctx || !values || n <= 0 || n > 64) return DDCMI_EINVAL;
   (void)hipSetDevice(ctx->device);
   if (ctx->group_)
   {
      /* in-process group: only meaningful when called once for rank 0 with pre-summed values */
      return DDCMI_OK;
   }
   return mg_allreduce_host_values(ctx, values, n);
}
/* ---- preflight: the first real multi-rank launch fails fast and legibly ------------------------------------
 * ddcUpdate.c:56-85 / energyInfo.c:37 meet their peers for the first time inside the first step; a fabric that does not carry one
 * of the seven links of a brick, a rank on the wrong device or a stale communicator id shows there as a hang.  This runs right
 * behind ddcmi_comm_init, before any state is uploaded: (1) ONE grouped exchange of a known pattern along every direction the brick
 * plan names -- the matching rule, peers and grouping of the per-step halo exchange and of the migration (mg_xchg_data), PF_N doubles
 * per direction, verified element by element on the receiver --, (2) one 24-double sum all-reduce (energyInfo.c allreduce()) and
 * (3) one all-gather of MG_BLK ints per rank (the rebuild's count round), each held against its closed form.  RCCL: everything is
 * queued on the context's stream with an event behind each stage and the host polls under a deadline -- when it passes, the rank
 * says which stage never finished and, for the exchange, from which peer ranks and directions nothing arrived (the receive buffer,
 * pre-filled with a sentinel, is read back on a second stream), aborts its communicator and returns DDCMI_ECOMM.  Host transport:
 * the rendezvous' own per-transfer timeouts name the peer.  The ranks then agree on the outcome (max of the error codes): a rank
 * whose data was wrong reports what it saw, every other rank reports that a peer failed -- all return non-zero from the same call. */
#define PF_N 512
__global__ void k_pf_fill(int n, double *send, double *recv, int rank, int corrupt_code)
{
   const int i = blockIdx.x * blockDim.x + threadIdx.x;
   if (i >= 27 * n) return;
   const int code = i / n, k = i % n;
   send[i] = (double)(rank * 27 + code) * 4096.0 + (double)k + (code == corrupt_code && k == 7 ? 0.5 : 0.0);
   recv[i] = -1.0;
}
static void pf_dir_name(int code, char *b, size_t nb) { snprintf(b, nb, "(%+d,%+d,%+d)", code % 3 - 1, (code / 3) % 3 - 1, code / 9 - 1); }
/* what every rank of a run must have been given alike -- box, cut-offs, tables, species, molecule tables, term counts, groups, the process grid -- as one
 * 64-bit number (FNV-1a over the values' bytes): the preflight's all-gather carries it, and ranks that read different decks say so before the first step
 * instead of exchanging halos of different widths */
static uint64_t mg_param_hash(const ddcmi_ctx *ctx)
{
   uint64_t h = 1469598103934665603ull;
   auto eat = [&](const void *p, size_t n) { const unsigned char *b = (const unsigned char *)p; for (size_t k = 0; k < n; k++) { h ^= b[k]; h *= 1099511628211ull; } };
   auto vd = [&](const std::vector<double> &v) { const size_t n = v.size(); eat(&n, sizeof(n)); if (n) eat(v.data(), n * sizeof(double)); };
   auto vi = [&](const std::vector<int> &v) { const size_t n = v.size(); eat(&n, sizeof(n)); if (n) eat(v.data(), n * sizeof(int)); };
   eat(ctx->h, sizeof(ctx->h)); eat(&ctx->pbc, sizeof(int));
   eat(&ctx->rmax, sizeof(double)); eat(&ctx->deltaR, sizeof(double)); eat(&ctx->updateRate, sizeof(int));
   eat(&ctx->keR, sizeof(double)); eat(&ctx->krf, sizeof(double)); eat(&ctx->crf, sizeof(double));
   eat(&ctx->nlj, sizeof(int)); vd(ctx->sigma); vd(ctx->eps); vd(ctx->shift);
   eat(&ctx->nspecies, sizeof(int)); vd(ctx->mass); vd(ctx->charge); vi(ctx->ljtype); vi(ctx->moltype);
   eat(&ctx->nmoltype, sizeof(int));
   if (ctx->nmoltype > 0) { vi(ctx->mol_nspecies); vi(ctx->bpair_off); vi(ctx->bpairI); vi(ctx->bpairJ); }
   eat(&ctx->nbond, sizeof(int)); eat(&ctx->nangle, sizeof(int)); eat(&ctx->ntors, sizeof(int)); eat(&ctx->nrest, sizeof(int)); eat(&ctx->ncgroup, sizeof(int));
   eat(&ctx->excludePotentialTerm, sizeof(int));
   eat(&ctx->ngroup, sizeof(int)); vi(ctx->gtype); vi(ctx->ginterval); vd(ctx->gTeq); vd(ctx->gtau);
   eat(&ctx->baro_beta, sizeof(double)); eat(&ctx->baro_tau, sizeof(double));
   eat(ctx->pgrid, sizeof(ctx->pgrid));
   return h;
}
extern "C" int ddcmi_comm_preflight(ddcmi_ctx *ctx, double timeout_s, int64_t report[16])
{
   if (!ctx) return DDCMI_EINVAL;
   if (report) memset(report, 0, 16 * sizeof(int64_t));
   if (!mg_transport(ctx)) SETERR(ctx, DDCMI_EINVAL, "ddcmi_comm_preflight: the context has no communicator (ddcmi_comm_init first)");
   (void)hipSetDevice(ctx->device);
   if (timeout_s <= 0.0) timeout_s = 60.0;
   hipStream_t st = ctx->stream;
   const int nr = std::max(ctx->nranks, 1);
   const bool hooks = getenv("DDCMI_DEBUG_HOOKS") != nullptr;
   const int corrupt = (hooks && getenv("DDCMI_DEBUG_PREFLIGHT_CORRUPT")) ? atoi(getenv("DDCMI_DEBUG_PREFLIGHT_CORRUPT")) : -1;      /* tests: this rank spoils its message along that direction */
   const int corrupt_rank = (hooks && getenv("DDCMI_DEBUG_PREFLIGHT_CORRUPT_RANK")) ? atoi(getenv("DDCMI_DEBUG_PREFLIGHT_CORRUPT_RANK")) : 0;
   const bool absent = hooks && getenv("DDCMI_DEBUG_PREFLIGHT_ABSENT") && atoi(getenv("DDCMI_DEBUG_PREFLIGHT_ABSENT")) == ctx->rank;      /* tests (host transport): this rank never joins the exchange */
   struct timespec t0; clock_gettime(CLOCK_MONOTONIC, &t0);
   auto elapsed = [&]() { struct timespec t1; clock_gettime(CLOCK_MONOTONIC, &t1); return (t1.tv_sec - t0.tv_sec) + 1e-9 * (t1.tv_nsec - t0.tv_nsec); };
   dbuf<double> buf;      /* [0, 27 PF_N) send, [27 PF_N, 54 PF_N) receive, then 24 + 24 doubles of the all-reduce */
   dbuf<int> ibuf;        /* MG_BLK ints in, MG_BLK nr ints out */
   int rc = DDCMI_OK;
   std::string msg;
   int soff[27], scnt[27], roff[27], rcnt[27], npeer = 0, peers[27], ndir = 0;
   for (int c = 0; c < 27; c++)
   {
      soff[c] = roff[c] = c * PF_N;
      scnt[c] = mg_remote(ctx, c) ? PF_N : 0;
      rcnt[c] = mg_remote(ctx, mg_opp(c)) ? PF_N : 0;
      if (scnt[c])
      {
         ndir++;
         bool seen = false;
         for (int k = 0; k < npeer; k++) seen |= peers[k] == ctx->dir_dest[c];
         if (!seen) peers[npeer++] = ctx->dir_dest[c];
      }
   }
   if (report) { report[0] = npeer; report[1] = ndir; report[2] = (int64_t)PF_N * sizeof(double); for (int k = 0; k < npeer && k < 8; k++) report[8 + k] = peers[k]; }
   std::vector<double> h(27 * PF_N + 48);
   std::vector<int> hi(MG_BLK * (size_t)nr + MG_BLK);
   hipEvent_t ev[3] = {nullptr, nullptr, nullptr};
   hipStream_t side = nullptr;
   int stage = 0;      /* stages verified */
   do
   {
      if (buf.ensure(54 * PF_N + 64) || ibuf.ensure(MG_BLK * (size_t)(nr + 1) + 64)) { rc = DDCMI_ENOMEM; msg = "preflight buffers"; break; }
      double *snd = buf.p, *rcv = buf.p + 27 * PF_N, *ar = buf.p + 54 * PF_N;
      hipLaunchKernelGGL(k_pf_fill, dim3(cdiv(27 * PF_N, 256)), dim3(256), 0, st, PF_N, snd, rcv, ctx->rank, ctx->rank == corrupt_rank ? corrupt : -1);
      for (int k = 0; k < 24; k++) h[k] = (double)(ctx->rank + 1) * (double)(k + 1);
      for (int k = 0; k < MG_BLK; k++) hi[k] = ctx->rank * 1000 + k;
      { const uint64_t ph = mg_param_hash(ctx); hi[MG_BLK - 3] = (int)(unsigned)(ph & 0xffffffffull); hi[MG_BLK - 2] = (int)(unsigned)(ph >> 32); }      /* (words 29, 30: the parameters' hash instead of the pattern) */
      if (hipMemcpyAsync(ar, h.data(), 24 * sizeof(double), hipMemcpyHostToDevice, st) != hipSuccess ||
          hipMemcpyAsync(ibuf.p, hi.data(), MG_BLK * sizeof(int), hipMemcpyHostToDevice, st) != hipSuccess) { rc = DDCMI_ENODEVICE; msg = "preflight upload"; break; }
      if (ctx->hcomm)
      {
         /* host transport: blocking transfers with the rendezvous' own deadlines and messages */
         if (!absent && (rc = mg_xchg_data(ctx, snd, soff, scnt, 0, rcv, roff, rcnt, 1)) != DDCMI_OK) { msg = "preflight: the grouped exchange with the brick's peers failed: " + ctx->err; break; }
         if (absent)
         {
            /* a rank that is stuck: its peers run into the rendezvous' deadline and name it; then it leaves for good */
            struct timespec ts = {(time_t)timeout_s + 1, 0}; nanosleep(&ts, nullptr);
            ctx->err = "preflight: this rank stayed away from the exchange (DDCMI_DEBUG_PREFLIGHT_ABSENT)";
            buf.release(); ibuf.release();
            return mg_fatal(ctx, DDCMI_ECOMM, nullptr);
         }
         if (hipMemcpyAsync(h.data(), ar, 24 * sizeof(double), hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) { rc = DDCMI_ENODEVICE; msg = "preflight download"; break; }
         if (ddcmi_rdzv_allreduce_f64(ctx->hcomm, h.data(), 24, 0) != DDCMI_OK) { rc = DDCMI_ECOMM; msg = std::string("preflight: the 24-double all-reduce failed: ") + ddcmi_rdzv_last_error(ctx->hcomm); break; }
         memcpy(h.data() + 27 * PF_N, h.data(), 24 * sizeof(double));
         if (ddcmi_rdzv_allgather(ctx->hcomm, hi.data(), hi.data() + MG_BLK, MG_BLK * sizeof(int)) != DDCMI_OK) { rc = DDCMI_ECOMM; msg = std::string("preflight: the count all-gather failed: ") + ddcmi_rdzv_last_error(ctx->hcomm); break; }
         if (hipMemcpyAsync(h.data(), rcv, 27 * PF_N * sizeof(double), hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) { rc = DDCMI_ENODEVICE; msg = "preflight download"; break; }
      }
      else
      {
         ncclComm_t comm = (ncclComm_t)ctx->comm;
         bool evok = true;
         for (int k = 0; k < 3; k++) evok &= hipEventCreateWithFlags(&ev[k], hipEventDisableTiming) == hipSuccess;
         if (!evok || hipStreamCreateWithFlags(&side, hipStreamNonBlocking) != hipSuccess) { rc = DDCMI_ENODEVICE; msg = "preflight events"; break; }
         if ((rc = mg_xchg_data(ctx, snd, soff, scnt, 0, rcv, roff, rcnt, 1)) != DDCMI_OK) { msg = "preflight: queueing the grouped exchange failed: " + ctx->err; break; }
         (void)hipEventRecord(ev[0], st);
         if (ncclAllReduce(ar, ar + 24, 24, ncclDouble, ncclSum, comm, st) != ncclSuccess) { rc = DDCMI_ECOMM; msg = "preflight: ncclAllReduce could not be queued"; break; }
         (void)hipEventRecord(ev[1], st);
         if (ncclAllGather(ibuf.p, ibuf.p + MG_BLK, MG_BLK, ncclInt, comm, st) != ncclSuccess) { rc = DDCMI_ECOMM; msg = "preflight: ncclAllGather could not be queued"; break; }
         (void)hipEventRecord(ev[2], st);
         bool done = false;
         while (!done)
         {
            const hipError_t q = hipEventQuery(ev[2]);
            if (q == hipSuccess) { done = true; break; }
            if (q != hipErrorNotReady) { rc = DDCMI_ENODEVICE; msg = std::string("preflight: the stream failed: ") + hipGetErrorString(q); break; }
            if (elapsed() > timeout_s) break;
            struct timespec ts = {0, 200000}; nanosleep(&ts, nullptr);
         }
         if (rc) break;
         if (!done)
         {
            /* which stage never finished; for the exchange, whose data never came (the receive buffer through a second stream: the first one is stuck) */
            const int reached = hipEventQuery(ev[0]) != hipSuccess ? 0 : hipEventQuery(ev[1]) != hipSuccess ? 1 : 2;
            char b[1024];
            int o = snprintf(b, sizeof(b), "preflight: rank %d of %d: no completion within %.0f s, stuck in %s", ctx->rank, nr, timeout_s,
                             reached == 0 ? "the grouped send/recv with the brick's peers" : reached == 1 ? "the 24-double all-reduce" : "the count all-gather");
            if (reached == 0 && hipMemcpyAsync(h.data(), rcv, 27 * PF_N * sizeof(double), hipMemcpyDeviceToHost, side) == hipSuccess && hipStreamSynchronize(side) == hipSuccess)
            {
               o += snprintf(b + o, sizeof(b) - o, "; nothing arrived from");
               for (int c = 0; c < 27 && o < (int)sizeof(b) - 64; c++)
                  if (rcnt[c] && h[(size_t)c * PF_N + PF_N - 1] == -1.0)
                  {
                     char d[32]; pf_dir_name(mg_opp(c), d, sizeof(d));
                     o += snprintf(b + o, sizeof(b) - o, " rank %d (my direction %s)", ctx->dir_dest[mg_opp(c)], d);
                     if (report && report[3] == 0) { report[3] = 1; report[4] = ctx->dir_dest[mg_opp(c)]; report[5] = mg_opp(c); }
                  }
            }
            msg = b;
            rc = mg_fatal(ctx, DDCMI_ECOMM, nullptr);      /* (releases the kernels that wait for the peer) */
            if (report) { report[6] = reached; report[7] = (int64_t)(elapsed() * 1e6); }
            /* no agreement round: the fabric has just shown that it does not answer */
            for (int k = 0; k < 3; k++) if (ev[k]) (void)hipEventDestroy(ev[k]);
            if (side) (void)hipStreamDestroy(side);
            ctx->err = msg;
            return rc;
         }
         if (hipMemcpyAsync(h.data(), rcv, 27 * PF_N * sizeof(double), hipMemcpyDeviceToHost, st) != hipSuccess ||
             hipMemcpyAsync(h.data() + 27 * PF_N, ar + 24, 24 * sizeof(double), hipMemcpyDeviceToHost, st) != hipSuccess ||
             hipMemcpyAsync(hi.data() + MG_BLK, ibuf.p + MG_BLK, MG_BLK * (size_t)nr * sizeof(int), hipMemcpyDeviceToHost, st) != hipSuccess ||
             hipStreamSynchronize(st) != hipSuccess) { rc = DDCMI_ENODEVICE; msg = "preflight download"; break; }
      }
      /* (1) every direction's message, element by element */
      for (int c = 0; c < 27 && !rc; c++)
      {
         if (!rcnt[c]) continue;
         const int src = ctx->dir_dest[mg_opp(c)];
         for (int k = 0; k < PF_N; k++)
         {
            const double want = (double)(src * 27 + c) * 4096.0 + (double)k, got = h[(size_t)c * PF_N + k];
            if (got != want)
            {
               char d[32], b[512]; pf_dir_name(mg_opp(c), d, sizeof(d));
               snprintf(b, sizeof(b), "preflight: rank %d of %d: the message from rank %d (my direction %s, its direction code %d) is wrong at element %d of %d: got %.17g, expected %.17g",
                        ctx->rank, nr, src, d, c, k, PF_N, got, want);
               msg = b; rc = DDCMI_ECOMM;
               if (report) { report[3] = 1; report[4] = src; report[5] = mg_opp(c); }
               break;
            }
         }
      }
      if (rc) break;
      stage = 1;
      /* (2) sum over ranks of (r + 1)(k + 1) */
      for (int k = 0; k < 24; k++)
      {
         const double want = 0.5 * (double)nr * (double)(nr + 1) * (double)(k + 1), got = h[27 * PF_N + k];
         if (got != want)
         {
            char b[256]; snprintf(b, sizeof(b), "preflight: rank %d of %d: the 24-double all-reduce gave %.17g at element %d, expected %.17g", ctx->rank, nr, got, k, want);
            msg = b; rc = DDCMI_ECOMM; break;
         }
      }
      if (rc) break;
      stage = 2;
      /* (3) every rank's block of the all-gather */
      for (int r = 0; r < nr && !rc; r++)
         for (int k = 0; k < MG_BLK; k++)
            if (k != MG_BLK - 3 && k != MG_BLK - 2 && hi[MG_BLK + (size_t)MG_BLK * r + k] != r * 1000 + k)
            {
               char b[256]; snprintf(b, sizeof(b), "preflight: rank %d of %d: the all-gather holds %d at word %d of rank %d's block, expected %d", ctx->rank, nr, hi[MG_BLK + (size_t)MG_BLK * r + k], k, r, r * 1000 + k);
               msg = b; rc = DDCMI_ECOMM; break;
            }
      if (rc) break;
      stage = 3;
      /* ... and what the block carried: do the ranks run the same system? */
      for (int r = 0; r < nr && !rc; r++)
         if (hi[MG_BLK + (size_t)MG_BLK * r + MG_BLK - 3] != hi[MG_BLK - 3] || hi[MG_BLK + (size_t)MG_BLK * r + MG_BLK - 2] != hi[MG_BLK - 2])
         {
            char b[384];
            snprintf(b, sizeof(b), "preflight: rank %d of %d: rank %d was given other parameters than this rank (box, cut-offs, neighbour settings, LJ table, species, molecule tables, "
                     "term counts, groups, barostat or process grid differ): every rank of a run must be set up from the same deck", ctx->rank, nr, r);
            msg = b; rc = DDCMI_EINVAL;
            if (report) { report[3] = 1; report[4] = r; report[5] = -1; }
         }
   } while (0);
   for (int k = 0; k < 3; k++) if (ev[k]) (void)hipEventDestroy(ev[k]);
   if (side) (void)hipStreamDestroy(side);
   buf.release(); ibuf.release();
   if (report) { report[6] = stage; report[7] = (int64_t)(elapsed() * 1e6); }
   if (rc) ctx->err = msg;
   /* every rank leaves with the same verdict (a transport error above has usually ended the agreement's transport too: then its own error stands) */
   if (rc == DDCMI_ECOMM && ctx->hcomm && msg.find("failed:") != std::string::npos) return rc;
   const int arc = mg_agree(ctx, rc);
   if (rc) { ctx->err = msg; return rc; }
   if (arc) { ctx->err = "preflight: another rank's check of the communicator failed (its own message says which peer and direction): " + ctx->err; return arc; }
   return DDCMI_OK;
}
void ddcmi_comm_destroy(ddcmi_ctx *ctx)
{
   if (ctx->comm) { (void)ncclCommDestroy((ncclComm_t)ctx->comm); ctx->comm = nullptr; }
   ctx->hcomm = nullptr;      /* owned by the caller */
}
