/* ddcmi_step.inl -- host side of a force evaluation and of an NGLF step: launch_forces, step_pre / step_post, ddcmi_step_nglf.
 * Part of the ONE translation unit ddcmi.hip (kernels, templates and the static helpers they share), included there in this order. */
/* ------------------------------------------------------------------------- */
/* defer_reduce: the caller (a time step) folds the nonbonded reduction and the final
 * energies into the launch that reduces the kinetic terms */
/* LDS of one k_nonbond workgroup.  Fixed layout (neighbourhoods of up to NB_ZOFF/16 beads: every Martini system): {x,y} [capl] at
 * address 0, z [capl] at NB_ZOFF -- and whatever else fits goes into the GAP between them, which used to be wasted (4 KB beside a
 * water neighbourhood, 9 KB beside the bilayer's): the fused step's rows of kinetic sums at its top, the pair table at its bottom.
 * The pair table is the direct one (32 bytes per class pair) or, where that would cost the second workgroup per CU (or fit no CU at
 * all), the two-level one (one byte per class pair + the distinct entries).  Types and shifted-copy flags that the list entries do
 * not carry ride in the staged z (k_nonbond): no per-bead byte arrays. */
static NbLds nb_lds_layout(const ddcmi_ctx *ctx, bool fused)
{
   const size_t capl = (size_t)ctx->stage_cap + 2;
   const size_t rows = ((NB_THREADS / 64) * 8 + 2) * sizeof(double);      /* the waves' rows of kinetic sums + (lean step) the displacement bound handed from wave 0 to the others */
   const size_t npair = (size_t)ctx->nnb * ctx->nnb;
   const size_t direct = npair * sizeof(double4), level = (size_t)ctx->nlvl * sizeof(double4) + ((npair + 15) & ~(size_t)15);
   auto lay = [&](size_t table, bool lvl)
   {
      NbLds l;
      l.lvl = lvl; l.zfix = capl * 16 <= NB_ZOFF;
      /* (while it stages, the kernel keeps its cell tables and slot -> cell map where the positions will lie -- 2 (NRC + 8) + 16 ints and
       * 2 bytes per staged bead from address 0: the table, loaded before the staging, must start behind them) */
      const size_t scratch = ((2 * (NRC + 8) + 16) * sizeof(int) + ((NRC + 15) & ~15) + 2 * capl + 15) & ~(size_t)15;      /* cell tables, image cells' shift codes, the slot -> cell map */
      size_t gap_lo = std::max(capl * 16, scratch), gap_hi = l.zfix ? NB_ZOFF : 0, end = l.zfix ? NB_ZOFF + capl * 8 : capl * 24;
      l.ke_off = 0;
      if (fused)
      {
         if (l.zfix && gap_lo + rows <= gap_hi) { gap_hi -= rows; l.ke_off = (int)gap_hi; }
         else { end = (end + 7) & ~(size_t)7; l.ke_off = (int)end; end += rows; }
      }
      if (l.zfix && gap_lo + table <= gap_hi) l.tab_off = (int)gap_lo;
      else { end = (end + 15) & ~(size_t)15; l.tab_off = (int)end; end += table; }
      l.total = end;
      l.wgs = end > 160 * 1024 ? 0 : (int)std::min<size_t>(2, (160 * 1024) / end);
      return l;
   };
   const NbLds d = lay(direct, false);
   if (ctx->nlvl <= 0) return d;
   const NbLds v = lay(level, true);
   return (ctx->force_lvl || v.wgs > d.wgs) ? v : d;
}
/* may this context run lean steps (ddcmi_ctx::lean_pending)?  A single domain whose step is the fused pair kernel (+ the bonded kernels in front of it) */
static bool lean_capable(const ddcmi_ctx *ctx)
{
   if (ctx->no_lean || ctx->group_ || ctx->updateRate <= 0 || ctx->nloc <= 0) return false;      /* (a decomposed rank: where its halo is staged from the receive buffer, launch_forces) */
   if (ctx->nrest != 0 || ctx->ncgroup > 0 || ctx->baro_beta > 0.0 || (ctx->excludePotentialTerm & 128) != 0) return false;
   for (int g = 0; g < ctx->ngroup; g++) if (ctx->gtype[g] != DDCMI_FREE && ctx->gtype[g] != DDCMI_BERENDSEN) return false;      /* (Berendsen: host scalars from the temperature last published) */
   return true;
}
/* a single domain's periodic images are staged by the pair kernel from their owners (NbTileArgs::self_img): nothing reads the image records
 * between two rebuilds, and nothing refreshes them -- where the step can be lean; elsewhere the reduction launch of every step refreshes
 * them on the side and the staging keeps its shorter path (the bilayer: 4 us per pair kernel).  DDCMI_NO_SELF_IMAGES=1: never. */
static bool self_images(const ddcmi_ctx *ctx)
{
   return !ctx->no_self_img && ctx->nranks == 1 && !ctx->loopback && lean_capable(ctx) && ctx->nhalo > 0 && ctx->stage_cap + 2 < 4096;
}
static int launch_forces(ddcmi_ctx *ctx, bool defer_reduce = false, FuseArgs *fuse = nullptr /* in: the integrator's pass rides in the pair kernel; out: ->dt = 0 if this launch could not take it */,
                         bool *lean = nullptr /* in: the caller could run this step lean (ddcmi_ctx::lean_pending); out: this launch did */)
{
   RoctxRange rng_force("DDCENERGY P_FORCE");      /* ddcenergy.c:160-238 */
   hipStream_t st = ctx->stream;
   int n = ctx->nloc, nh = ctx->nhalo;
   /* Decomposed runs, between rebuilds: the halo exchange (pack, one RCCL message per peer,
    * unpack) runs on a second stream while this stream computes the tiles whose
    * neighbourhoods hold owned beads only; the other tiles wait for it. */
   bool halo_pending = false;
   /* the received beads' displacement since the rebuild (NbTileArgs::hdisp): measured by the halo update of a decomposed run whose pair
    * kernel may end its rows early; the word of this step's parity is the one this step's pair kernel reads */
   /* direct halo staging (ddcmi_ctx::halo_in_recv): the pair kernel takes the received beads out of the exchange's receive buffer, no
    * update launch -- for halos of received beads only, needed by the pair kernel only */
   /* (round 6: bonded terms are no obstacle any more -- their kernels take a received partner out of the receive buffer too, k_bonded_gather's bead();
    *  restraints, constraint groups and the barostat still want every halo bead in pos[]) */
   const bool direct = !ctx->no_direct_halo && (ctx->nranks > 1 || ctx->loopback) && (ctx->comm || ctx->hcomm) && !ctx->group_ && !ctx->halo_overlap && ctx->nself_images == 0 &&
                       ctx->nrest == 0 && ctx->ncgroup == 0 && !(ctx->baro_beta > 0.0) && ctx->stage_cap + 2 < 4096 && (ctx->excludePotentialTerm & 128) == 0 && ctx->updateRate > 0;
   const bool hdisp_on = !direct && ctx->shell_skip && nh > 0 && (ctx->nranks > 1 || ctx->loopback || ctx->group_);
   const int hpar = (int)(ctx->loop & 1);
   unsigned long long *hmax = hdisp_on ? (unsigned long long *)(ctx->d_results + R_DISP + 1) : nullptr;
   if (direct && !ctx->halo_fresh)
   {
      int rc0 = ddcmi_mg_refresh_halo(ctx, st);
      if (rc0) return rc0;
      ctx->halo_in_recv = true;      /* (pos[nloc..] keeps the rebuild's records from here on) */
   }
   else if ((ctx->nranks > 1 || ctx->loopback) && !ctx->halo_fresh && !ctx->halo_overlap)
   {
      int rc0 = ddcmi_mg_refresh_halo(ctx, st);
      if (rc0) return rc0;
      ctx->halo_in_recv = false;      /* (the update below places every received bead in pos[]) */
      if (nh > 0)
         hipLaunchKernelGGL(k_halo_update, dim3(cdiv(nh, HU_PER)), dim3(HU_THREADS), 0, st, n, nh, ctx->halo_src.p, ctx->halo_shift.p,
                            ctx->gp.L[0], ctx->gp.L[1], ctx->gp.L[2], ctx->pos.p, ctx->gid.p, false, ctx->hrecv3.p, ctx->hrecv5.p, (const int *)nullptr, hmax, hpar);
   }
   else if ((ctx->nranks > 1 || ctx->loopback) && !ctx->halo_fresh)
   {
      if (!ctx->stream2) HIPCHK(ctx, hipStreamCreateWithFlags(&ctx->stream2, hipStreamNonBlocking));      /* (the rebuild's interior search may have made it already) */
      if (!ctx->ev_drift)
      {
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
   /* (not after a rebuild: it made the images from these very positions; and never on a rank of a transport whose halo is marked fresh --
    * its last exchange or its rebuild placed every image and halo bead, and the receive buffer may hold velocities or nothing by now) */
   else if (self_images(ctx)) { }      /* (the pair kernel finds the images at their owners) */
   else if (nh > 0 && !ctx->images_fresh && !((ctx->nranks > 1 || ctx->loopback) && !ctx->group_))
      hipLaunchKernelGGL(k_halo_update, dim3(cdiv(nh, HU_PER)), dim3(HU_THREADS), 0, st, n, nh, ctx->halo_src.p, ctx->halo_shift.p,
                         ctx->gp.L[0], ctx->gp.L[1], ctx->gp.L[2], ctx->pos.p, ctx->gid.p, false, ctx->hrecv3.p, ctx->hrecv5.p, (const int *)nullptr, hmax, hpar);
   ctx->images_fresh = false;
   const bool has_bonded = (ctx->nbond + ctx->nangle + ctx->ntors + ctx->nrest) > 0;
   const double self = ((ctx->excludePotentialTerm & 128) == 0) ? ctx->self_ele : 0.0;
   if ((ctx->excludePotentialTerm & 128) == 0)
   {
      /* Bonded terms and restraints FIRST, into a zeroed array of force records (ddcmi_ctx::fb): the pair kernel then finishes every bead's
       * force in ONE place -- in memory (plain launch: f = f_pair + f_bonded) or in registers in front of the integrator's pass (FUSE), so
       * that systems with bonded terms take the fused step too (the lipid box paid a separate 54 us kick kernel and a force
       * store + re-read).  Both kinds of launch form the same sum, so the fused and the split step stay bit for bit alike, and both hand
       * the records back zeroed: no launch is ever spent on clearing them. */
      int ntile = ctx->ntile;
      bool useq = ctx->has_charge;
      const bool shbit = ctx->pack_type == 2;
      const size_t capl = (size_t)ctx->stage_cap + 2;      /* + sentinel slot 0, kept even so every LDS array stays 16-byte aligned */
      FuseArgs fa;
      memset(&fa, 0, sizeof(fa));
      NbLds lay = nb_lds_layout(ctx, fuse != nullptr);
      if (fuse)
      {
         /* the integrator's pass rides in the pair kernel only with the fixed layout, and never at the price of the second workgroup per CU */
         const NbLds plain = nb_lds_layout(ctx, false);
         if (lay.zfix && lay.wgs >= plain.wgs && lay.wgs > 0) { fa = *fuse; fa.ke_off = lay.ke_off; }
         else { fuse->dt = 0.0; fuse = nullptr; lay = plain; }
      }
      /* lean (the caller could run this step lean): the fused launch, every tile in ONE launch, the images found at their owners */
      /* (a decomposed rank: its next exchange packs its messages by a launch of its own, mg_pack3, instead of riding in the reduction launch) */
      const bool lean_now = lean && *lean && fuse != nullptr && ctx->ntile_class[1] <= 0 && ((ctx->nranks > 1 || ctx->loopback) ? direct : (nh == 0 || self_images(ctx))) && ctx->lean_pending < LEAN_W &&
                            ctx->lean_since < LEAN_W && (ctx->lean_since == 0 || ctx->lean_dt == fuse->dt);
      if (lean) *lean = lean_now;
      if (has_bonded && n == 0)
         /* a domain that holds no bead at the moment: no bonded launch -- and the sums of the last one it ran must not be reported again (until round 6 an
          * emptied domain kept adding its last bonded energies and virial to the run's; forces were never affected.  tools/soak_migration_r06.py lipid) */
         HIPCHK(ctx, hipMemsetAsync(ctx->d_results + R_SCR_BOND, 0, (R_SCR_MOLV - R_SCR_BOND) * sizeof(double), st));
      if (has_bonded && n > 0)
      {
         if (halo_pending) { HIPCHK(ctx, hipStreamWaitEvent(st, ctx->ev_halo, 0)); halo_pending = false; }      /* bonded partners may be halo beads */
         /* (the record array: all zero -- every pair launch hands back zeroed what it consumed; a grown array is cleared once) */
         if (ctx->fb.cap < (size_t)ctx->npad) { if (ctx->fb.ensure(ctx->npad)) SETERR(ctx, DDCMI_ENOMEM, "bonded force records"); ctx->fb_zeroed = 0; }
         if (ctx->fb_zeroed < ctx->fb.cap) { HIPCHK(ctx, hipMemsetAsync(ctx->fb.p, 0, ctx->fb.cap * sizeof(double4), st)); ctx->fb_zeroed = ctx->fb.cap; }
         ctx->fb_zeroed = 0;      /* (until the pair launches below have consumed the records) */
         int rcb = ddcmi_launch_bonded(ctx, ctx->fb.p, lean_now ? ctx->lean_pending : -1);      /* (lean: the kernels' sums wait in the step's slot of the ring) */
         if (rcb) return rcb;
      }
      /* fixed LDS layout ({x,y} at 0, z at NB_ZOFF) for neighbourhoods of up to NB_ZOFF/16 beads, which is every Martini system; else the run-time layout */
      const bool zfix = lay.zfix, lvl = lay.lvl;
      const size_t lds = lay.total;
      if (lay.wgs <= 0) SETERR(ctx, DDCMI_EUNSUPPORTED, "nonbonded kernel needs %zu bytes of LDS (> 160 KiB)", lds);
      NbTileArgs na;
      na.ntile = ntile; na.stage_stride = ctx->stage_cap; na.cap = (int)capl; na.nlj = ctx->nnb;
      na.cell_start_o = ctx->cell_start_o.p; na.stage_idx = ctx->stage_idx.p; na.tile_nstage = ctx->tile_nstage.p;
      na.cell_start = ctx->cell_start.p; na.cell_cnt = ctx->cell_cnt.p;
      na.tile_base = ctx->tile_base.p; na.tile_width = ctx->tile_width.p; na.tile_rows = ctx->tile_rows.p;
      na.nbr16 = ctx->nbr16.p; na.nbr_cnt = ctx->nbr_cnt.p; na.perm = ctx->tile_perm.p;
      na.tile_work = ctx->tile_work.p; na.halo_shift = ctx->halo_shift.p; na.nloc = n;
      na.disp = ctx->shell_skip ? ctx->d_results + R_DISP : nullptr; na.nbr_cum = ctx->nbr_cum.p; na.sh_r0sq = ctx->sh_r0sq; na.sh_step = ctx->sh_step;
      na.sh_reach[0] = -1e300;
      for (int sq = 1; sq < NSHELL; sq++) na.sh_reach[sq] = sqrt(ctx->sh_r0sq + (double)(sq - 1) * ctx->sh_step) * (1.0 - 1e-4) - ctx->rmax;
      na.hdisp = hdisp_on ? ctx->d_results + R_DISP + 1 + hpar : nullptr;
      na.hrecv3 = ctx->halo_in_recv ? ctx->hrecv3.p : nullptr; na.halo_src = ctx->halo_src.p;
      na.halo_full_walk = (direct && ctx->shell_skip) ? 1 : 0;
      na.lvlidx = ctx->d_lvlidx.p; na.nlvl = ctx->nlvl; na.tab_off = lay.tab_off;
      na.fb = (has_bonded && n > 0) ? ctx->fb.p : nullptr;
      na.self_img = self_images(ctx) ? 1 : 0; na.vring_w = nullptr; na.vring_dt = ctx->lean_dt;
      na.vring = (na.disp && ctx->lean_since > 0) ? ctx->d_vring.p : nullptr; na.vring_n = ctx->lean_since;
      double *partials_p = ctx->partials.p;
      if (lean && *lean)
      {
         const size_t stride = (size_t)(ctx->nitems + 8) * 8;
         if (ctx->lean_pending > 0 && stride != ctx->lean_stride) SETERR(ctx, DDCMI_EINVAL, "internal: the lean steps' rows changed size without a flush");
         ctx->lean_stride = stride;
         ENSURE(ctx, ctx->lean_part, stride * LEAN_W); ENSURE(ctx, ctx->lean_kpart, stride * LEAN_W);
         partials_p = ctx->lean_part.p + stride * ctx->lean_pending;
         fa.kpartials = ctx->lean_kpart.p + stride * ctx->lean_pending;
         ctx->lean_dt = fa.dt; na.vring_dt = fa.dt;
         if (na.disp) na.vring_w = ctx->d_vring.p + (size_t)LEAN_VSTRIDE * ctx->lean_since;
         ctx->lean_since++;
      }
#define LAUNCH_NB(Q, P, S, NT) do { if (zfix) LAUNCH_NBZ(Q, P, S, NT, NB_ZOFF); else LAUNCH_NBZ(Q, P, S, NT, 0); } while (0)
#define LAUNCH_NBZ(Q, P, S, NT, Z) LAUNCH_NBF(Q, P, S, NT, Z, false)
#define LAUNCH_NBF(Q, P, S, NT, Z, F) do { if (lvl) LAUNCH_NBL(Q, P, S, NT, Z, F, true); else LAUNCH_NBL(Q, P, S, NT, Z, F, false); } while (0)
#define LAUNCH_NBL(Q, P, S, NT, Z, F, L) do { \
         size_t sb_ = 0; \
         if (!lds_starts_at_zero(ctx->device, (const void *)k_nonbond<Q, P, S, NT, NB_WPE, NB_CH, Z, F, L>, &sb_)) \
            SETERR(ctx, DDCMI_EUNSUPPORTED, "k_nonbond was built with %zu bytes of static LDS: its staged arrays no longer start at LDS address 0 (toolchain change) -- rebuild libddcmi.so with a compiler that gives it none", sb_); \
         HIPCHK(ctx, dyn_lds_limit(ctx->device, (const void *)k_nonbond<Q, P, S, NT, NB_WPE, NB_CH, Z, F, L>, (int)lds)); \
         hipLaunchKernelGGL((k_nonbond<Q, P, S, NT, NB_WPE, NB_CH, Z, F, L>), dim3(grid), dim3(NT), lds, st, ctx->gp, na, ctx->npad, ctx->pos.p, ctx->d_kqtab.p, \
                            ctx->excl16.p, ctx->excl_cnt.p, (L) ? ctx->d_lvltab.p : ctx->d_ljtab.p, ctx->rmax * ctx->rmax, ctx->krf, ctx->crf, ctx->keR, \
                            ctx->fx.p, ctx->fy.p, ctx->fz.p, partials_p, fa); } while (0)
#define LAUNCH_NB2(Q, P, S) LAUNCH_NB(Q, P, S, NB_THREADS)
      /* class 0: tiles with all-owned neighbourhoods (every tile on a single domain);
       * class 1: tiles that stage image/halo beads, after the halo exchange */
      RoctxRange rng_nb("CHARMM_NONBOND");      /* martiniNonBond, bioMartini.c:989-1122 */
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
         else if (fuse && useq) LAUNCH_NBF(true, false, false, NB_THREADS, NB_ZOFF, true);
         else if (fuse && shbit) LAUNCH_NBF(false, true, true, NB_THREADS, NB_ZOFF, true);
         else if (fuse) LAUNCH_NBF(false, false, false, NB_THREADS, NB_ZOFF, true);
         else if (useq && shbit) LAUNCH_NB2(true, true, true);
         else if (useq) LAUNCH_NB2(true, false, false);
         else if (shbit) LAUNCH_NB2(false, true, true);
         else LAUNCH_NB2(false, false, false);
         if (ctx->timing) HIPCHK(ctx, hipEventRecord(e1, st));
         na.hdisp = hd_keep;
      }
#undef LAUNCH_NB2
#undef LAUNCH_NB
#undef LAUNCH_NBZ
#undef LAUNCH_NBF
#undef LAUNCH_NBL
      if (na.fb) ctx->fb_zeroed = ctx->fb.cap;
      if (ctx->timing) { ctx->t_launches++; if (fuse) ctx->t_launches_fused++; }          /* per force evaluation: the event pairs of both classes add up */
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
   }
   if (halo_pending) HIPCHK(ctx, hipStreamWaitEvent(st, ctx->ev_halo, 0));      /* bonded partners may be halo beads */
   int rc = ddcmi_launch_bonded(ctx);
   if (rc) return rc;
   if (!defer_reduce && (has_bonded || (ctx->excludePotentialTerm & 128) != 0))
      hipLaunchKernelGGL(k_finish_energy, dim3(1), dim3(64), 0, st, ctx->d_results, self);
   ctx->forces_valid = true;
   return DDCMI_OK;
}

/* the lean steps' pending sums, all in one launch (k_reduce_hist) */
int ddcmi_lean_flush(ddcmi_ctx *ctx)
{
   if (ctx->lean_pending <= 0) return DDCMI_OK;
   const int np = ctx->lean_pending;
   ctx->lean_pending = 0;
   if (!ctx->lean_tmp.p)
   {
      const size_t nt = (size_t)2 * LEAN_W * RED_SPLIT * 8 + 2 * LEAN_W;      /* rows, then the jobs' tickets (left at zero by their last workgroup) */
      ENSURE(ctx, ctx->lean_tmp, nt); ENSURE(ctx, ctx->lean_hist, (size_t)LEAN_HW * LEAN_W);
      HIPCHK(ctx, hipMemsetAsync(ctx->lean_hist.p, 0, (size_t)LEAN_HW * LEAN_W * sizeof(double), ctx->stream));
      HIPCHK(ctx, hipMemsetAsync(ctx->lean_tmp.p, 0, nt * sizeof(double), ctx->stream));
   }
   RedJob jf = {ctx->lean_part.p, ctx->nitems, 8, nullptr, 0, 0.0, nullptr};
   RedJob jk = {ctx->lean_kpart.p, ctx->nitems, 7, nullptr, 0, 0.0, nullptr};
   hipLaunchKernelGGL(k_reduce_hist, dim3(RED_SPLIT, 2 * np), dim3(1024), 0, ctx->stream, jf, jk, ctx->lean_stride, np, ctx->lean_hist.p, ctx->lean_tmp.p);
   if ((ctx->nbond + ctx->nangle + ctx->ntors) > 0) { int rcb = ddcmi_lean_flush_bonded(ctx, np); if (rcb) return rcb; }
   ctx->lean_hist_n = np;
   return DDCMI_OK;
}
/* test entry point (ddcmi_test.h): the sums of the lean steps of the last flush, LEAN_HW = 32 per step: pair kernel {lj, ele, virial xx yy zz xy xz yz} as
 * the full list counts them (x 2), {rk, tion xx yy zz xy xz yz}, 0, the bonded kernels' {e_bond, e_angle, e_tors, e_impr, virial xx yy zz xy xz yz}, 0 ... */
extern "C" int ddcmi_debug_lean_history(ddcmi_ctx *ctx, int *nsteps, double *sums)
{
   if (!ctx || !nsteps || !sums) return DDCMI_EINVAL;
   (void)hipSetDevice(ctx->device);
   int rc = ddcmi_lean_flush(ctx);
   if (rc) return rc;
   *nsteps = ctx->lean_hist_n;
   if (ctx->lean_hist_n > 0)
   {
      HIPCHK(ctx, hipMemcpyAsync(sums, ctx->lean_hist.p, (size_t)LEAN_HW * ctx->lean_hist_n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
      HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
   }
   return DDCMI_OK;
}
/* test entry point: the displacement bound's base word and the lean steps' words (largest |v|^2 per step since the rebuild) */
extern "C" int ddcmi_debug_disp(ddcmi_ctx *ctx, double *disp, float *ring, int *nring)
{
   if (!ctx || !disp || !ring || !nring) return DDCMI_EINVAL;
   (void)hipSetDevice(ctx->device);
   unsigned w[LEAN_W * LEAN_VSTRIDE];
   memset(w, 0, sizeof(w));
   HIPCHK(ctx, hipMemcpyAsync(disp, ctx->d_results + R_DISP, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
   if (ctx->d_vring.p) HIPCHK(ctx, hipMemcpyAsync(w, ctx->d_vring.p, sizeof(w), hipMemcpyDeviceToHost, ctx->stream));
   HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
   *nring = ctx->lean_since;
   for (int q = 0; q < LEAN_W; q++) memcpy(ring + q, w + q * LEAN_VSTRIDE, sizeof(float));
   return DDCMI_OK;
}
static int fetch_results(ddcmi_ctx *ctx)
{
   RoctxRange rng_e("EVAL_ETYPE");      /* energyInfo.c:75-148 */
   { int rcl = ddcmi_lean_flush(ctx); if (rcl) return rcl; }
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
   /* new species or nonbonded parameters under an uploaded state (ddcmi_set_species / ddcmi_set_nonbonded without a new upload): the class tables, the
    * beads' tags and with them the list are rebuilt before anything is evaluated with the old ones */
   if (ctx->tables_dirty && ctx->nloc > 0 && (rc = nb_tables(ctx))) return rc;
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
   RoctxRange rng_ke("KINETIC_TERMS");      /* energy.c:48-163 */
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
      /* lean: a single domain of FREE beads with nothing but the pair kernel in its step */
      bool lean = lean_capable(ctx);
      if ((rc = launch_forces(ctx, true, &fa, &lean))) return rc;
      if (fa.dt != 0.0 && lean)
      {
         /* the kernel has left the step's rows of sums in the ring, moved the displacement bound on and will find the images at their owners */
         std::swap(ctx->pos, ctx->pos2);
         ctx->lean_pending++;
         ctx->drift_done = true;
         return ctx->lean_pending >= LEAN_W ? ddcmi_lean_flush(ctx) : DDCMI_OK;
      }
      if (fa.dt != 0.0)
      {
         RedJob jf = {ctx->partials.p, ctx->nitems, 8, ctx->d_results + R_NB_LJ, 1};
         RedJob jk = {ctx->kpartials.p, ctx->nitems, 7, ctx->d_results + R_RK, 0, dt, ctx->d_results + R_DISP};      /* + this drift's share of the displacement bound */
         std::swap(ctx->pos, ctx->pos2);
         PackJob pk;
         memset(&pk, 0, sizeof(pk));
         if (ctx->nhalo > 0 && ctx->nranks == 1 && !ctx->loopback && !self_images(ctx))
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
   if (ctx->comm) { int rca = ddcmi_agree_poll(ctx); if (rca) return rca; }      /* (this path waits on the host every step: a peer whose rebuild failed is reported here, not after the mailbox's timeout) */
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
   ARGCHK(ctx, nsteps < 0, "ddcmi_step_nglf: %d steps", nsteps);
   ARGCHK(ctx, !std::isfinite(dt), "ddcmi_step_nglf: the time step dt = %g is not finite", dt);
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
      RoctxRange rng_step("MDSTEP");      /* nglf.c:67-112 */
      if ((rc = step_pre(ctx, dt))) return rc;
      /* ddcUpdateAll.c:64-71: rebuild when loop % updateRate == 0, or (updateRate == 0) when neighborCheck asks */
      bool due = false;
      if ((rc = rebuild_due(ctx, &due))) return rc;
      if (due) { RoctxRange rng_upd("UPDATEALL PAIRLIST"); if ((rc = ddcmi_build_list(ctx))) return rc; ctx->images_fresh = true; }      /* (nothing moves between here and this step's forces; ddcUpdateAll.c:120-156) */
      if ((rc = step_post(ctx, dt, s + 1 < nsteps))) return rc;
   }
   return DDCMI_OK;
}

