/* ddcmi_rebuild.inl -- host side of a list rebuild: grid, sort, periodic images, tile schedule, ddcmi_bl_finish.
 * Part of the ONE translation unit ddcmi.hip (kernels, templates and the static helpers they share), included there in this order. */
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
   if (ctx->stage_cap_want > 0 && ctx->stage_cap_want < ctx->stage_cap) ctx->stage_cap = ctx->stage_cap_want;      /* (the old list is dead from here on) */
   ctx->stage_cap_want = 0;
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
   if (ctx->nhalo_hint > 0)
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

static int bl_launch_interior(ddcmi_ctx *ctx);
static NbLds nb_lds_layout(const ddcmi_ctx *ctx, bool fused);
int ddcmi_lean_flush(ddcmi_ctx *ctx);
/* what the setters could not check one by one (they may come in any order): every species' molecule type has its entry in the molecule
 * tables, every index-named term, constraint pair and molecule names a bead of the uploaded state -- the search and the bonded kernels index
 * with them.  Every rebuild starts here: ddcmi_build_list, and the first phase of a decomposed rank's or an in-process group's rebuild. */
static int bl_validate_tables(ddcmi_ctx *ctx)
{
   if (ctx->nmoltype > 0)
      for (int sp = 0; sp < ctx->nspecies; sp++)
         if (ctx->moltype[sp] < 0 || ctx->moltype[sp] >= ctx->nmoltype)
            SETERR(ctx, DDCMI_EINVAL, "species %d: molecule type %d outside the %d types of ddcmi_set_molecules", sp, ctx->moltype[sp], ctx->nmoltype);
   const bool one = ctx->nranks == 1 && !ctx->loopback && !ctx->group_;      /* (several domains: index-named tables are refused where they are set) */
   if (one && !ctx->bonded_gid && ctx->inc_nrow > ctx->nloc)
      SETERR(ctx, DDCMI_EINVAL, "a bonded term of ddcmi_set_bonded names bead %d, the uploaded state holds %d", ctx->inc_nrow - 1, ctx->nloc);
   if (one && !ctx->cons_gid && ctx->ncgroup > 0 && ctx->idx_amax_cons >= ctx->nloc)
      SETERR(ctx, DDCMI_EINVAL, "a constraint pair of ddcmi_set_constraints names bead %d, the uploaded state holds %d", ctx->idx_amax_cons, ctx->nloc);
   if (one && !ctx->mol_gid && ctx->nmol_multi > 0 && ctx->idx_amax_mol >= ctx->nloc)
      SETERR(ctx, DDCMI_EINVAL, "a molecule of ddcmi_set_molecule_lists names bead %d, the uploaded state holds %d", ctx->idx_amax_mol, ctx->nloc);
   return DDCMI_OK;
}
static void bl_drop_interior(ddcmi_ctx *ctx);
extern "C" int ddcmi_build_list(ddcmi_ctx *ctx)
{
   if (!ctx) return DDCMI_EINVAL;
   /* an interior search a failed rebuild left in flight on the second stream (an error between bl_launch_interior and ddcmi_bl_finish: the
    * self-image count, the halo sort, a post, a phase of the decomposed path) must not meet this rebuild's sort, which rewrites the
    * positions and cell tables it reads and zeroes the counters it adds to */
   bl_drop_interior(ctx);
   { int rcl = ddcmi_lean_flush(ctx); if (rcl) return rcl; }      /* (the pending steps' rows are as many as this list's work items) */
   if (!ctx->have_box || ctx->nlj <= 0 || ctx->updateRate < 0 || (ctx->nloc <= 0 && ctx->nranks == 1))
      SETERR(ctx, DDCMI_EINVAL, "ddcmi_build_list needs box, nonbonded parameters, neighbor settings and an uploaded state");
   { int rcv = bl_validate_tables(ctx); if (rcv) return rcv; }
   (void)hipSetDevice(ctx->device);
   int rc;
   ctx->pack_fresh = false;      /* (the rebuild's own exchange reuses the send buffer) */
   if ((rc = nb_tables(ctx))) return rc;
   /* (a rebuild places every image and halo bead from the positions it sorts: the force evaluation that follows needs no update of
    * them -- and a decomposed rank has no 3-wide records to take one from until its first per-step exchange) */
   if (ctx->nranks > 1 || ctx->loopback) { rc = ddcmi_mg_rebuild(ctx); if (!rc) ctx->images_fresh = true; return rc; }
   ctx->phase(-1, nullptr);
   for (int pass = 0;; pass++)
   {
      if ((rc = ddcmi_bl_sort_owned(ctx))) return rc;      /* (idempotent: a second pass sorts sorted beads) */
      if ((rc = bl_launch_interior(ctx))) return rc;
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
   ctx->images_fresh = true;
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
         constexpr int lpt_rounds = 8;
         if (n < lpt_rounds * S)
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
         else if (n > 0 && n < 8 * S)
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
   if (!ctx->no_lean && !ctx->d_vring.p) ENSURE(ctx, ctx->d_vring, (size_t)LEAN_W * LEAN_VSTRIDE);      /* (zeroed by the tail launch below, like the bound) */
   {
      /* the tile order and the ranges, the displacement words of the shell-limited walk back to zero, and the NEXT rebuild's counters
       * cleared while nothing reads them (ddcmi_bl_sort_owned, mg_phase1_launch): one launch */
      TailJobs tj;
      tj.fetch(ctx->tile_perm.p, perm, std::max(nitems, 1)).fetch(ctx->sched.p, sched, 32);
      const int ncell = ctx->gp.ncell;
      tj.zero.add(ctx->d_results + R_DISP, 6);      /* three doubles */
      if (ctx->d_vring.p) tj.zero.add(ctx->d_vring.p, LEAN_W * LEAN_VSTRIDE);      /* (and the lean steps' words of the bound) */
      ctx->lean_since = 0;
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
/* sizes, buffers and kernel arguments of the search: everything k_tile_build needs, from the sorted owned beads alone (so that the
 * interior tiles' search can start before the halo exists: bl_launch_interior).  Idempotent. */
typedef void (*tile_build_fn)(GridParams, TileArgs, int, const double4 *, const uint64_t *, const int *, int, const int *, const int *, const int *,
                              const int *, const int *, const unsigned long long *, int, unsigned short *, int *, int *, TileSel);
struct BuildPlan { TileArgs ta; size_t lds; tile_build_fn kbuild; ShellCuts shc; };
static int bl_plan(ddcmi_ctx *ctx, BuildPlan &bp)
{
   GridParams &gp = ctx->gp;
   const int n = ctx->nloc;
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
   {
      const double r0 = rcut - 0.25 * dR;
      bp.shc.r0sq = (float)(r0 * r0); bp.shc.one = !(dR > 1e-9 * rcut);      /* no skin: one shell */
   }
   unsigned long long *d_arena = (unsigned long long *)(ctx->d_flags + 32);    /* arena entries handed out: the one device-wide counter of the build, on a cache line of its own */
   bool has_mol = false;
   for (int m = 0; m < ctx->nmoltype; m++) has_mol |= ctx->mol_nspecies[m] > 1;
   /* LDS image: the ring of accepted words (16 KB), 16 B per staged bead (+ 2 B of molecule id when pairs can be excluded), the region cell tables */
   size_t lds = TB_RING_BYTES + (size_t)ctx->stage_cap * (has_mol ? 18 : 16) + (2 * NRC + 16 + 2 * (TB_THREADS / 64) + 8) * sizeof(int) + 16;
   if ((size_t)ctx->stage_cap * sizeof(unsigned short) > TB_RING_BYTES) lds += (size_t)ctx->stage_cap * sizeof(unsigned short);      /* (bare 16-bit entries: the slot -> cell map outgrows the ring) */
   if (lds > 160 * 1024) SETERR(ctx, DDCMI_EUNSUPPORTED, "a tile neighbourhood of %d beads does not fit the 160 KiB LDS", ctx->stage_cap);
   ENSURE(ctx, ctx->stage_idx, (size_t)ntile * ctx->stage_cap);
   if (ctx->nbr16.ensure(ctx->arena_cap)) SETERR(ctx, DDCMI_ENOMEM, "neighbour arena of %llu entries failed", ctx->arena_cap);
   if (ctx->excl16.ensure((size_t)ctx->maxexcl * ctx->npad)) SETERR(ctx, DDCMI_ENOMEM, "excluded-pair entries");
   /* (round 6: the middle form of rounds 4-5 -- 9 to 16 classes, the class in the nibble and the shift flag in the staged z -- is gone: it was
    * 4 % faster than bare entries for decks of exactly that size and a third of k_nonbond's instantiations; such decks take bare entries now) */
   ctx->pack_type = (ctx->stage_cap < 4096 && ctx->nnb <= 8) ? 2 : 0;
   TileArgs &ta = bp.ta;
   ta.ntile = ntile; ta.stage_stride = ctx->stage_cap; ta.cap = ctx->stage_cap; ta.pack_type = ctx->pack_type; ta.nloc = n; ta.halo_shift = ctx->halo_shift.p;
   ta.cell_start_o = ctx->cell_start_o.p; ta.cell_start = ctx->cell_start.p; ta.cell_cnt = ctx->cell_cnt.p;
   ta.stage_idx = ctx->stage_idx.p; ta.tile_nstage = ctx->tile_nstage.p;
   ta.tile_base = ctx->tile_base.p; ta.tile_width = ctx->tile_width.p; ta.tile_rows = ctx->tile_rows.p; ta.tile_work = ctx->tile_work.p;
   ta.nbr16 = ctx->nbr16.p; ta.arena_cap = ctx->arena_cap; ta.arena_used = d_arena;
   ta.nbr_cnt = ctx->nbr_cnt.p; ta.nbr_cum = ctx->nbr_cum.p;
   if (ctx->tmp32.ensure(((size_t)ctx->npad + (size_t)TB_CHUNK * (ntile + 1)) * ctx->tmpw)) SETERR(ctx, DDCMI_ENOMEM, "scratch list allocation failed");      /* every tile rounded up to whole chunks */
   ta.tmp32 = ctx->tmp32.p; ta.tmpw = ctx->tmpw; ta.shc = bp.shc;
   if (ctx->pack_type && ctx->tile_nib.ensure((size_t)ntile * ctx->stage_cap + 16)) SETERR(ctx, DDCMI_ENOMEM, "nibble table allocation failed");
   ta.tile_nib = ctx->tile_nib.p;
   bp.kbuild = has_mol ? (ctx->pack_type == 2 ? k_tile_build<true, 2> : k_tile_build<true, 0>)
                       : (ctx->pack_type == 2 ? k_tile_build<false, 2> : k_tile_build<false, 0>);
   bp.lds = lds;
   HIPCHK(ctx, dyn_lds_limit(ctx->device, (const void *)bp.kbuild, (int)lds));
   return DDCMI_OK;
}
static void bl_launch_tiles(ddcmi_ctx *ctx, const BuildPlan &bp, hipStream_t st, const TileSel &sel, int nblocks)
{
   hipLaunchKernelGGL(bp.kbuild, dim3(nblocks), dim3(TB_THREADS), bp.lds, st, ctx->gp, bp.ta, ctx->npad, ctx->pos.p, ctx->gid.p, ctx->species.p,
                      ctx->nmoltype, ctx->d_moltype_sp.p, ctx->d_mol_nspecies.p, ctx->d_bpair_off.p, ctx->d_bpairI.p, ctx->d_bpairJ.p, ctx->d_exmask.p,
                      ctx->maxexcl, ctx->excl16.p, ctx->excl_cnt.p, ctx->d_flags, sel);
}
/* The interior tiles' search, started as soon as the owned beads are sorted (round 5).  A tile whose neighbourhood -- the tile plus
 * two cells on every side -- lies inside the owned cells stages owned beads only: nothing of the halo enters its search.  On a
 * decomposed rank that is half the tiles, and between the owned sort and the halo's arrival the rebuild exchanges counts (two host
 * round trips), packs, runs the RCCL kernel, assembles and sorts the halo: ~170 us of a 500 k-bead brick's 740 us window in which
 * the chip ran a dozen latency-bound launches or nothing.  The interior launch goes to a second stream behind the sort and reads
 * the owned cell tables; the boundary launch follows on the main stream when the halo is in place; the transposition waits for both.
 * (Rounds 5's DDCMI_NO_INTERIOR_FIRST -- one launch -- was never faster and is gone.) */
static void bl_drop_interior(ddcmi_ctx *ctx);
static int bl_launch_interior(ddcmi_ctx *ctx)
{
   bl_drop_interior(ctx);      /* (a rebuild that ended early on an error may have left one behind) */
   if (ctx->nloc <= 0) return DDCMI_OK;
   const GridParams &gp = ctx->gp;
   const int tdim[3] = {TCX, TCY, TCZ};
   TileSel sel;
   sel.mode = 1;
   long nin = 1;
   for (int a = 0; a < 3; a++)
   {
      int lo = 0, hi = gp.T[a] - 1;
      if (gp.m[a] > 0)
      {
         /* region cells [tdim t - 2, tdim (t + 1) + 2) inside the owned cells [m, m + n) */
         lo = gp.T[a]; hi = -1;
         for (int t = 0; t < gp.T[a]; t++)
            if (tdim[a] * t - 2 >= gp.m[a] && tdim[a] * (t + 1) + 2 <= gp.m[a] + gp.n[a]) { lo = std::min(lo, t); hi = std::max(hi, t); }
      }
      sel.lo[a] = lo; sel.n[a] = hi - lo + 1;
      if (sel.n[a] <= 0) return DDCMI_OK;
      nin *= sel.n[a];
   }
   const long ntile = (long)gp.T[0] * gp.T[1] * gp.T[2];
   if (nin < 64 || 8 * nin < ntile) return DDCMI_OK;      /* (too few to be worth a launch of their own) */
   BuildPlan bp;
   int rc = bl_plan(ctx, bp);
   if (rc) return rc;
   if (!ctx->stream2) HIPCHK(ctx, hipStreamCreateWithFlags(&ctx->stream2, hipStreamNonBlocking));
   if (!ctx->ev_sorted) { HIPCHK(ctx, hipEventCreateWithFlags(&ctx->ev_sorted, hipEventDisableTiming)); HIPCHK(ctx, hipEventCreateWithFlags(&ctx->ev_interior, hipEventDisableTiming)); }
   HIPCHK(ctx, hipEventRecord(ctx->ev_sorted, ctx->stream));
   HIPCHK(ctx, hipStreamWaitEvent(ctx->stream2, ctx->ev_sorted, 0));
   bp.ta.cell_start = ctx->cell_start_o.p; bp.ta.cell_cnt = ctx->cell_cnt_o.p;      /* (interior neighbourhoods: the merged tables, which do not exist yet, say the same) */
   bl_launch_tiles(ctx, bp, ctx->stream2, sel, (int)nin);
   HIPCHK(ctx, hipEventRecord(ctx->ev_interior, ctx->stream2));
   ctx->interior_launched = true; ctx->interior_sel = sel;
   ctx->interior_key[0] = ctx->stage_cap; ctx->interior_key[1] = ctx->tmpw; ctx->interior_key[2] = ctx->maxexcl; ctx->interior_key[3] = (long long)ctx->arena_cap;
   return DDCMI_OK;
}
/* an early interior launch that cannot be used (the rebuild starts over): wait for it, forget it */
static void bl_drop_interior(ddcmi_ctx *ctx)
{
   if (ctx->interior_launched && ctx->stream2) (void)hipStreamSynchronize(ctx->stream2);
   ctx->interior_launched = false;
}
int ddcmi_bl_finish(ddcmi_ctx *ctx)
{
   GridParams &gp = ctx->gp;
   hipStream_t st = ctx->stream;
   int n = ctx->nloc;
   ctx->phase(10, "-> bl_finish");
   int ntile = gp.T[0] * gp.T[1] * gp.T[2];
   unsigned long long *d_arena = (unsigned long long *)(ctx->d_flags + 32);
   ShellCuts shc;
   for (int attempt = 0;; attempt++)
   {
      if (attempt == 8) { bl_drop_interior(ctx); SETERR(ctx, DDCMI_ENOMEM, "neighbour list capacity could not be settled"); }
      if (attempt > 0)
      {
         bl_drop_interior(ctx);
         ddcmi_zero_ints(ctx, st, ZeroJobs().add(ctx->d_flags, 8).add(d_arena, 2));      /* first attempt: zeroed with the cell counters (ddcmi_bl_sort_owned) */
      }
      BuildPlan bp;
      { int rcp = bl_plan(ctx, bp); if (rcp) { bl_drop_interior(ctx); return rcp; } }
      shc = bp.shc;
      TileArgs &ta = bp.ta;
      ctx->phase(17, "bl_finish: buffers");
      /* (an interior launch made for other capacities cannot be completed: cannot happen -- the capacities only change in this loop) */
      const bool split = ctx->interior_launched && ctx->interior_key[0] == ctx->stage_cap && ctx->interior_key[1] == ctx->tmpw &&
                         ctx->interior_key[2] == ctx->maxexcl && ctx->interior_key[3] == (long long)ctx->arena_cap;
      if (ctx->interior_launched && !split) bl_drop_interior(ctx);
      TileSel sel;
      memset(&sel, 0, sizeof(sel));
      if (split) { sel = ctx->interior_sel; sel.mode = 2; }
      bl_launch_tiles(ctx, bp, st, sel, ntile);
      if (split)
      {
         HIPCHK(ctx, hipStreamWaitEvent(st, ctx->ev_interior, 0));      /* the transposition and the post below need every tile */
         ctx->interior_launched = false;
      }
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
         /* a box of fewer tiles than the chip has CUs: up to eight workgroups per tile (the 16 tiles of a 6.9 k-bead box took 61 us as 16 workgroups) */
         const int ny = ntile < 256 ? std::max(1, std::min(8, 1024 / std::max(ntile, 1))) : 1;
         hipLaunchKernelGGL(ktr, dim3(ntile, ny), dim3(TR_THREADS), lds2, st, ta);
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
      if (ctx->h_flags[4] > 0) { ctx->stage_cap = (((int)(ctx->h_flags[4] * 1.02) + 40) + 63) & ~63; again = true; }
      if (ctx->h_flags[0] > 0) { ctx->arena_cap = (unsigned long long)((double)tot[2] * 1.10) + 65536ull; again = true; }
      if (ctx->h_flags[1] > 0) { ctx->maxexcl = ctx->h_flags[1] + 4; again = true; }
      if (ctx->h_flags[5] > 0) { ctx->tmpw = ((int)(ctx->h_flags[5] * 1.1) + 8 + 7) & ~7; again = true; }
      if (ctx->tmpw > 768) SETERR(ctx, DDCMI_EUNSUPPORTED, "neighbour lists of more than 768 entries per bead (list radius %g) are not supported", gp.rlist);
      if (again) HIPCHK(ctx, hipStreamSynchronize(st));      /* the transposition still runs on buffers the next attempt may grow */
      if (!again)
      {
         ctx->list_entries = (int64_t)tot[0]; ctx->excl_entries = (int64_t)tot[1];
         ctx->maxnbr = maxw;
         {
            /* the largest neighbourhood of this build (the tiles' staging costs are 7 x their bead counts): what the NEXT rebuild needs */
            int mx = 0;
            for (int t = 0; t < ntile; t++) mx = std::max(mx, h_work[(size_t)ntile + t] / 7);
            ctx->stage_cap_want = std::max(384, (((int)(mx * 1.015) + 40) + 63) & ~63);
         }
         break;
      }
   }
   {
      /* workgroups of k_nonbond a CU holds: its LDS image of a neighbourhood (launch_forces), at most two by registers */
      int rcs = schedule_tiles(ctx, std::max(1, nb_lds_layout(ctx, true).wgs));
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

