/* ddcmi_nonbond.inl -- k_nonbond: martiniNonBond + martiniIntraMoleReaction over the full list, optionally with the integrator's pass as its epilogue.
 * Part of the ONE translation unit ddcmi.hip (kernels, templates and the static helpers they share), included there in this order. */
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
 *        ir2  = 1/r2   (v_rcp_f64 seed + one Newton step; with charges ir = 1/sqrt(r2) from v_rsq_f64 the same way)
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
/* a staged z with the partner's tags in its lowest mantissa bits (see k_nonbond): bit 0 the shifted-copy flag; bare entries also
 * bits 1-8 the (LJ type, charge) class from the record's tag word */
template <bool PACKED>
__device__ __forceinline__ double z_with_tags(double z, double tagword, bool shifted)
{
   unsigned lo = (unsigned)__double2loint(z);
   if (PACKED) lo = (lo & ~1u) | (shifted ? 1u : 0u);
   else lo = (lo & ~0x1ffu) | (shifted ? 1u : 0u) | (((unsigned)__double2loint(tagword) & 0xffu) << 1);
   return __hiloint2double(__double2hiint(z), (int)lo);
}
/* The lean steps' words of the displacement bound (NbTileArgs::vring): lane s holds W_s, the largest |v|^2 FILED for lean step s -- by the
 * workgroups that lay above LEAN_C times the step before's effective value only (0.9 of its speed: half a percent of the workgroups; 8800 atomic
 * maxima on one word per launch were 10 us of it).  Effective value E_s = max_j LEAN_C^(s-j) W_j >= the step's true maximum: a decaying maximum
 * scan, the same arithmetic in the launch that files (its threshold LEAN_C E_(n-1)) and in every launch that reads. */
#define LEAN_C 0.81f
__device__ __forceinline__ float lean_effective(float w, const int lane)
{
   float e = w, c = LEAN_C;
#pragma unroll
   for (int i = 1; i < LEAN_W; i <<= 1)
   {
      const float t = __shfl_up(e, i, 64);
      if (lane >= i) e = fmaxf(e, t * c);
      c = c * c;
   }
   return e;
}
template <bool HAS_Q, bool PACKED, bool SHBIT, int NB_BLOCK, int WPE, int CH, int ZOFF, bool FUSE, bool LVL>
__global__ __launch_bounds__(NB_BLOCK, WPE) void k_nonbond(GridParams gp, NbTileArgs ta, int npad,
                                                         const double4 *__restrict__ pos, const double *__restrict__ kqtab,
                                                         const unsigned short *__restrict__ excl16, const int *__restrict__ excl_cnt,
                                                         const double4 *__restrict__ ljtab,
                                                         double rc2, double krf, double crf, double keR,
                                                         double *__restrict__ fx, double *__restrict__ fy, double *__restrict__ fz,
                                                         double *__restrict__ partials, FuseArgs fa)
{
   /* LDS: staged neighbourhood as {x,y} pairs + z (24 B per bead), the pair table (and the fused step's rows of kinetic sums) */
   extern __shared__ double2 smem[];
   /* staged positions, 24 B per bead.  ZOFF > 0 (neighbourhoods of up to ZOFF/16 beads: every Martini system): {x,y} [cap] at LDS
    * address 0 and z [cap] at the compile-time byte offset ZOFF, so a gather's addresses are the entry's slot bits themselves
    * (slot * 16 for {x,y}; slot * 8 + the instruction's immediate offset for z).  ZOFF = 0: z [cap], then {x,y} [cap] at a run-time offset */
   double2 *XY_s = ZOFF ? (double2 *)smem : (double2 *)((double *)smem + ta.cap);
   double *Z_s = ZOFF ? (double *)((char *)smem + ZOFF) : (double *)smem;
   double4 *s_lj = (double4 *)((char *)smem + ta.tab_off);      /* (the host places it: nb_lds_layout) */
   /* charges: a bead's "type" is its (LJ type, charge) class, so ke/eps_r q_i q_j is one more
    * per-type-pair table entry -- no per-bead charge array in LDS (it cost 8 B/bead: one
    * workgroup per CU instead of two) and no charge gather per pair */
   /* LVL: s_lj holds the DISTINCT entries of the pair table and L_s one byte per class pair (NbTileArgs::lvlidx) -- 40 Martini types are
    * 1.6 KB + a few hundred bytes instead of 51 KB; one more (one-byte) LDS read per accepted pair */
   const int ntab = LVL ? ta.nlvl : ta.nlj * ta.nlj;
   unsigned char *L_s = (unsigned char *)(s_lj + ntab);
   /* What a list entry does not say about the partner rides in the LOWEST MANTISSA BITS of its staged z (round 5): bit 0 = the bead is a
    * periodically shifted copy (entries without the shift bit: more than 8 classes), bits 1-8 = its (LJ type, charge) class (bare
    * entries: more than 16 classes).  A per-bead byte array for each cost 1 B/bead of LDS -- beside a neighbourhood at the bilayer's
    * density that was the second workgroup per CU (20 LJ types: 0.65 ms against 0.39 ms at 6) -- and one more LDS gather per entry.
    * The price: z moves by < 2^-43 of itself (3e-11 A at 400 A from the origin: 1e-10 relative in a pair force at worst); systems of
    * up to 8 classes carry everything in the entry and keep their z to the last bit. */
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
   float wg_v2 = 0.0f;                        /* (lean step: the workgroup's largest |v|^2, thread 64 + 7) */
   /* the last shell this launch walks (NbTileArgs::disp) */
   int smax = NSHELL - 1;
   /* (the lean steps' words are asked for here and used behind the staging: a wait for them in front of it cost every workgroup a memory round trip) */
   float ring_v2 = 0.0f;
   const double disp_base = ta.disp ? *ta.disp : 0.0;      /* (asked for here as well: behind the staging's barrier it was a round trip of its own, 2 us per workgroup) */
   if (FUSE && ta.disp && ta.vring && (int)threadIdx.x < ta.vring_n) ring_v2 = __uint_as_float(ta.vring[threadIdx.x * LEAN_VSTRIDE]);      /* (words of launches that have ended: an ordinary load, by the first wave only -- 70 000 waves asking for the same twenty cache lines were 7 us of the launch) */
   if (!FUSE && ta.disp && ta.vring && (int)(threadIdx.x & 63) < ta.vring_n) ring_v2 = __uint_as_float(ta.vring[(threadIdx.x & 63) * LEAN_VSTRIDE]);
   if (ta.disp && !ta.vring)
   {
      const double Down = disp_base;
      const double Dhalo = ta.hdisp ? sqrt(*ta.hdisp) : 0.0;
      const double twoD = Down + fmax(Down, Dhalo);
#pragma unroll
      for (int s_ = NSHELL - 1; s_ >= 1; s_--) if (ta.sh_reach[s_] > twoD) smax = s_ - 1;      /* (sh_reach grows with s: the smallest such s decides) */
   }
   int nown = 0, ts = 0, r_lo = 0, r_hi = 0;
   if (ta.halo_full_walk && mine && ((ta.tile_work[t] >> 30) & 1)) smax = NSHELL - 1;      /* (received beads nobody measures: NbTileArgs::hrecv3) */
   if (mine)
   {
      ts = ta.cell_start_o[TCELLS * t];
      nown = ta.cell_start_o[TCELLS * t + TCELLS] - ts;
      r_lo = (int)(((long long)nown * part) / nparts); r_hi = (int)(((long long)nown * (part + 1)) / nparts);
   }
   if (nown > 0)
   {
      for (int k = threadIdx.x; k < ntab; k += NB_BLOCK) { double4 e_ = ljtab[k]; if (HAS_Q && !LVL) e_.w = kqtab[k]; s_lj[k] = e_; }      /* (LVL: the entries come with their fourth word) */
      if (LVL) for (int k = threadIdx.x; k < ta.nlj * ta.nlj; k += NB_BLOCK) L_s[k] = ta.lvlidx[k];

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
         unsigned char *shc_s = (unsigned char *)(s_w + 16);          /* [NRC]  self_img: shift code of an image cell (13: none) */
         unsigned short *cellof = (unsigned short *)(shc_s + ((NRC + 15) & ~15));      /* [ns]   region cell of each staged slot */
         constexpr int CPT = (NRC + NB_BLOCK - 1) / NB_BLOCK;
         const int tx = t % gp.T[0], ty = (t / gp.T[0]) % gp.T[1], tz = t / (gp.T[0] * gp.T[1]);
         int v[CPT], g[CPT], sc[CPT], vsum = 0;
         const bool from_owner = ta.self_img != 0 && tshift;      /* a single domain's images: the owner's record + the shift, found by cell arithmetic (image_dirs) */
#pragma unroll
         for (int h = 0; h < CPT; h++)
         {
            const int c = CPT * (int)threadIdx.x + h;
            v[h] = 0; g[h] = 0; sc[h] = 13;
            if (c < NRC)
            {
               const int cx = TCX * tx - 2 + (c % RGX), cy = TCY * ty - 2 + ((c / RGX) % RGY), cz = TCZ * tz - 2 + (c / (RGX * RGY));
               if (cx >= 0 && cy >= 0 && cz >= 0 && cx < gp.g[0] && cy < gp.g[1] && cz < gp.g[2])
               {
                  const int id = cell_linear(gp, cx, cy, cz);
                  v[h] = ta.cell_cnt[id];
                  /* an image cell (known by its coordinates: no load to wait for) holds the beads of the owned cell a whole box away, one for
                   * one and in their order: stage those */
                  const int cc[3] = {cx, cy, cz};
                  int oc[3], code = 0, w3 = 1;
                  bool img = false;
#pragma unroll
                  for (int a = 0; a < 3; a++)
                  {
                     const bool lowside = cc[a] < gp.m[a], highside = cc[a] >= gp.m[a] + gp.n[a];
                     oc[a] = cc[a] + (lowside ? gp.n[a] : highside ? -gp.n[a] : 0);
                     code += w3 * (lowside ? 0 : highside ? 2 : 1);      /* the image lies at owner - L | owner + L */
                     w3 *= 3;
                     img |= lowside | highside;
                  }
                  if (from_owner && img) { g[h] = ta.cell_start_o[cell_linear(gp, oc[0], oc[1], oc[2])]; sc[h] = code; }
                  else g[h] = ta.cell_start[id];
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
               if (from_owner) shc_s[c] = (unsigned char)sc[h];
               for (int j = 0; j < v[h]; j++) cellof[ex + j] = (unsigned short)c;
            }
            ex += v[h];
         }
         __syncthreads();
         int gj[MAXR];
         unsigned long long simg = 0ull;      /* self_img: the rounds' shift codes, 5 bits each (13: not an image) */
         static_assert(MAXR * 5 <= 64, "shift codes of the staging rounds");
#pragma unroll
         for (int u = 0; u < MAXR; u++)
         {
            const int k = (int)threadIdx.x + u * NB_BLOCK;
            gj[u] = ts;      /* rounds past the end re-read the tile's first bead (a cache hit) and drop it */
            if (k < ns)
            {
               const int c = cellof[k];
               gj[u] = gst_s[c] + (k - ofs_s[c]);
               if (from_owner) simg |= (unsigned long long)shc_s[c] << (5 * u);
            }
            else if (from_owner) simg |= 13ull << (5 * u);
         }
         /* direct halo staging: a received bead's position is in the exchange's receive buffer (NbTileArgs::hrecv3); its place there
          * is asked for here, for all rounds at once, so the one dependent round trip is paid once per tile */
         const bool from_recv = ta.hrecv3 != nullptr && tshift;
         int hk[MAXR];
         if (from_recv)
         {
#pragma unroll
            for (int u = 0; u < MAXR; u++) hk[u] = (gj[u] >= ta.nloc) ? ta.halo_src[gj[u] - ta.nloc] : 0;
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
               if (from_recv && hk[b + u] < 0)
               {
                  const double *rr = ta.hrecv3 + 3 * (size_t)(-1 - hk[b + u]);
                  pp[u].x = rr[0]; pp[u].y = rr[1]; pp[u].z = rr[2];
                  pp[u].w = PACKED ? 0.0 : pos[gj[b + u]].w;      /* (packed entries carry the partner's type themselves) */
               }
               else pp[u] = pos[gj[b + u]];
               if (from_owner) sh[u] = (int)((simg >> (5 * (b + u))) & 31ull);
               else sh[u] = (!SHBIT && tshift && gj[b + u] >= ta.nloc) ? ta.halo_shift[gj[b + u] - ta.nloc] : 13;
            }
#pragma unroll
            for (int u = 0; u < SU; u++)
            {
               const int k = (int)threadIdx.x + (b + u) * NB_BLOCK;
               if (from_owner && sh[u] != 13)
               {
                  /* (the image update's own arithmetic: k_reduce_jobs_images, k_halo_update) */
                  pp[u].x += (double)(sh[u] % 3 - 1) * gp.L[0];
                  pp[u].y += (double)((sh[u] / 3) % 3 - 1) * gp.L[1];
                  pp[u].z += (double)(sh[u] / 9 - 1) * gp.L[2];
               }
               if (k < ns)
               {
                  XY_s[k + 1] = make_double2(pp[u].x, pp[u].y);
                  Z_s[k + 1] = SHBIT ? pp[u].z : z_with_tags<PACKED>(pp[u].z, pp[u].w, tshift && sh[u] != 13);
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
               Z_s[k + 1] = SHBIT ? pp[u].z : z_with_tags<PACKED>(pp[u].z, pp[u].w, tshift && sh[u] != 13);
            }
         }
      }
      }
      if (FUSE && threadIdx.x < (NB_BLOCK / 64) * 8) ((double *)((char *)smem + fa.ke_off))[threadIdx.x] = 0.0;
      if (FUSE && ta.disp && ta.vring && threadIdx.x < 64)
      {
         /* the lean steps' share of the displacement bound: their largest |v|^2, one word each, added up by the first wave for all */
         const float eff = lean_effective(ring_v2, (int)threadIdx.x);      /* (lanes behind the last word hold decayed values: not steps) */
         const double sum = wave_sum_dpp(((int)threadIdx.x < ta.vring_n ? ta.vring_dt * sqrt((double)eff) : 0.0) * (1.0 + 2e-6)) * (1.0 + 1e-6);      /* (every term and the sum rounded up, with room for the scan's roundings) */
         const float thr = ta.vring_n > 0 ? __shfl(eff, ta.vring_n - 1, 64) * LEAN_C : 0.0f;      /* this step's workgroups file what lies above it */
         if (threadIdx.x == 0) { double *h = (double *)((char *)smem + fa.ke_off) + (NB_BLOCK / 64) * 8; h[0] = sum; h[1] = (double)thr; }
      }
      if (threadIdx.x == 0)
      {
         /* staged slot 0: a bead far outside every cutoff.  List padding (entry 0) points
          * at it, so the walk needs no per-slot validity masks. */
         XY_s[0] = make_double2(1e30, 1e30); Z_s[0] = SHBIT ? 1e30 : z_with_tags<PACKED>(1e30, 0.0, false);
      }
      __syncthreads();
      if (ta.disp && ta.vring && !(ta.halo_full_walk && tshift))      /* (tiles that stage received beads walk their rows to the end) */
      {
         /* the displacement bound + the lean steps since the rebuild: their largest |v|^2, one word each, added up in the same order by every wave
          * (a single domain: no received beads, no hdisp) */
         const double Down = disp_base + (FUSE ? ((const double *)((const char *)smem + fa.ke_off))[(NB_BLOCK / 64) * 8]
                                                : wave_sum_dpp(((int)(threadIdx.x & 63) < ta.vring_n ? ta.vring_dt * sqrt((double)lean_effective(ring_v2, (int)(threadIdx.x & 63))) : 0.0) * (1.0 + 2e-6)) * (1.0 + 1e-6));
         const double twoD = 2.0 * Down;
#pragma unroll
         for (int s_ = NSHELL - 1; s_ >= 1; s_--) if (ta.sh_reach[s_] > twoD) smax = s_ - 1;
      }
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
         /* entries without the shift bit: the bead meets its partners at the z its partners meet IT at -- its staged z, tag bits and
          * all (z_with_tags) -- so that f_ij = -f_ji holds to the last bit, as it does for an exact z.  (With the exact z here the
          * two sides of a pair differed by 1e-11 of the force, which the per-bead virial 2 F_i (x) r_i multiplies by the bead's
          * distance from the origin: 1e-8 of the virial on the 2 M-bead bilayer.)  The drift starts from the exact value. */
         const double zexact = pi.z;
         if (!SHBIT) pi.z = z_with_tags<PACKED>(pi.z, pi.w, false);
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
                  int tjj = PACKED ? (SHBIT ? (nib_ & 7) : nib_) : (int)((zb[u] >> 1) & 0xffu); \
                  double4 lj = s_lj[LVL ? (int)L_s[ti * nlj + tjj] : ti * nlj + tjj];            /* {sigma^2, 4eps, shift, 24eps} */ \
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
                  if (tshift && (SHBIT ? (nib_ & 8) : (int)(zb[u] & 1u))) \
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
               unsigned o[CH], zb[CH];      /* zb: the low word of the partner's staged z (its tag bits, z_with_tags) */
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
                  zb[u] = (unsigned)__double2loint(pz);
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
               const int tje = PACKED ? (SHBIT ? (int)(e16 & 7u) : (int)(e16 & 0xfu)) : (int)(((unsigned)__double2loint(pz) >> 1) & 0xffu);
               double x = pi.x - pxy.x, y = pi.y - pxy.y, z = pi.z - pz;
               double r2 = x * x + y * y + z * z;
               if (r2 < rc2)
               {
                  double kqij = s_lj[LVL ? (int)L_s[ti * nlj + tje] : ti * nlj + tje].w;
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
               if (ta.fb) { const double4 b = ta.fb[a]; fxi += b.x; fyi += b.y; fzi += b.z; ta.fb[a] = make_double4(0.0, 0.0, 0.0, 0.0); }
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
               if (ta.fb)
               {
                  /* + the bonded terms' force on the bead (the same sum the plain launch leaves in memory); the record goes back zeroed */
                  const double4 b = ta.fb[a];
                  fxi += b.x; fyi += b.y; fzi += b.z;
                  ta.fb[a] = make_double4(0.0, 0.0, 0.0, 0.0);
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
               if (!SHBIT) p.z = zexact;
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
         if (k == 7) wg_v2 = (float)a;      /* (a float's value held in a double: the conversion is exact) */
      }
   }
   /* lean step: the workgroup's largest |v|^2 joins the step's word of the ring by an atomic maximum of the float's bits (non-negative
    * floats order like their bits: exact and order-free; no return value, no fence -- a device-scope release here writes back an XCD's L2
    * in the middle of the launch's 340 MB of stores: +40 % on the kernel) */
   if (FUSE && ta.vring_w && threadIdx.x == 64 + 7)
   {
      const float thr = (ta.vring && nown > 0) ? (float)((const double *)((const char *)smem + fa.ke_off))[(NB_BLOCK / 64) * 8 + 1] : 0.0f;      /* (the first lean step after a rebuild: everyone files) */
      if (wg_v2 > thr) atomicMax(ta.vring_w, __float_as_uint(wg_v2));
   }
}
