/* ddcmi_listbuild.inl -- the neighbour-list build: k_tile_build (search) and k_tile_transpose (slot-major slices in shell order).
 * Part of the ONE translation unit ddcmi.hip (kernels, templates and the static helpers they share), included there in this order. */
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
   double sh_reach[NSHELL];             /* [s], s >= 1: how far shell s begins beyond the cut-off, sqrt(sh_r0sq + (s - 1) sh_step) (1 - 1e-4) - r_cut: the walk ends with shell s - 1 while that exceeds the pairs' possible approach (formed once on the host: a square root per shell and wave was 6 % of the kernel's instructions) */
   /* decomposed runs: D bounds the OWNED beads' moves only; hdisp (not null) points at the largest squared distance of a received
    * halo bead from its place at the rebuild (k_halo_update), and a pair distance has changed by at most D + max(D, sqrt(*hdisp)) */
   const double *hdisp;
   /* decomposed runs, direct halo staging (ddcmi_ctx::halo_in_recv): received bead h = staged global index - nloc lies at
    * hrecv3[3 k], k = -1 - halo_src[h] (the per-step exchange's receive buffer, sender's order); its record in pos[] carries the
    * tag only.  Tiles that stage such beads walk their rows to the end (halo_full_walk): the displacement bound D covers this
    * rank's beads, and nothing measures the neighbours' any more -- the shell-limited walk stays with the all-owned tiles. */
   const double *hrecv3; const int *halo_src; int halo_full_walk;
   /* the lean step of a single domain (launch_forces, ddcmi_ctx::lean_pending): nothing runs between two pair kernels.  self_img: the
    * periodic images are staged from their OWNERS' records + the shift (halo_src >= 0, halo_shift) -- no image update launch.  The
    * displacement bound: every workgroup of a lean step files its largest |v|^2 in the step's word vring_w of a ring of LEAN_W (atomic maximum
    * of the float's bits), and every launch until the next rebuild adds vring_dt sqrt(word) over the ring to *disp (vring not null) */
   int self_img; const unsigned *vring; int vring_n /* words in use */; unsigned *vring_w; double vring_dt;
   /* k_nonbond<..., LVL>: the pair table in two levels (ddcmi_ctx::d_lvltab): lvlidx [nlj*nlj] = index of the class pair's entry among the nlvl distinct ones */
   const unsigned char *lvlidx; int nlvl;
   int tab_off;                         /* LDS byte offset of the pair table (nb_lds_layout: the gap between {x,y} and z when it fits, else behind z) */
   /* bonded terms / restraints: their kernels ran first and left their force on every owned bead in fb (one 32-byte record per bead,
    * zero where a bead has none); the pair kernel adds the bead's pair force to it -- into fx, fy, fz (plain launch) or in registers,
    * in front of the integrator's pass (FUSE) -- and hands the record back zeroed for the next evaluation's bonded kernels */
   double4 *fb;
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
                                                            int maxexcl, unsigned short *excl16, int *excl_cnt, int *flags, TileSel sel)
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
   if (sel.mode == 1)
   {
      const int b = (int)blockIdx.x, bx = b % sel.n[0], by = (b / sel.n[0]) % sel.n[1], bz = b / (sel.n[0] * sel.n[1]);
      t = ((sel.lo[2] + bz) * gp.T[1] + (sel.lo[1] + by)) * gp.T[0] + (sel.lo[0] + bx);
   }
   else if (sel.mode == 2)
   {
      const int qx = t % gp.T[0] - sel.lo[0], qy = (t / gp.T[0]) % gp.T[1] - sel.lo[1], qz = t / (gp.T[0] * gp.T[1]) - sel.lo[2];
      if (qx >= 0 && qx < sel.n[0] && qy >= 0 && qy < sel.n[1] && qz >= 0 && qz < sel.n[2]) return;      /* (searched by the interior launch) */
   }
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
   /* (few tiles -- a small system: the rows of a tile are shared by gridDim.y workgroups, every wave still works alone) */
   /* A batch is one chain of dependent round trips -- the rows' counts, then their pieces, then the LDS passes -- and a wave has nothing else to
    * do while it waits (round 6: a build of EMPTY rows took 385 of the kernel's 680 us at 4.24 M beads).  The memory part of the chain is
    * cut: the counts are asked for two batches ahead and the pieces one batch ahead, while the batch in hand is sorted (-5 %) */
   const int wstep = TR_WROWS * (TR_THREADS / 64) * (int)gridDim.y;
   auto row_cnt = [&](const int r0_) -> int { const int row_ = r0_ + rl; return (r0_ < rows && row_ < nown) ? ta.nbr_cnt[ts + row_] : 0; };
   auto row_load = [&](uint4 (&v)[NQ], const int r0_, const int cnt_)
   {
      /* the row's 16-byte pieces: piece p of row l of a chunk at (p TB_CHUNK + l) 16 bytes (TileArgs::tmp32) -- eight rows side by side
       * are 128 contiguous bytes per piece */
      const int nq_ = (cnt_ + EPQ - 1) / EPQ;
      const int rowc = max(min(r0_ + rl, nown - 1), 0);
      const uint4 *src = (const uint4 *)((const char *)ta.tmp32 + ((size_t)ts + (size_t)TB_CHUNK * t + (size_t)(rowc & ~(TB_CHUNK - 1))) * ta.tmpw * (SCR16 ? 2 : 4)) + (rowc & (TB_CHUNK - 1));      /* tmpw is a multiple of 8 */
#pragma unroll
      for (int j = 0; j < NQ; j++) v[j] = (q + 8 * j < nq_) ? src[(size_t)(q + 8 * j) * TB_CHUNK] : make_uint4(0, 0, 0, 0);
   };
   int r0 = (w + (TR_THREADS / 64) * (int)blockIdx.y) * TR_WROWS;
   int cnt_cur = row_cnt(r0), cnt_nxt = row_cnt(r0 + wstep);
   uint4 wv[NQ];
   row_load(wv, r0, cnt_cur);
   for (; r0 < rows; r0 += wstep)
   {
      const int row = r0 + rl;
      const int cnt = cnt_cur;
      uint4 wn[NQ];
      row_load(wn, r0 + wstep, cnt_nxt);      /* (the next batch's pieces: in flight while this one is sorted) */
      const int cnt_n2 = row_cnt(r0 + 2 * wstep);
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
#pragma unroll
      for (int j = 0; j < NQ; j++) wv[j] = wn[j];
      cnt_cur = cnt_nxt; cnt_nxt = cnt_n2;
   }
}

