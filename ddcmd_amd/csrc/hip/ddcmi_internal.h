/* ddcmi_internal.h -- private declarations shared by the .hip translation units. */
#ifndef DDCMI_INTERNAL_H
#define DDCMI_INTERNAL_H
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <time.h>
#include <string.h>
#include <string>
#include <vector>
#include "ddcmi.h"
#include "ddcmi_test.h"      /* the test-only entry points: compiled into the objects, exported by libddcmi_test.so only */

#define DDCMI_BLOCK 256
/* d_flags slots: [0,64) the rebuild's flags and the arena counter ([32,34), a cache line of its own), posted to the host as one block;
 * [64] the ticket of the mailbox post (scan.hip: k_post), [96] / [97] the in / out word of the rebuild's error agreement */
#define DDCMI_NFLAGS 128
#define DDCMI_FLAG_TICKET 64
#define DDCMI_FLAG_AGREE 96
#define DDCMI_RETRY_IMAGES 1001        /* internal: ddcmi_bl_finish found more periodic images than the rebuild was launched for */

/* tile = TCX x TCY x TCZ cells (8x4x4 of >= (rmax+deltaR)/2 wide cells: ~500 beads, a
 * 64 x 32 x 32 A slab); its neighbourhood = the tile plus two cells on every side */
#ifndef TCX
#define TCX 8
#endif
#ifndef TCY
#define TCY 4
#endif
#ifndef TCZ
#define TCZ 4
#endif
#define TCELLS (TCX * TCY * TCZ)
#define RGX (TCX + 4)
#define RGY (TCY + 4)
#define RGZ (TCZ + 4)
#define NRC (RGX * RGY * RGZ)
#ifndef NB_THREADS
#define NB_THREADS 512       /* k_nonbond workgroup: 8 waves per tile */
#endif
#ifndef NB_WPE
#define NB_WPE 4             /* waves per SIMD the register budget is sized for (2 workgroups per CU) */
#endif
#ifndef NB_SU
#define NB_SU 4              /* staging gathers in flight per thread and batch (divides 8) */
#endif
#ifndef NB_CH
#define NB_CH 4              /* list slots gathered and tested together */
#endif
#define NB_ZOFF 54400        /* k_nonbond's fixed LDS layout: byte offset of the staged z array (3400 beads x 16 B of {x,y} in front of it) */

/* cell grid over the local domain plus a margin of image/halo cells.
 * Cells are numbered tile-major (4x4x4 cells per tile) so that 256 consecutive
 * sorted atoms form a compact region instead of a long row. */
struct GridParams
{
   double lo[3];     /* lower corner of the owned domain */
   double L[3];      /* global box lengths */
   double cinv[3];   /* 1/cell size */
   double rlist;
   int n[3];         /* interior cells */
   int m[3];         /* margin cells per side (0 on non-periodic, undivided axes) */
   int g[3];         /* n + 2m */
   int T[3];         /* tiles per axis */
   int pbc;
   int ncell;        /* T0*T1*T2*TCELLS */
};

/* segments of a halo send/recv buffer in BUFFER order: seg q holds direction code[q] and
 * starts at off[q].  Remote segments are ordered by (peer rank, direction code), so everything
 * for one peer is contiguous and travels as ONE message per step. */
struct SegTab { int off[28]; signed char code[28]; int nseg; };
/* the messages of the per-step halo exchange: one per peer and direction of travel (send / receive), in beads */
struct HaloMsgs { int n; int peer[27], off[27], cnt[27]; };

/* DDCMI_DEBUG_GUARD=1 (debugging aid): device buffers get exactly the requested size plus a 256-byte canary that
 * is verified when the buffer is grown or released -- a write beyond a buffer aborts with a message instead of
 * landing in the slack the normal sizing leaves; =2 also fills fresh buffers with 0xFF bytes */
static inline int ddcmi_debug_guard()
{
   static int g = -1;
   if (g < 0) { const char *e = getenv("DDCMI_DEBUG_GUARD"); g = e ? atoi(e) : 0; }
   return g;
}
template <class T> struct dbuf
{
   T *p = nullptr;
   size_t cap = 0;
   static size_t guard() { return ddcmi_debug_guard() ? (256 + sizeof(T) - 1) / sizeof(T) : 0; }
   void check() const
   {
      if (!p || !ddcmi_debug_guard()) return;
      unsigned char h[512];
      const size_t nb = guard() * sizeof(T);
      if (hipMemcpy(h, p + cap, nb, hipMemcpyDeviceToHost) != hipSuccess) return;
      for (size_t k = 0; k < nb; k++)
         if (h[k] != 0xA5) { fprintf(stderr, "ddcmi: write beyond a device buffer of %zu elements of %zu bytes (byte %zu of the canary)\n", cap, sizeof(T), k); abort(); }
   }
   int ensure(size_t n, bool keep = false, hipStream_t s = 0)
   {
      if (n <= cap) return 0;
      size_t ncap = ddcmi_debug_guard() ? n : n + n / 8 + 64;
      T *q = nullptr;
      if (hipMalloc((void **)&q, (ncap + guard()) * sizeof(T)) != hipSuccess) return -1;
      if (guard() && hipMemset(q + ncap, 0xA5, guard() * sizeof(T)) != hipSuccess) return -1;
      if (ddcmi_debug_guard() >= 2 && hipMemset(q, 0xFF, ncap * sizeof(T)) != hipSuccess) return -1;      /* 2: poison fresh buffers (NaN / -1): reads of never-written elements show */
      if (ddcmi_debug_guard()) (void)hipDeviceSynchronize();      /* null-stream memsets are not ordered with the contexts' non-blocking streams */
      if (keep && p && cap) { if (hipMemcpyAsync(q, p, cap * sizeof(T), hipMemcpyDeviceToDevice, s) != hipSuccess) return -1; (void)hipStreamSynchronize(s); }
      if (p) { check(); (void)hipFree(p); }
      p = q; cap = ncap;
      return 0;
   }
   void release() { if (p) { check(); (void)hipFree(p); } p = nullptr; cap = 0; }
};

/* which tiles a k_tile_build launch searches.  mode 0: all (block b = tile b).  mode 1: the box of tile coordinates [lo, lo + n) --
 * the INTERIOR tiles, whose 12x8x8-cell neighbourhoods hold owned cells only: their search needs nothing of the halo and runs on a
 * second stream while the rebuild exchanges, assembles and sorts the halo (block b = b-th tile of the box).  mode 2: all tiles
 * except that box (the workgroups of its tiles return at once). */
struct TileSel { int mode; int lo[3], n[3]; };

/* LDS layout of one k_nonbond workgroup (nb_lds_layout, ddcmi_step.inl) */
#define LEAN_W 32
#define LEAN_HW 32             /* doubles of a lean step's row of sums: 8 pair sums, 7 kinetic sums, 0, the 10 bonded sums, 0 ... */
#define LEAN_VSTRIDE 32      /* words: 128 B */
struct NbLds { size_t total; int tab_off, ke_off; bool lvl, zfix; int wgs; };

/* results block on the device / pinned host mirror */
enum
{
   R_NB_LJ = 0, R_NB_ELE = 1, R_NB_VIR = 2,        /* raw full-list sums (x2) */
   R_SCR_BOND = 8,                                 /* bonded kernels' reduced sums: {e, vir6} */
   R_SCR_ANGLE = 16,                               /* {e, vir6} */
   R_SCR_TORS = 24,                                /* {e_tors, e_impr, vir6} */
   R_SCR_REST = 32,                                /* RESTRAINT potential: {e, vir6} */
   R_SCR_MOLV = 40,                                /* molecularVirial correction: sum (r - R_mol) f, xx yy zz */
   R_RK = 56, R_TION = 57,
   R_E = 64,                                       /* final energies[DDCMI_NE] */
   R_VIR = 72,                                     /* final virial[6] */
   R_GROUP = 80,                                   /* per-group rk, count pairs */
   R_DISP = 152,                                   /* displacement bound of the shell-limited walk: sum over the steps since the rebuild of max_i |dt v_i| */
   R_SIZE = 160
};

struct ddcmi_ctx
{
   int device = 0;
   hipStream_t stream = 0;
   std::string err;
   /* parameters */
   double h[9] = {0}; int pbc = 7; bool have_box = false;
   int nspecies = 0; std::vector<double> mass, charge; std::vector<int> ljtype, moltype;
   int nlj = 0; std::vector<double> sigma, eps, shift; double rmax = 0, keR = 0, krf = 0, crf = 0;
   int nmoltype = 0; std::vector<int> mol_nspecies, bpair_off, bpairI, bpairJ;
   double deltaR = 0; int updateRate = 0;
   int ngroup = 1; std::vector<int> gtype, ginterval; std::vector<double> gTeq, gtau;
   std::vector<double> glambda, gTsum, gT; std::vector<int> gnT, gdoScaling;
   std::vector<double> gvcm;           /* LANGEVIN groups: drift velocity [3 ngroup] (ddcmi_set_group_vcm; empty = zero) */
   int64_t loop = 0; double time = 0;
   int excludePotentialTerm = 0;
   bool has_charge = false;
   /* device tables */
   dbuf<double> d_invmass, d_mass, d_charge_sp; dbuf<int> d_ljtype_sp, d_moltype_sp;      /* d_ljtype_sp: (LJ type, charge) class of each species */
   dbuf<double> d_kqtab; int nnb = 0; bool tables_dirty = true;                           /* ke/eps_r q_a q_b per class pair; classes; rebuild tables */
   dbuf<double4> d_ljtab;          /* nlj*nlj {sigma^2, 4eps, shift, 24eps} */
   /* the same table in two levels (round 5: the type-count cliff): Martini's nspecies^2 table (bioMartini.c:868-950) holds few DISTINCT
    * entries -- a dozen interaction levels x two or three sigmas, x the charge products -- so the pair kernel can keep one byte per class
    * pair (d_lvlidx [nnb*nnb]) and the distinct entries (d_lvltab [nlvl] {sigma^2, 4eps, shift, kq or 24eps}) in LDS instead of
    * 32 bytes per class pair: 40 classes cost 1.6 KB + the levels instead of 51 KB.  Used when the direct table would cost the pair
    * kernel its second workgroup per CU (nb_lds_bytes); force_lvl: DDCMI_FORCE_LEVEL_TABLE=1 (tests) */
   dbuf<double4> d_lvltab; dbuf<unsigned char> d_lvlidx; int nlvl = 0; bool force_lvl = false;
   dbuf<int> d_mol_nspecies, d_bpair_off, d_bpairI, d_bpairJ;
   dbuf<unsigned long long> d_exmask;   /* [nmoltype][64] bonded-pair masks by atom code (list build) */
   /* particle state: [0,nloc) owned, [nloc,nloc+nhalo) images/halo */
   int nloc = 0, nhalo = 0, npad = 0;
   dbuf<double4> pos, pos2;
   dbuf<double> vx, vy, vz, vx2, vy2, vz2, fx, fy, fz;
   dbuf<int> species, species2, group, group2, orig, orig2, slot_of_orig;
   dbuf<uint64_t> gid, gid2;
   /* sort / cells */
   GridParams gp;
   dbuf<int> cid, crank, order, cell_cnt_o, cell_start_o, cell_cnt_h, cell_start_h, cell_start, cell_cnt;
   dbuf<int> nimg, img_off, hsrc_t, hshift_t, hcid, hrank, horder, halo_src, halo_shift;
   dbuf<int> scan_tmp;
   /* lists */
   int maxnbr = 0, maxexcl = 0;
   dbuf<int> nbr_cnt, excl, excl_cnt;
   dbuf<uint64_t> hkey; bool hkey_valid = false;      /* gid of every halo descriptor (k_halo_assemble): the key of the halo's in-cell order */
   dbuf<int> tile_work, sched, tile_perm;   /* per-tile cost estimate (bit 30: stages halo beads); XCD ranges [2][16]; tile order */
   int nitems = 0;                     /* work items of k_nonbond: the tiles with owned beads, the last ones of each XCD's run cut into parts */
   int sched_cache[2][8][3] = {};      /* tail split of each class and XCD run: {tiles it was found for, tiles cut, parts} */
   int sched_longest[2] = {0, 0}, ntile_class[2] = {0, 0};      /* class 0: all-owned neighbourhoods (all tiles on one domain), class 1: the rest */
   hipStream_t stream_post = nullptr;  /* the post of the build's results to the host, beside the transposition */
   hipStream_t stream2 = nullptr;      /* decomposed runs: halo exchange, concurrent with the class-0 tiles */
   hipEvent_t ev_drift = nullptr, ev_halo = nullptr, ev_build = nullptr;
   /* rebuild: the interior tiles' search runs on stream2 behind the owned sort (bl_launch_interior); interior_key = the capacities it was launched with */
   hipEvent_t ev_sorted = nullptr, ev_interior = nullptr; bool interior_launched = false; TileSel interior_sel; long long interior_key[4] = {0, 0, 0, 0};
   bool halo_overlap = false;          /* DDCMI_HALO_OVERLAP=1: exchange on stream2 under the all-owned tiles */
   /* tiles (4x4x4 cells): staging lists + 16-bit ELL arena */
   int stage_cap_want = 0;             /* the staging capacity the last build's largest neighbourhood asks for (+1.5 % + 40 beads): taken at the next rebuild if smaller -- every staged bead of slack is 24 B of k_nonbond's LDS */
   int ntile = 0, stage_cap = 0; int pack_type = 0;      /* 0 bare slots, 1 slot<<4|type, 2 + shift bit (see TileArgs) */
   dbuf<int> stage_idx, tile_nstage, tile_width, tile_rows;
   dbuf<long long> tile_base;
   dbuf<unsigned char> tile_nib;       /* packed entries: type nibble of every staged slot of every tile (k_tile_build -> k_tile_transpose) */
   dbuf<unsigned short> nbr16, excl16; dbuf<unsigned int> tmp32; int tmpw = 0;      /* excl16: the excluded pairs as staged-slot entries (k_nonbond) */
   unsigned long long arena_cap = 0;
   dbuf<double> kpartials;             /* per-workgroup kinetic terms (k_kick_ke) */
   dbuf<double> red_tmp;               /* k_reduce_jobs: per-workgroup rows of a split job + ticket counters */
   dbuf<double4> pos0; dbuf<double> disp;   /* updateRate == 0: positions at the last rebuild; [0..2] sum of r-r0, [4] max |dr|^2 */
   double baro_T = 0, baro_P0 = 0, baro_beta = 0, baro_tau = 0;      /* NGLFCONSTRAINT's Berendsen barostat; beta = 0: off */
   bool baro_iso = false;                                           /* one scale factor from the mean of the three pressures (changeVolumeGPUisotropic) */
   double pmol[3] = {0, 0, 0};                                      /* molecular pressure (xx, yy, zz) the barostat last acted on */
   dbuf<ulonglong2> lcg, lcg2; bool lcg_on = false; /* Langevin groups, RANDOM type LCG64: LCG64_PARM {state; multID | prime << 32} of the owned beads in slot order; off = the counter-based stream */
   dbuf<uint4> nbr_cum;                /* [bead] entries in shells 0..s as eight 16-bit counts (k_tile_transpose) */
   bool shell_skip = false, no_shell_skip = false; double sh_r0sq = 0, sh_step = 0;      /* k_nonbond may end its rows at the last shell that can matter (NbTileArgs::disp); DDCMI_NO_SHELL_SKIP */
   int nhalo_hint = 0; const int *nhalo_dev = nullptr; int debug_image_bound = 0;      /* single-domain rebuilds after the first: the image count stays on the device until the build's post (bl_self_images) */
   bool pack_fresh = false;            /* decomposed runs: the halo send buffer already holds the current positions (packed in the fused step's reduction launch) */
   /* the lean step (round 5, step_post): a single domain of FREE beads without bonded terms runs ONE launch per step between rebuilds --
    * the pair kernel with the integrator's pass, which also stages the periodic images from their owners and keeps the displacement bound;
    * the second stage of its energy / virial / kinetic sums waits in a ring of per-step rows (lean_part, lean_kpart: LEAN_W steps of
    * lean_stride doubles) and is formed for all pending steps by one launch (lean_flush -> lean_hist: 16 sums per step) at the next rebuild,
    * when the ring is full or when the host asks.  DDCMI_NO_LEAN_STEP=1: the reduction launch after every step, as before. */
   int lean_pending = 0, lean_hist_n = 0, lean_since = 0 /* lean steps since the rebuild: the next one's word of the ring */; size_t lean_stride = 0; bool no_lean = false, no_self_img = false; double lean_dt = 0;
   dbuf<double> lean_part, lean_kpart, lean_hist, lean_tmp, lean_bpart; size_t lean_bstride = 0; int lean_bpstride = 0, lean_bnblk = 0;
   dbuf<unsigned> d_vring;             /* largest |v|^2 (float bits) of each lean step since the rebuild, one word per step at a stride of LEAN_VSTRIDE words: the step that
                                          is being written (atomic maxima of every workgroup) shares no cache line with the words the same launch reads */
   bool images_fresh = false;          /* the list was rebuilt in front of this force evaluation: the periodic images of a single domain need no update */
   int64_t fuse_tags_of = -1;              /* the rebuild whose halo tag words the second position buffer holds (fused steps swap the buffers) */
   uint64_t rng_seed = 0;              /* Langevin groups: seed of the counter-based normal stream (RANDOM seed) */
   bool slot_valid = false;            /* slot_of_orig (caller index -> device slot) belongs to the current order: refilled by the sort of a rebuild only when something
                                          names beads by caller index (bonded terms, constraint groups, molecule lists), else on demand (ddcmi_ensure_slots) */
   bool drift_done = false;            /* the FRONT kick + drift of the coming step ran fused with the last step's BACK kick */
   bool list_valid = false;
   int64_t nrebuild = 0, list_entries = 0, excl_entries = 0;
   int64_t mg_rebuilds = 0;            /* decomposed runs: rebuilds entered (the same number on every rank) */
   /* bonded */
   int nbond = 0, nangle = 0, ntors = 0;          /* term counts (of the whole system in a decomposed run) */
   /* RESTRAINT potential: restraints by gid; rest_slot = owned device slot of each (or -1), found at rebuilds */
   int nrest = 0, rest_origin = 0;
   /* nglfconstraint (one domain): constraint groups over caller-order atoms; pairs name group-local atoms */
   int ncgroup = 0, ncpair = 0, cons_maxA = 0, cons_maxP = 0;
   dbuf<int> cg_atom_off, cg_atoms, cg_pair_off, cons_status; dbuf<unsigned char> cg_pa, cg_pb; dbuf<double> cg_dist;
   /* decomposed runs: constraint groups and molecules named by gid (ddcmi_set_constraints_gid / ddcmi_set_molecule_lists_gid);
    * cg_slot / mol_slot = device slot of every listed atom, refilled at rebuilds (INT_MAX: not on this rank) */
   bool cons_gid = false, mol_gid = false;
   int cg_natom = 0, mol_natom = 0;
   dbuf<uint64_t> cg_atom_gid, mol_atom_gid; dbuf<int> cg_slot, mol_slot;
   /* molecules with atoms on several ranks ("split"): their centre of mass and total force need the ranks' partial sums.
    * mol_info[4m..] = {owning ranks, anchor x y z} all-reduced at rebuilds (the anchor = where the molecule's first listed
    * atom was then: the common reference for nearest images); mol_split[m] = index among the split ones or -1;
    * mol_red[6k..] = this rank's {sum m x, sum f} of split molecule k, all-reduced every step the barostat acts */
   int nsplit = 0;
   dbuf<double> mol_mtot, mol_info, mol_red; dbuf<int> mol_split;
   double baro_sums[12] = {0};         /* this step's {virial xx yy zz, molecular term xx yy zz, beads of multi... } before / after the all-reduce */
   /* molecules of more than one bead (molecular virial of the barostat), caller-order atoms */
   long nmol_total = 0; int nmol_multi = 0; bool molv_valid = false;   /* R_SCR_MOLV belongs to the forces now in fx */
   dbuf<int> mol_off, mol_atoms;
   dbuf<uint64_t> rest_gid; dbuf<int> rest_fc, rest_slot; dbuf<double> rest_r0, rest_kb;
   /* rows of the bead-parallel bonded kernel (atom -> its terms), built once in ddcmi_set_bonded[_gid] */
   int idx_amax_cons = -1, idx_amax_mol = -1;      /* largest caller index the index-named constraint groups / molecule lists use: checked against the bead count at the rebuild */
   int inc_nrow = 0, inc_heavy = 0, inc_light = 0, inc_lanes = 0, inc_hlanes = 0; dbuf<int> inc_boff, inc_aoff, inc_haoff, inc_toff, inc_brow, inc_arow, inc_harow, inc_trow, inc_hatoms, inc_latoms, inc_ldesc, inc_hdesc, inc_tab, inc_htab; int inc_tab_pieces[2] = {0, 0}, inc_tab_off[2][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}}; dbuf<double> inc_bpar, inc_apar, inc_tpar;
   bool bonded_gid = false;
   int natom_g = 0; dbuf<uint64_t> atom_gid;      /* sorted gids of the atoms that occur in terms: an atom's number is its place here */
   dbuf<int> slot_of_atom;                         /* [natom_g] lowest slot holding that gid on this rank, INT_MAX if absent (refilled at rebuilds) */
   dbuf<unsigned long long> hkeys; dbuf<int> hvals; unsigned hmask = 0;
   /* reductions */
   dbuf<double> partials, bpartials; int npartial_blocks = 0;
   double *d_results = nullptr; double *h_results = nullptr;
   int *d_flags = nullptr; int *h_flags = nullptr;      /* d_flags: DDCMI_NFLAGS ints, see the DDCMI_FLAG_* slots */
   /* decomposed runs, RCCL transport: the outcome of a rebuild's local phase (mg_phase4_finish) is agreed on by an all-reduce that
    * nobody waits for -- its result lands in agree_h (mapped host memory, [0] = sequence word, [1] = worst error code) and is
    * looked at in front of the next host wait (ddcmi_agree_poll) */
   int *agree_h = nullptr, *agree_d = nullptr; int agree_seq = 0; bool agree_pending = false; int64_t agree_loop = 0;
   /* growable pinned host staging (so that small copies are truly asynchronous): [0] tile work, [1] tile order, [2] count exchange */
   /* the mailbox (scan.hip: ddcmi_post / ddcmi_post_wait): mapped coherent host memory, [0] = sequence word */
   int *mbox_h = nullptr, *mbox_d = nullptr; size_t mbox_cap = 0; int mbox_seq = 0;
   int *h_pin[3] = {nullptr, nullptr, nullptr}; size_t h_pin_cap[3] = {0, 0, 0};
   int *pinned(int which, size_t n)
   {
      if (n <= h_pin_cap[which]) return h_pin[which];
      if (h_pin[which]) (void)hipHostFree(h_pin[which]);
      h_pin[which] = nullptr; h_pin_cap[which] = 0;
      size_t cap = n + n / 4 + 64;
      if (hipHostMalloc((void **)&h_pin[which], cap * sizeof(int), hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess) return nullptr;
      h_pin_cap[which] = cap;
      return h_pin[which];
   }
   double self_ele = 0.0; std::vector<long> sp_count;      /* sp_count: beads per species of the last ddcmi_upload_state (one domain: the self term follows new charges / constants without the beads) */
   /* what the last rebuild's tail left zeroed for this one (k_rebuild_tail): the cell counters (for this cell count, in these buffers),
    * the flags, the arena counter, the direction counters; a rebuild that cannot rely on it clears them itself */
   bool counters_clean = false, dircnt_clean = false, dir28_clean = false; int clean_ncell = 0; const int *clean_po = nullptr, *clean_ph = nullptr;
   bool sort_renumbers = false;        /* decomposed rebuild: the coming sort numbers the beads (orig = place in front of the sort) in its first kernel */
   bool forces_valid = false;
   /* the bonded kernels' (and restraints') force on every owned bead, ONE 32-byte record per bead: the bead-parallel kernels touch a
    * scattered bead once (a store) instead of six times (read-modify-write of three arrays), the pair kernel reads it with one load
    * and hands it back zeroed -- every pair launch does, so the array is all zero whenever the bonded kernels start */
   dbuf<double4> fb; size_t fb_zeroed = 0;      /* fb_zeroed: elements known to be zero (a grown buffer is cleared once) */
   /* timing */
   /* (rounds 3-5 could replay the steady step as a hipGraph, DDCMI_GRAPH_MAX_BEADS: slower than plain launches on ROCm 7.2 at every size measured
    * -- 6.9 k beads 86 vs 74 us -- and since the lean step the steady step of such systems IS one launch: removed in round 6) */
   bool timing = false; std::vector<hipEvent_t> ev; size_t ev_used = 0; int64_t t_launches = 0; double t_ms = 0;
   std::vector<char> ev_fused; int64_t t_launches_fused = 0; double t_ms_fused = 0, t_last_fused[2] = {0, 0};      /* of those: launches whose epilogue was the integrator's pass */
   /* DDCMI_DEBUG_PHASES=1: host wall time between the marks of a rebuild (where the host waits, where it is busy), printed at ddcmi_destroy */
   int ph_on = -1; double ph_last = 0, ph_sum[32] = {0}; long ph_cnt[32] = {0}; const char *ph_name[32] = {nullptr};
   void phase(int k, const char *name)
   {
      if (ph_on < 0) ph_on = getenv("DDCMI_DEBUG_PHASES") ? 1 : 0;
      if (!ph_on) return;
      struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts);
      const double now = ts.tv_sec * 1e6 + ts.tv_nsec * 1e-3;
      if (k >= 0 && k < 32) { ph_sum[k] += now - ph_last; ph_cnt[k]++; ph_name[k] = name; }
      ph_last = now;
   }
   /* comm */
   int rank = 0, nranks = 1; void *comm = nullptr; int pgrid[3] = {1, 1, 1}, pcoord[3] = {0, 0, 0};
   ddcmi_rdzv *hcomm = nullptr;        /* host transport (ddcmi_comm_init_host): messages staged through the host, carried by TCP streams */
   std::vector<double> hstage_s, hstage_r;      /* its staging */
   HaloMsgs hmsg_s, hmsg_r;            /* per-step halo messages (mg_layout_halo) */
   struct ddcmi_group *group_ = nullptr;        /* in-process multi-domain emulation (tests) */
   bool halo_fresh = false;
   /* Decomposed runs over a transport whose halo holds received beads only (2x2x2 bricks, the loopback; no undivided periodic axis) and
    * whose forces need the halo in the pair kernel only (no bonded terms, no constraint groups): the per-step exchange leaves the
    * neighbours' positions in the receive buffer and k_nonbond stages them from THERE through halo_src -- no k_halo_update launch in
    * the steady step (VERDICT r4 #1a).  halo_in_recv: the current positions of the received beads live in hrecv3, pos[nloc..] holds
    * their records as of the last rebuild (tags valid, x y z stale).  DDCMI_NO_DIRECT_HALO=1 keeps the update launch. */
   int nself_images = 0; bool halo_in_recv = false, no_direct_halo = false;
   SegTab sseg, rseg;                  /* halo send / receive buffer layout (peer-major) */
   bool loopback = false;              /* one rank whose periodic neighbours are reached through RCCL (test facility, DDCMI_RCCL_LOOPBACK=1) */
   int dir_dest[27], dir_shift[27][3];            /* 26 neighbour directions, code = (dx+1)+3(dy+1)+9(dz+1) */
   int hs_cap = 0, mig_cap = 0;
   dbuf<int> hs_idx, dir_cnt;                      /* halo send lists per direction */
   dbuf<unsigned> send_map;                        /* send slot -> owned bead | direction << 27 (flattened at rebuilds) */
   int hs_cnt[27], hr_cnt[27], send_off[28], recv_off[28], nsend = 0, nrecv = 0;
   int mig_scnt[27], mig_rcnt[27];
   dbuf<double> sendbuf, hrecv3, hrecv5, mig_out, mig_in;
   dbuf<int> keep, cnt_xchg;
};

#define SETERR(ctx, code, ...) do { char _b[512]; snprintf(_b, sizeof(_b), __VA_ARGS__); (ctx)->err = _b; return (code); } while (0)
/* an entry point's argument check: a NULL context has nowhere to leave a message; anything else says what was wrong (tools/fuzz_abi.py, round 6) */
#define ARGCHK(ctx, bad, ...) do { if (!(ctx)) return DDCMI_EINVAL; if (bad) SETERR(ctx, DDCMI_EINVAL, __VA_ARGS__); } while (0)
#define HIPCHK(ctx, call) do { hipError_t _e = (call); if (_e != hipSuccess) { SETERR(ctx, DDCMI_ENODEVICE, "%s failed: %s (%s:%d)", #call, hipGetErrorString(_e), __FILE__, __LINE__); } } while (0)
#define ENSURE(ctx, buf, n) do { if ((buf).ensure((n)) != 0) SETERR(ctx, DDCMI_ENOMEM, "device allocation of %zu elements failed (%s:%d)", (size_t)(n), __FILE__, __LINE__); } while (0)

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

/* roctx ranges (ddcmi.hip has the story): RoctxRange r("MDSTEP"); -- nothing unless DDCMI_ROCTX=1 */
#include <dlfcn.h>
#include <atomic>
extern std::atomic<long> g_roctx_ranges;
struct RoctxRange
{
   typedef int (*push_fn)(const char *);
   typedef int (*pop_fn)(void);
   static int state() { static int s = -1; if (s < 0) { const char *e = getenv("DDCMI_ROCTX"); s = (e && atoi(e) != 0) ? 1 : 0; } return s; }
   static bool bind(push_fn *pu, pop_fn *po)
   {
      static push_fn push = nullptr; static pop_fn pop = nullptr; static bool tried = false;
      if (!tried)
      {
         tried = true;
         for (const char *lib : {"librocprofiler-sdk-roctx.so", "librocprofiler-sdk-roctx.so.1", "libroctx64.so", "libroctx64.so.4"})
         {
            void *h = dlopen(lib, RTLD_NOW | RTLD_GLOBAL);
            if (!h) continue;
            push = (push_fn)dlsym(h, "roctxRangePushA"); pop = (pop_fn)dlsym(h, "roctxRangePop");
            if (push && pop) break;
            push = nullptr; pop = nullptr;
         }
      }
      *pu = push; *po = pop;
      return push != nullptr;
   }
   bool on = false;
   explicit RoctxRange(const char *name)
   {
      if (!state()) return;
      push_fn pu; pop_fn po;
      g_roctx_ranges++;      /* (counted whether or not a marker library is there: tests) */
      if (bind(&pu, &po)) { pu(name); on = true; }
   }
   ~RoctxRange() { if (on) { push_fn pu; pop_fn po; bind(&pu, &po); po(); } }
};


#ifdef __HIPCC__
/* double-precision 1/sqrt(x) and 1/x from the single-precision hardware seeds (shared by the pair and the bonded kernels) */
__device__ __forceinline__ double rsqrt_f64(double x)
{
   /* v_rsq_f32 seed (23 bits) + two Newton steps y += y*(1/2 - (x/2) y^2): 3 FP64 ops
    * each, < 2 ulp; checked against the closed form in tests */
   float xf = (float)x;
   double y = (double)__builtin_amdgcn_rsqf(xf);      /* the bare instruction: __frsqrt_rn expands to a correctly rounded sequence */
   double h = 0.5 * x;
   double e = fma(-(h * y), y, 0.5);
   y = fma(y, e, y);
   e = fma(-(h * y), y, 0.5);
   y = fma(y, e, y);
   return y;
}
/* the pair kernel's versions: the double-precision hardware seeds (v_rsq_f64 / v_rcp_f64: not full precision by themselves -- bare, they
 * fail the force parity by 1e-8 -- but good for one Newton step instead of two, without the two conversions: k_nonbond -2 % at 4 M
 * water, -3 % on the lipid box).  The bonded kernels keep the versions above: their branch census (tests/test_gpu_branches.py) is
 * pinned to that rounding. */
__device__ __forceinline__ double rsqrt_f64_pair(double x)
{
   double y = __builtin_amdgcn_rsq(x);
   const double h = 0.5 * x;
   const double e = fma(-(h * y), y, 0.5);
   return fma(y, e, y);
}
__device__ __forceinline__ double rcp_f64_pair(double x)
{
   double y = __builtin_amdgcn_rcp(x);
   const double e = fma(-x, y, 1.0);
   return fma(y, e, y);
}
/* 1/x: v_rcp_f32 seed + two Newton steps y += y*(1 - x y): 2 FP64 ops each.  Used
 * when no bead carries a charge: Lennard-Jones needs 1/r^2 only, no square root. */
__device__ __forceinline__ double rcp_f64(double x)
{
   float xf = (float)x;
   double y = (double)__builtin_amdgcn_rcpf(xf);      /* the bare instruction: __frcp_rn expands to a 10-instruction IEEE division */
   double e = fma(-x, y, 1.0);
   y = fma(y, e, y);
   e = fma(-x, y, 1.0);
   y = fma(y, e, y);
   return y;
}
#endif

/* scan.hip */
int ddcmi_scan_exclusive(ddcmi_ctx *ctx, const int *src, int *dst, int n, int *d_total);
struct ZeroJobs
{
   int *p[12]; int n[12]; int cnt = 0;
   ZeroJobs &add(void *ptr, size_t nints) { p[cnt] = (int *)ptr; n[cnt] = (int)nints; cnt++; return *this; }
};
int ddcmi_zero_ints(ddcmi_ctx *ctx, hipStream_t st, const ZeroJobs &z);
struct PostJobs
{
   const int *src[6]; int n[6]; int off[6]; int cnt = 0;
   PostJobs &add(const void *ptr, size_t nints) { src[cnt] = (const int *)ptr; n[cnt] = (int)nints; off[cnt] = 0; cnt++; return *this; }
};
int ddcmi_post(ddcmi_ctx *ctx, hipStream_t st, PostJobs &j);            /* device arrays -> the mailbox; fills j.off */
int ddcmi_post_wait(ddcmi_ctx *ctx, hipStream_t st);                    /* spin until the post has landed */
int ddcmi_agree_poll(ddcmi_ctx *ctx);                                   /* in front of a host wait: has a peer reported a failed rebuild? (ddcmi_multigpu.inl) */
int ddcmi_fetch(ddcmi_ctx *ctx, hipStream_t st, int *dst, const int *src_host_mapped, int n);
struct TailJobs
{
   int *fdst[2]; const int *fsrc[2]; int fn[2]; int nfetch = 0;      /* fsrc: mapped host memory (host addresses in, device addresses at the launch) */
   ZeroJobs zero;
   TailJobs &fetch(int *dst, const int *src_host_mapped, size_t n) { fdst[nfetch] = dst; fsrc[nfetch] = src_host_mapped; fn[nfetch] = (int)n; nfetch++; return *this; }
};
int ddcmi_rebuild_tail(ddcmi_ctx *ctx, hipStream_t st, TailJobs &j);
int ddcmi_bonded_localize(ddcmi_ctx *ctx);
int ddcmi_ensure_slots(ddcmi_ctx *ctx);
int ddcmi_group_ke_sums(ddcmi_ctx *ctx);
int ddcmi_displacement_check(ddcmi_ctx *ctx, int *need);
/* bonded.hip */
int ddcmi_launch_bonded(ddcmi_ctx *ctx, double4 *fb = nullptr, int lean_slot = -1);
int ddcmi_lean_flush_bonded(ddcmi_ctx *ctx, int np);      /* fb: the force goes to the (zeroed) record array instead of being added to fx, fy, fz */
int ddcmi_launch_constraints(ddcmi_ctx *ctx, double dt, int location);   /* 0 FRONT, 1 BACK */
int ddcmi_launch_mol_virial(ddcmi_ctx *ctx);
int ddcmi_groups_localize(ddcmi_ctx *ctx);      /* rebuild: constraint groups / molecules named by gid -> device slots */
int ddcmi_mol_split_finish(ddcmi_ctx *ctx);     /* rebuild, after mol_info has been summed over the ranks */
int ddcmi_mol_split_term(ddcmi_ctx *ctx, double out[3]);      /* after mol_red has been summed: sum over split molecules of (P/M) o F */
/* comm.hip */
void ddcmi_comm_destroy(ddcmi_ctx *ctx);
/* ddcmi.hip: rebuild phases shared by the single- and multi-domain paths */
int ddcmi_bl_sort_owned(ddcmi_ctx *ctx);
int ddcmi_bl_reserve_halo(ddcmi_ctx *ctx, int nh);
int ddcmi_bl_halo_sort(ddcmi_ctx *ctx);
int ddcmi_bl_finish(ddcmi_ctx *ctx);
/* multi-domain path (ddcmi_multigpu.inl, compiled into ddcmi.hip) */
int ddcmi_mg_rebuild(ddcmi_ctx *ctx);
int ddcmi_mg_refresh_halo(ddcmi_ctx *ctx, hipStream_t st);

#endif
