/* ddcmi_integrator.inl -- reductions, final energies, the NGLF integrator kernels, kinetic terms, export / inspection kernels.
 * Part of the ONE translation unit ddcmi.hip (kernels, templates and the static helpers they share), included there in this order. */
/* zero forces (nonbonded excluded via excludePotentialTerm) */
__global__ void k_zero3(int n, double *a, double *b, double *c)
{
   int i = blockIdx.x * blockDim.x + threadIdx.x;
   if (i < n) { a[i] = 0; b[i] = 0; c[i] = 0; }
}

/* Fixed-order reductions of per-workgroup partials (8 doubles per row), one workgroup per
 * job: thread (g = tid/8, k = tid%8) adds column k of rows g, g+128, ...; a tree over the
 * 128 row groups finishes all columns at once.  Same order every run => bitwise
 * reproducible.  A step runs its two jobs -- the nonbonded sums (+ the final energies)
 * and the kinetic terms -- in one launch. */
struct RedJob { const double *partials; int nblocks; int nv; double *out; int finish;
                double disp_dt; double *disp; /* disp_dt > 0: column 7 holds max |v|^2 of the step's drift -- reduced by max, then *disp += disp_dt * sqrt(max) (NbTileArgs::disp) */ };
__device__ void finish_energy(double *r, double self_ele);
/* RED_SPLIT workgroups share a job (one workgroup reading the 0.5 MB of per-tile partials of
 * a 4 M-bead box took 21 us); each leaves its 8 sums in tmp, the last one to arrive (ticket)
 * adds the RED_SPLIT rows in index order.  tmp: [2 jobs][RED_SPLIT][8] doubles, then 2 ticket
 * counters (left at zero). */
#define RED_SPLIT 8
__device__ __forceinline__ void reduce_jobs_block(const RedJob &j, const int bx, const int by, double *results, double self_ele, double *tmp, const int njobs = 2)
{
   __shared__ double s[1024];
   __shared__ int s_last;
   const int k = threadIdx.x & 7, g = threadIdx.x >> 3;
   const bool mx = k == 7 && j.disp_dt > 0.0;      /* this thread's column is a maximum (of non-negative numbers) */
   double a = 0.0;
   if (mx)
   {
      for (int b = g + 128 * bx; b < j.nblocks; b += 128 * RED_SPLIT) a = fmax(a, j.partials[(size_t)b * 8 + 7]);
   }
   else if (k < j.nv)
   {
      /* independent partial sums: a single chain of dependent loads is latency-bound */
      double p[4] = {0, 0, 0, 0};
      int b = g + 128 * bx;
      const int stride = 128 * RED_SPLIT;
      for (; b + 3 * stride < j.nblocks; b += 4 * stride)
      {
#pragma unroll
         for (int u = 0; u < 4; u++) p[u] += j.partials[(size_t)(b + u * stride) * 8 + k];
      }
      for (int u = 0; b < j.nblocks; b += stride, u++) p[u] += j.partials[(size_t)b * 8 + k];
      a = (p[0] + p[1]) + (p[2] + p[3]);
   }
   s[threadIdx.x] = a;
   __syncthreads();
   for (int off = 512; off >= 8; off >>= 1)
   {
      if (threadIdx.x < off) s[threadIdx.x] = mx ? fmax(s[threadIdx.x], s[threadIdx.x + off]) : s[threadIdx.x] + s[threadIdx.x + off];
      __syncthreads();
   }
   double *mytmp = tmp + ((size_t)by * RED_SPLIT + bx) * 8;
   unsigned int *ticket = (unsigned int *)(tmp + (size_t)njobs * RED_SPLIT * 8) + by;
   if (threadIdx.x < 8) { mytmp[threadIdx.x] = s[threadIdx.x]; __threadfence(); }
   __syncthreads();
   if (threadIdx.x == 0)
   {
      unsigned int t = atomicAdd(ticket, 1u);
      s_last = (t == RED_SPLIT - 1);
      if (s_last) { *ticket = 0u; __threadfence(); }
   }
   __syncthreads();
   if (!s_last) return;
   if (threadIdx.x == 7 && j.disp_dt > 0.0)
   {
      const double *row = tmp + (size_t)by * RED_SPLIT * 8 + 7;
      double t = 0.0;
      __threadfence();
#pragma unroll
      for (int q = 0; q < RED_SPLIT; q++) t = fmax(t, row[q * 8]);
      *j.disp += j.disp_dt * sqrt(t) * (1.0 + 1e-7);      /* (rounded up: |v|^2 came as a float rounded up) */
   }
   if (threadIdx.x < (unsigned)j.nv)
   {
      const double *row = tmp + (size_t)by * RED_SPLIT * 8 + threadIdx.x;
      double t = 0.0;
      __threadfence();          /* acquire: the other workgroups' rows, written on other XCDs */
#pragma unroll
      for (int q = 0; q < RED_SPLIT; q++) t += row[q * 8];
      j.out[threadIdx.x] = t;
   }
   if (j.finish)
   {
      __syncthreads();          /* orders the out[] stores before thread 0 reads them */
      if (threadIdx.x == 0) finish_energy(results, self_ele);
   }
}
__global__ __launch_bounds__(1024) void k_reduce_jobs(RedJob j0, RedJob j1, double *results, double self_ele, double *tmp)
{
   reduce_jobs_block(blockIdx.y ? j1 : j0, (int)blockIdx.x, (int)blockIdx.y, results, self_ele, tmp);
}
/* the lean steps' sums, all pending steps in ONE launch (ddcmi_ctx::lean_pending): step q's rows lie q * stride doubles behind the first
 * step's; its eight pair sums go to hist[LEAN_HW q], its seven kinetic sums to hist[LEAN_HW q + 8] -- the same workgroups, the same order of
 * additions as the per-step launch (reduce_jobs_block), so the sums are bit for bit the ones that launch forms */
__global__ __launch_bounds__(1024) void k_reduce_hist(RedJob jf, RedJob jk, size_t stride, int nsteps, double *hist, double *tmp)
{
   const int by = (int)blockIdx.y, q = by >> 1;
   RedJob j = (by & 1) ? jk : jf;
   j.partials += (size_t)q * stride;
   j.out = hist + (size_t)LEAN_HW * q + ((by & 1) ? 8 : 0);
   j.finish = 0; j.disp_dt = 0.0; j.disp = nullptr;
   reduce_jobs_block(j, (int)blockIdx.x, by, nullptr, 0.0, tmp, 2 * nsteps);
}
/* the same two jobs and, in further workgroups of the same launch, the periodic images of a single domain brought up to the positions
 * the fused pair kernel has just drifted to (k_halo_update's self-image arm): both only wait for that kernel, one launch instead of two */
struct ImageJob { int nloc, nhalo; const int *halo_src, *halo_shift; double L0, L1, L2; double4 *pos; };
/* ... or, in a decomposed run, the halo messages packed from the drifted positions (k_pack_halo, width 3) */
struct PackJob { int nsend; const unsigned *send_map; int shift[27][3]; double L0, L1, L2; const double4 *pos; double *out; };
__global__ __launch_bounds__(1024) void k_reduce_jobs_images(RedJob j0, RedJob j1, double *results, double self_ele, double *tmp, ImageJob im, PackJob pk)
{
   const int b = (int)blockIdx.x;
   if (b < 2 * RED_SPLIT) { reduce_jobs_block(b < RED_SPLIT ? j0 : j1, b % RED_SPLIT, b / RED_SPLIT, results, self_ele, tmp); return; }
   const int nimb = (im.nhalo + 1023) / 1024;
   if (b >= 2 * RED_SPLIT + nimb)
   {
      const int k = (b - 2 * RED_SPLIT - nimb) * 1024 + (int)threadIdx.x;
      if (k >= pk.nsend) return;
      const unsigned m = pk.send_map[k];
      const int i = (int)(m & 0x7ffffffu), code = (int)(m >> 27);
      const double4 p = pk.pos[i];
      double *o = pk.out + (size_t)k * 3;
      o[0] = p.x + pk.shift[code][0] * pk.L0;
      o[1] = p.y + pk.shift[code][1] * pk.L1;
      o[2] = p.z + pk.shift[code][2] * pk.L2;
      return;
   }
   const int h = (b - 2 * RED_SPLIT) * 1024 + (int)threadIdx.x;
   if (h >= im.nhalo) return;
   const int src = im.halo_src[h], code = im.halo_shift[h];
   double4 p = im.pos[src];
   p.x += (double)(code % 3 - 1) * im.L0;
   p.y += (double)((code / 3) % 3 - 1) * im.L1;
   p.z += (double)(code / 9 - 1) * im.L2;
   im.pos[im.nloc + h] = p;
}

/* final energies / virial: full list counts every pair twice */
__device__ void finish_energy(double *r, double self_ele)
{
   double lj = 0.5 * r[R_NB_LJ];
   double ele = 0.5 * r[R_NB_ELE] + self_ele;
   r[R_E + DDCMI_E_LJ] = lj;
   r[R_E + DDCMI_E_ELE] = ele;
   /* bonded scratch: bond {e,vir6} angle {e,vir6} tors {e_tors,e_impr,vir6} */
   double eb[4] = {r[R_SCR_BOND], r[R_SCR_ANGLE], r[R_SCR_TORS], r[R_SCR_TORS + 1]};
   double etot = lj + ele;
   for (int k = 0; k < 4; k++) { r[R_E + DDCMI_E_BOND + k] = eb[k]; etot += eb[k]; }
   r[R_E + DDCMI_E_RESTRAINT] = r[R_SCR_REST];
   r[R_E + DDCMI_E_TOTAL] = etot + r[R_SCR_REST];
   for (int k = 0; k < 6; k++)
      r[R_VIR + k] = 0.5 * r[R_NB_VIR + k] + ((r[R_SCR_BOND + 1 + k] + r[R_SCR_ANGLE + 1 + k]) + r[R_SCR_TORS + 2 + k]) + r[R_SCR_REST + 1 + k];
}

__global__ void k_finish_energy(double *r, double self_ele)
{
   if (threadIdx.x == 0 && blockIdx.x == 0) finish_energy(r, self_ele);
}

/* ------------------------------------------------------------------------- */
/* NGLF integrator kernels (nglf.c:67-112)                                    */
/* FRONT half kick (free.c:13-28 / berendsen.c:64-89) fused with the drift
 * (nglf.c:80-87).  The wrap of nglf.c:90 is applied at rebuild/download time
 * instead (positions stay continuous between rebuilds so image atoms and the
 * list remain valid); the downloaded coordinates are identical up to rounding. */
/* per-group data of the velocity updates, by value.  v = Berendsen scale factor of the FRONT
 * kick (1 otherwise); groups in lang_mask use the Langevin update (langevin.c:92-128, vcm = 0):
 *   FRONT  v = a v + c f + d g        BACK  v = a (v + c f + d g)
 * a = exp(-dt_half/tau), c = dt_half/m, d = sqrt(2 dt_half kB T/(m tau)) = dfac/sqrt(m), g = three unit normals.
 * The reference draws g from a per-particle LCG64 stream stored with the particle; here it is a
 * counter-based stream keyed by (seed, gid, 2*loop + BACK): the same numbers whatever the domain
 * decomposition or launch shape -- statistical, not bitwise, parity with ddcMD. */
struct GroupLambda { double v[32]; double a[32]; double dfac[32]; unsigned lang_mask; unsigned long long seed, counter_front, counter_back;
                     double scale[3]; /* barostat: positions are scaled by this (adjustPosn) before the drift; 1 otherwise */
                     unsigned vcm_mask; double vw[32][3]; /* Langevin groups with a drift velocity (langevin.c:106,167 `vcm`): v = vcm + a (v - vcm) + ... adds vw = (1 - a) vcm to either update */
                     ulonglong2 *lcg; /* RANDOM type LCG64 (ddcmi_set_random_lcg64): the beads' own streams (the reference's), in slot order; nullptr = the counter-based stream */ };
__device__ __forceinline__ unsigned long long smix64(unsigned long long z)
{
   z += 0x9E3779B97F4A7C15ull;
   z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
   z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
   return z ^ (z >> 31);
}
__device__ __forceinline__ void gauss3(unsigned long long seed, unsigned long long gid, unsigned long long counter, double &g0, double &g1, double &g2)
{
   const unsigned long long key = smix64(seed ^ smix64(gid)) + 4ull * counter;
   const double two53 = 1.0 / 9007199254740992.0;
   double u1 = ((double)(smix64(key) >> 11) + 0.5) * two53, u2 = ((double)(smix64(key + 1) >> 11) + 0.5) * two53;
   double u3 = ((double)(smix64(key + 2) >> 11) + 0.5) * two53, u4 = ((double)(smix64(key + 3) >> 11) + 0.5) * two53;
   double r = sqrt(-2.0 * log(u1)), t = 6.283185307179586476925 * u2;
   g0 = r * cos(t); g1 = r * sin(t);
   g2 = sqrt(-2.0 * log(u3)) * cos(6.283185307179586476925 * u4);
}
/* The reference's own noise: lcg64_2 (lcg64.c:137-146) under gasdev3d (random.c:135-160) -- two polar draws, x and y from the
 * first accepted pair, z from the second.  A record is LCG64_PARM {state; multID, prime}; it travels with its bead (sorts,
 * migration) and only its state changes.  The products and the sum of rsq are rounded one by one, as the accept test of the CPU code sees them. */
__device__ __forceinline__ void lcg_pair(unsigned long long &st, unsigned long long mult, unsigned long long prime, double &x, double &y, double &rsq)
{
   do
   {
      st = mult * st + prime; x = __dmul_rn((double)st, 5.4210108624275222e-20);
      st = mult * st + prime; y = __dmul_rn((double)st, 5.4210108624275222e-20);
      x = __dadd_rn(__dmul_rn(2.0, x), -1.0); y = __dadd_rn(__dmul_rn(2.0, y), -1.0);
      rsq = __dadd_rn(__dmul_rn(x, x), __dmul_rn(y, y));
   } while (rsq >= 1.0 || rsq == 0.0);
}
__device__ __forceinline__ void lcg_gauss3(ulonglong2 *lcg, int o, double &g0, double &g1, double &g2)
{
   const ulonglong2 q = lcg[o];
   const unsigned id = (unsigned)q.y;
   const unsigned long long mult = id == 0 ? 0x27bb2ee687b0b0fdull : id == 1 ? 0x2c6fe96ee78b6955ull : 0x369dea0f31a53f85ull, prime = q.y >> 32;
   unsigned long long st = q.x;
   double x, y, rsq;
   lcg_pair(st, mult, prime, x, y, rsq);
   double fac = sqrt(-2.0 * log(rsq) / rsq);
   g0 = x * fac; g1 = y * fac;
   lcg_pair(st, mult, prime, x, y, rsq);
   fac = sqrt(-2.0 * log(rsq) / rsq);
   g2 = x * fac;
   lcg[o].x = st;
}
__device__ __forceinline__ void group_gauss3(const GroupLambda &gl, int i, const uint64_t *gid, unsigned long long counter, double &g0, double &g1, double &g2)
{
   if (gl.lcg) lcg_gauss3(gl.lcg, i, g0, g1, g2);
   else gauss3(gl.seed, gid[i], counter, g0, g1, g2);
}
/* the largest |v|^2 of a workgroup's drifting beads, rounded up, into column 7 of its row of partials: the displacement bound of the
 * shell-limited walk (NbTileArgs::disp) adds dt * sqrt(max over the rows) per step (k_reduce_jobs) */
template <int NW>
__device__ __forceinline__ void block_vmax_store(float v2, double *row)
{
   __shared__ int s_vm[NW];
   const int m = wave_max_dpp(__float_as_int(v2));      /* non-negative floats order like their bit patterns */
   if ((threadIdx.x & 63) == 0) s_vm[threadIdx.x >> 6] = m;
   __syncthreads();
   if (threadIdx.x == 0)
   {
      int t = s_vm[0];
#pragma unroll
      for (int q = 1; q < NW; q++) t = max(t, s_vm[q]);
      row[7] = (double)__int_as_float(t);
   }
}
__global__ void k_kick_drift(int nloc, double dt, const double *__restrict__ invmass, const int *__restrict__ species,
                             const int *__restrict__ group, GroupLambda glambda, const uint64_t *__restrict__ gid,
                             const double *__restrict__ fx, const double *__restrict__ fy, const double *__restrict__ fz,
                             double *__restrict__ vx, double *__restrict__ vy, double *__restrict__ vz, double4 *__restrict__ pos, int mode,
                             double *__restrict__ vpart /* mode 3, not null: [block][8], column 7 = the block's largest |v|^2 of the drift */)
{
   const int i = blockIdx.x * blockDim.x + threadIdx.x;
   float v2 = 0.0f;
   if (i < nloc)
   {
      bool done = false;
      if (mode != 3)
      {
         /* nglfconstraint splits the pass around the FRONT constraint solve: mode 1 = barostat scaling of the
          * positions (adjustPosn) + kick, mode 2 = drift with the constrained velocities */
         double4 p = pos[i];
         if (mode == 2) { p.x = fma(dt, vx[i], p.x); p.y = fma(dt, vy[i], p.y); p.z = fma(dt, vz[i], p.z); pos[i] = p; done = true; }
         else if (glambda.scale[0] != 1.0 || glambda.scale[1] != 1.0 || glambda.scale[2] != 1.0)
         { p.x *= glambda.scale[0]; p.y *= glambda.scale[1]; p.z *= glambda.scale[2]; pos[i] = p; }
      }
      if (!done)
      {
         const double im = invmass[species[i]];
         double a = (0.5 * dt) * im;
         const int gr = group[i] & 31;
         double lam = glambda.v[gr];
         double x = vx[i], y = vy[i], z = vz[i];
         if (glambda.lang_mask >> gr & 1u)
         {
            double g0, g1, g2, d = glambda.dfac[gr] * sqrt(im), al = glambda.a[gr];
            group_gauss3(glambda, i, gid, glambda.counter_front, g0, g1, g2);
            x = fma(d, g0, fma(a, fx[i], al * x)); y = fma(d, g1, fma(a, fy[i], al * y)); z = fma(d, g2, fma(a, fz[i], al * z));
            if (glambda.vcm_mask >> gr & 1u) { x += glambda.vw[gr][0]; y += glambda.vw[gr][1]; z += glambda.vw[gr][2]; }
         }
         else
         {
            if (lam != 1.0) { x *= lam; y *= lam; z *= lam; }
            /* explicit fma: k_kick_ke_drift must produce the same bits as this kernel */
            x = fma(a, fx[i], x); y = fma(a, fy[i], y); z = fma(a, fz[i], z);
         }
         vx[i] = x; vy[i] = y; vz[i] = z;
         if (mode == 3)
         {
            double4 p = pos[i];
            p.x = fma(dt, x, glambda.scale[0] * p.x); p.y = fma(dt, y, glambda.scale[1] * p.y); p.z = fma(dt, z, glambda.scale[2] * p.z);
            pos[i] = p;
            v2 = __double2float_ru(x * x + y * y + z * z);
         }
      }
   }
   if (vpart) block_vmax_store<4>(v2, vpart + (size_t)blockIdx.x * 8);      /* (uniform: every thread of the block gets here) */
}
__global__ void k_scale_pos(int n, double s0, double s1, double s2, double4 *pos)
{
   int i = blockIdx.x * blockDim.x + threadIdx.x;
   if (i >= n) return;
   double4 p = pos[i];
   p.x *= s0; p.y *= s1; p.z *= s2;
   pos[i] = p;
}
#define KE_PER 4
/* BACK half kick (nglf.c:100-104) fused with kinetic_terms (energy.c:48-163):
 * rk = sum 1/2 m v^2, tion = sum m v (x) v */
__global__ __launch_bounds__(DDCMI_BLOCK) void k_kick_ke(int nloc, double dt, const double *__restrict__ invmass, const double *__restrict__ massv,
                                                         const int *__restrict__ species,
                                                         const double *__restrict__ fx, const double *__restrict__ fy, const double *__restrict__ fz,
                                                         double *__restrict__ vx, double *__restrict__ vy, double *__restrict__ vz,
                                                         double *__restrict__ partials, int do_kick,
                                                         const int *__restrict__ group, GroupLambda glambda, const uint64_t *__restrict__ gid)
{
   double acc[7] = {0, 0, 0, 0, 0, 0, 0};
   /* KE_PER beads per thread: a quarter of the partial rows for the reduction launch to read */
#pragma unroll
   for (int u = 0; u < KE_PER; u++)
   {
      int i = (blockIdx.x * KE_PER + u) * DDCMI_BLOCK + threadIdx.x;
      if (i >= nloc) continue;
      int sp = species[i];
      double x = vx[i], y = vy[i], z = vz[i];
      if (do_kick)
      {
         const double im = invmass[sp];
         double a = (0.5 * dt) * im;
         const int gr = group[i] & 31;
         if (glambda.lang_mask >> gr & 1u)
         {
            double g0, g1, g2, d = glambda.dfac[gr] * sqrt(im), al = glambda.a[gr];
            group_gauss3(glambda, i, gid, glambda.counter_back, g0, g1, g2);
            x = al * fma(d, g0, fma(a, fx[i], x)); y = al * fma(d, g1, fma(a, fy[i], y)); z = al * fma(d, g2, fma(a, fz[i], z));
            if (glambda.vcm_mask >> gr & 1u) { x += glambda.vw[gr][0]; y += glambda.vw[gr][1]; z += glambda.vw[gr][2]; }
         }
         else { x = fma(a, fx[i], x); y = fma(a, fy[i], y); z = fma(a, fz[i], z); }
         vx[i] = x; vy[i] = y; vz[i] = z;
      }
      double m = massv[sp];
      double vxx = x * x, vyy = y * y, vzz = z * z;
      acc[0] += 0.5 * m * (vxx + vyy + vzz);
      acc[1] += m * vxx; acc[2] += m * vyy; acc[3] += m * vzz;
      acc[4] += m * (x * y); acc[5] += m * (x * z); acc[6] += m * (y * z);
   }
   block_reduce_store<7>(acc, partials + (size_t)blockIdx.x * 8);
}
/* The BACK half kick + kinetic terms of step n and the FRONT half kick + drift of step
 * n+1 use the same forces: inside a batch of steps they are one pass over v and f
 * (k_kick_ke followed by k_kick_drift, bit for bit). */
__global__ __launch_bounds__(DDCMI_BLOCK) void k_kick_ke_drift(int nloc, double dt, const double *__restrict__ invmass, const double *__restrict__ massv,
                                                               const int *__restrict__ species, const int *__restrict__ group, GroupLambda glambda,
                                                               const double *__restrict__ fx, const double *__restrict__ fy, const double *__restrict__ fz,
                                                               double *__restrict__ vx, double *__restrict__ vy, double *__restrict__ vz,
                                                               double4 *__restrict__ pos, double *__restrict__ partials, const uint64_t *__restrict__ gid)
{
   double acc[7] = {0, 0, 0, 0, 0, 0, 0};
   float v2 = 0.0f;
#pragma unroll
   for (int u = 0; u < KE_PER; u++)
   {
      int i = (blockIdx.x * KE_PER + u) * DDCMI_BLOCK + threadIdx.x;
      if (i >= nloc) continue;
      int sp = species[i];
      const double im = invmass[sp];
      double a = (0.5 * dt) * im;
      double f0 = fx[i], f1 = fy[i], f2 = fz[i];
      const int gr = group[i] & 31;
      const bool lang = glambda.lang_mask >> gr & 1u;
      double x, y, z, g0, g1, g2, dl = 0.0, al = 0.0;
      if (lang)
      {
         dl = glambda.dfac[gr] * sqrt(im); al = glambda.a[gr];
         group_gauss3(glambda, i, gid, glambda.counter_back, g0, g1, g2);
         x = al * fma(dl, g0, fma(a, f0, vx[i])); y = al * fma(dl, g1, fma(a, f1, vy[i])); z = al * fma(dl, g2, fma(a, f2, vz[i]));
         if (glambda.vcm_mask >> gr & 1u) { x += glambda.vw[gr][0]; y += glambda.vw[gr][1]; z += glambda.vw[gr][2]; }
      }
      else { x = fma(a, f0, vx[i]); y = fma(a, f1, vy[i]); z = fma(a, f2, vz[i]); }
      double m = massv[sp];
      double vxx = x * x, vyy = y * y, vzz = z * z;
      acc[0] += 0.5 * m * (vxx + vyy + vzz);
      acc[1] += m * vxx; acc[2] += m * vyy; acc[3] += m * vzz;
      acc[4] += m * (x * y); acc[5] += m * (x * z); acc[6] += m * (y * z);
      if (lang)
      {
         group_gauss3(glambda, i, gid, glambda.counter_front, g0, g1, g2);
         x = fma(dl, g0, fma(a, f0, al * x)); y = fma(dl, g1, fma(a, f1, al * y)); z = fma(dl, g2, fma(a, f2, al * z));
         if (glambda.vcm_mask >> gr & 1u) { x += glambda.vw[gr][0]; y += glambda.vw[gr][1]; z += glambda.vw[gr][2]; }
      }
      else
      {
         double lam = glambda.v[gr];
         if (lam != 1.0) { x *= lam; y *= lam; z *= lam; }
         x = fma(a, f0, x); y = fma(a, f1, y); z = fma(a, f2, z);
      }
      vx[i] = x; vy[i] = y; vz[i] = z;
      v2 = fmaxf(v2, __double2float_ru(x * x + y * y + z * z));
      double4 p = pos[i];
      p.x = fma(dt, x, glambda.scale[0] * p.x); p.y = fma(dt, y, glambda.scale[1] * p.y); p.z = fma(dt, z, glambda.scale[2] * p.z);
      pos[i] = p;
   }
   block_reduce_store<7>(acc, partials + (size_t)blockIdx.x * 8);
   block_vmax_store<DDCMI_BLOCK / 64>(v2, partials + (size_t)blockIdx.x * 8);      /* column 7: the displacement bound's share of this drift */
}
/* neighborCheck (neighbor.c:117-208), constant box: displacement of every owned bead since the
 * list was built, measured relative to the centroid of the domain's beads (positions are not
 * wrapped between rebuilds here, so r - r0 needs no image logic).  Pass 1: sum of r - r0. */
__global__ __launch_bounds__(DDCMI_BLOCK) void k_disp_sum(int nloc, const double4 *__restrict__ pos, const double4 *__restrict__ pos0, double *__restrict__ partials)
{
   int i = blockIdx.x * blockDim.x + threadIdx.x;
   double acc[3] = {0, 0, 0};
   if (i < nloc) { double4 p = pos[i], q = pos0[i]; acc[0] = p.x - q.x; acc[1] = p.y - q.y; acc[2] = p.z - q.z; }
   block_reduce_store<3>(acc, partials + (size_t)blockIdx.x * 8);
}
/* pass 2: max_i |(r_i - r0_i) - mean|^2; non-negative doubles order like their bit patterns */
__global__ __launch_bounds__(DDCMI_BLOCK) void k_disp_max(int nloc, const double4 *__restrict__ pos, const double4 *__restrict__ pos0, const double *__restrict__ sum, unsigned long long *out)
{
   int i = blockIdx.x * blockDim.x + threadIdx.x;
   double d2 = 0.0;
   if (i < nloc)
   {
      double inv = 1.0 / (double)nloc;
      double4 p = pos[i], q = pos0[i];
      double x = (p.x - q.x) - sum[0] * inv, y = (p.y - q.y) - sum[1] * inv, z = (p.z - q.z) - sum[2] * inv;
      d2 = x * x + y * y + z * z;
   }
#pragma unroll
   for (int off = 32; off > 0; off >>= 1) d2 = fmax(d2, __shfl_down(d2, off, 64));
   if ((threadIdx.x & 63) == 0 && d2 > 0.0) atomicMax(out, (unsigned long long)__double_as_longlong(d2));
}
/* per-group kinetic energy and member count (energy.c:124-133) */
#define GKE_BLOCKS 512
/* per-group kinetic energy and bead count: GKE_BLOCKS workgroups stride over the beads and leave one
 * partial pair per group; k_group_ke_sum adds them in a fixed order */
__global__ __launch_bounds__(DDCMI_BLOCK) void k_group_ke(int nloc, int ngroup, const double *__restrict__ massv, const int *__restrict__ species,
                                                          const int *__restrict__ group,
                                                          const double *__restrict__ vx, const double *__restrict__ vy, const double *__restrict__ vz,
                                                          double *partials /* [GKE_BLOCKS][2*ngroup] */)
{
   __shared__ double s_red[DDCMI_BLOCK / 64][2];
   for (int g = 0; g < ngroup; g++)
   {
      double k = 0.0, c = 0.0;
      for (int i = blockIdx.x * DDCMI_BLOCK + threadIdx.x; i < nloc; i += GKE_BLOCKS * DDCMI_BLOCK)
         if (group[i] == g)
         {
            double m = massv[species[i]];
            k += 0.5 * m * (vx[i] * vx[i] + vy[i] * vy[i] + vz[i] * vz[i]);
            c += 1.0;
         }
      k = wave_sum(k); c = wave_sum(c);
      if ((threadIdx.x & 63) == 0) { s_red[threadIdx.x >> 6][0] = k; s_red[threadIdx.x >> 6][1] = c; }
      __syncthreads();
      if (threadIdx.x < 2)
      {
         double a = 0.0;
         for (int w = 0; w < DDCMI_BLOCK / 64; w++) a += s_red[w][threadIdx.x];
         partials[(size_t)blockIdx.x * 2 * ngroup + 2 * g + threadIdx.x] = a;
      }
      __syncthreads();
   }
}
__global__ void k_group_ke_sum(int ngroup, const double *__restrict__ partials, double *out)
{
   int q = threadIdx.x;
   if (q >= 2 * ngroup) return;
   double a = 0.0;
   for (int b = 0; b < GKE_BLOCKS; b++) a += partials[(size_t)b * 2 * ngroup + q];
   out[q] = a;
}

/* The per-group and per-species copies of kinetic_terms (energy.c:104-147) and the thermal flux: for every class c (a
 * group or a species) {rk, tion xx yy zz xy xz yz, mass, number, J x y z} with J_k = (K_k + U_k) v_k - 1/2 S_k v_k; on this
 * path the per-atom potential energy U_k and stress S_k are zero (martiniNonBond and the bonded terms book e->eion and
 * e->virial only, bioMartini.c:1111-1120), so J = sum K v and a class's eion stays 0.  Read at print steps only: one pass
 * per class, fixed-order sums (bitwise reproducible) like k_group_ke. */
#define KD_NV 12
__global__ __launch_bounds__(DDCMI_BLOCK) void k_class_kinetic(int nloc, int nclass, int by_species, const double *__restrict__ massv, const int *__restrict__ species,
                                                                const int *__restrict__ group,
                                                                const double *__restrict__ vx, const double *__restrict__ vy, const double *__restrict__ vz,
                                                                double *partials /* [GKE_BLOCKS][nclass][16] */)
{
   for (int c = 0; c < nclass; c++)
   {
      double acc[KD_NV] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
      for (int i = blockIdx.x * DDCMI_BLOCK + threadIdx.x; i < nloc; i += GKE_BLOCKS * DDCMI_BLOCK)
      {
         const int sp = species[i];
         if ((by_species ? sp : group[i]) != c) continue;
         const double m = massv[sp], x = vx[i], y = vy[i], z = vz[i];
         const double K = 0.5 * m * (x * x + y * y + z * z);
         acc[0] += K;
         acc[1] += m * (x * x); acc[2] += m * (y * y); acc[3] += m * (z * z);
         acc[4] += m * (x * y); acc[5] += m * (x * z); acc[6] += m * (y * z);
         acc[7] += m; acc[8] += 1.0;
         acc[9] += K * x; acc[10] += K * y; acc[11] += K * z;
      }
      block_reduce_store<KD_NV>(acc, partials + ((size_t)blockIdx.x * nclass + c) * 16);
      __syncthreads();      /* the reduction's scratch is reused by the next class */
   }
}
__global__ void k_class_kinetic_sum(int nclass, const double *__restrict__ partials, double *out)
{
   const int q = blockIdx.x * blockDim.x + threadIdx.x;
   if (q >= nclass * KD_NV) return;
   const int c = q / KD_NV, k = q % KD_NV;
   double a = 0.0;
   for (int b = 0; b < GKE_BLOCKS; b++) a += partials[((size_t)b * nclass + c) * 16 + k];
   out[q] = a;
}

/* download helpers: caller order + wrap */
__global__ void k_export_pos(GridParams gp, int nloc, const double4 *pos, const int *orig, double *ox, double *oy, double *oz)
{
   int i = blockIdx.x * blockDim.x + threadIdx.x;
   if (i >= nloc) return;
   double4 p = pos[i];
   if (gp.pbc & 1) { if (p.x > 0.5 * gp.L[0]) p.x -= gp.L[0]; if (p.x < -0.5 * gp.L[0]) p.x += gp.L[0]; }
   if (gp.pbc & 2) { if (p.y > 0.5 * gp.L[1]) p.y -= gp.L[1]; if (p.y < -0.5 * gp.L[1]) p.y += gp.L[1]; }
   if (gp.pbc & 4) { if (p.z > 0.5 * gp.L[2]) p.z -= gp.L[2]; if (p.z < -0.5 * gp.L[2]) p.z += gp.L[2]; }
   int o = orig[i];
   ox[o] = p.x; oy[o] = p.y; oz[o] = p.z;
}
__global__ void k_export3(int nloc, const double *a, const double *b, const double *c, const int *orig, double *oa, double *ob, double *oc)
{
   int i = blockIdx.x * blockDim.x + threadIdx.x;
   if (i >= nloc) return;
   int o = orig[i];
   oa[o] = a[i]; ob[o] = b[i]; oc[o] = c[i];
}
/* record tag w (bit-cast into pos.w): [63:32] molecule id (gid>>32, bioGid.h) [31:16] species
 * [15:8] atom-in-molecule code (gid & 0xffff when < 255, else 255 = "look at the gid")
 * [7:0] LJ type.  The atom code lets the list build decide bonded-pair exclusions from
 * LDS instead of two dependent global loads per same-molecule candidate. */
__global__ void k_init_state(int n, const double *rx, const double *ry, const double *rz, const int *species, const int *ljtype_sp,
                             const uint64_t *gid, double4 *pos, int *orig, int *slot)
{
   int i = blockIdx.x * blockDim.x + threadIdx.x;
   if (i >= n) return;
   int sp = species[i];
   unsigned code = (unsigned)(gid[i] & DDCMI_GID_ATMGRPMASK);
   long long w = (long long)(gid[i] & DDCMI_GID_MOLMASK) | ((long long)(sp & 0xffff) << 16) | ((long long)min(code, 255u) << 8) | (long long)(ljtype_sp[sp] & 0xff);
   pos[i] = make_double4(rx[i], ry[i], rz[i], __longlong_as_double(w));
   orig[i] = i;
   slot[i] = i;
}
/* test/inspection export: decode the tile ELL (16-bit staged indices) into CSR
 * over caller-order indices; image atoms map back to their source bead */
__global__ void k_tilelist_to_csr(NbTileArgs ta, int pack_type, int nloc, const int *orig, const int *halo_src, const int *start, int *jout)
{
   int t = blockIdx.x;
   int ts = ta.cell_start_o[TCELLS * t];
   int nown = ta.cell_start_o[TCELLS * t + TCELLS] - ts;
   if (nown <= 0) return;
   const int *sidx = ta.stage_idx + (size_t)t * ta.stage_stride;
   long long base = ta.tile_base[t];
   int rows = ta.tile_rows[t];
   for (int al = threadIdx.x; al < nown; al += blockDim.x)
   {
      int a = ts + al;
      int s = start[orig[a]];
      int cnt = ta.nbr_cnt[a];
      for (int k = 0; k < cnt; k++)
      {
         int ee = ta.nbr16[base + ((size_t)(k >> 3) * rows + al) * 8 + (k & 7)];
         int j = sidx[(pack_type ? (ee >> 4) : ee) - 1];      /* (pack_type 2: the nibble's shift bit is not part of the slot) */
         if (j >= nloc) j = halo_src[j - nloc];
         jout[s + k] = orig[j];
      }
   }
}
/* the same for the excluded (same-molecule bonded) pairs: their list-format entries (excl16) name staged slots of the bead's tile */
__global__ void k_tileexcl_to_csr(NbTileArgs ta, int pack_type, int nloc, int npad, const unsigned short *excl16, const int *excl_cnt,
                                  const int *orig, const int *halo_src, const int *start, int *jout)
{
   int t = blockIdx.x;
   int ts = ta.cell_start_o[TCELLS * t];
   int nown = ta.cell_start_o[TCELLS * t + TCELLS] - ts;
   if (nown <= 0) return;
   const int *sidx = ta.stage_idx + (size_t)t * ta.stage_stride;
   for (int al = threadIdx.x; al < nown; al += blockDim.x)
   {
      int a = ts + al;
      int s = start[orig[a]];
      int cnt = excl_cnt[a];
      for (int k = 0; k < cnt; k++)
      {
         int ee = excl16[(size_t)k * npad + a];
         int j = sidx[(pack_type ? (ee >> 4) : ee) - 1];
         if (j >= nloc) j = halo_src[j - nloc];
         jout[s + k] = orig[j];
      }
   }
}
__global__ void k_list_to_csr(int nloc, int npad, const int *lst, const int *cnt, const int *orig, const int *halo_src, const int *start, int *jout)
{
   int i = blockIdx.x * blockDim.x + threadIdx.x;
   if (i >= nloc) return;
   int o = orig[i];
   int s = start[o];
   for (int k = 0; k < cnt[i]; k++)
   {
      int j = lst[(size_t)k * npad + i];
      if (j >= nloc) j = halo_src[j - nloc];
      jout[s + k] = orig[j];
   }
}
__global__ void k_counts_by_orig(int nloc, const int *cnt, const int *orig, int *out)
{
   int i = blockIdx.x * blockDim.x + threadIdx.x;
   if (i < nloc) out[orig[i]] = cnt[i];
}

