/*
 * ddcmi.h -- C-ABI of libddcmi.so: the MI355X (gfx950) device side of ddcMD's
 * Martini MD inner loop.  Plain pointers and sizes only; the caller owns host
 * arrays, the library owns device memory.  Every call returns 0 on success or a
 * negative DDCMI_E* code; ddcmi_last_error() gives the message.  No exceptions,
 * no callbacks.  One host thread per context; all work is queued on the
 * context's HIP stream and only the calls marked [sync] wait for it.
 *
 * Each entry point names the reference interface it replaces (paths relative to
 * /root/reference/src).  All quantities are in ddcMD internal units (bohr, fs,
 * Rydberg, e; kB = 1: ddcMD.c:71-73).  Particle arrays are in the CALLER's order
 * on both upload and download; the library keeps its own cell-sorted order
 * internally.
 */
#ifndef DDCMI_H
#define DDCMI_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ddcmi_ctx ddcmi_ctx;

enum
{
   DDCMI_OK = 0,
   DDCMI_ENODEVICE = -1,    /* no HIP device / HIP runtime error */
   DDCMI_EINVAL = -2,       /* bad argument or call order */
   DDCMI_ENOMEM = -3,
   DDCMI_EUNSUPPORTED = -4, /* e.g. non-orthorhombic box, box < 2*(rmax+deltaR) */
   DDCMI_ECOMM = -5         /* RCCL error */
};

/* energies[] slots of ddcmi_eval_forces / ddcmi_get_energies (BIOENERGIES,
 * bioCharmmParms.h; e->eion is slot DDCMI_E_TOTAL) */
enum { DDCMI_E_LJ = 0, DDCMI_E_ELE, DDCMI_E_BOND, DDCMI_E_ANGLE, DDCMI_E_TORS, DDCMI_E_IMPR, DDCMI_E_TOTAL, DDCMI_E_RESTRAINT, DDCMI_NE };
/* virial / tion component order (THREE_SMATRIX as summed in bioMartini.c:1098-1103) */
enum { DDCMI_XX = 0, DDCMI_YY, DDCMI_ZZ, DDCMI_XY, DDCMI_XZ, DDCMI_YZ };
/* download mask */
enum { DDCMI_POS = 1, DDCMI_VEL = 2, DDCMI_FORCE = 4 };
/* group (thermostat) kinds: free.c, berendsen.c */
enum { DDCMI_FREE = 0, DDCMI_BERENDSEN = 1, DDCMI_LANGEVIN = 2 };

/* particle labels (gid_type = uint64_t, gid.h:13): MOL 32 | RESID 16 | GROUP 8 | ATOM 8 (bioGid.h:13-22).  The molecule id
 * (reOrgPairs, molecule lists), the residue runs of the bonded terms (charmmResidues) and the group:atom codes of the
 * bonded-pair lists are read off a label with these masks; tests/test_abi.py holds them against the reference's own header */
#define DDCMI_GID_MOLSHIFT   32
#define DDCMI_GID_ATMMASK    0x00000000000000ffull
#define DDCMI_GID_ATMGRPMASK 0x000000000000ffffull
#define DDCMI_GID_GRPMASK    0x000000000000ff00ull
#define DDCMI_GID_RESMASK    0x00000000ffff0000ull
#define DDCMI_GID_MOLMASK    0xffffffff00000000ull
#define DDCMI_GID_MOLRESMASK 0xffffffffffff0000ull

/* ---- context -------------------------------------------------------------
 * replaces accelerator_init / accelerator_getAccelerator (accelerator.c:10-56)
 * and the allocation half of allocSendGPUState / allocGPUBoxInfo
 * (gpuMemUtils.cu).  device = HIP device ordinal. */
int ddcmi_create(ddcmi_ctx **ctx, int device);
void ddcmi_destroy(ddcmi_ctx *ctx);
const char *ddcmi_last_error(const ddcmi_ctx *ctx);   /* ctx may be NULL: last create error */
int ddcmi_device_count(void);                          /* hipGetDeviceCount; 0 without a GPU */
const char *ddcmi_version(void);

/* ---- parameters ------------------------------------------------------------
 * Every setter checks what it is handed -- counts, NULL arrays, offsets (start at 0, never decrease), indices (not negative), constants (finite; rmax > 0,
 * deltaR >= 0, masses > 0, pbc in 0..7) -- and refuses with DDCMI_EINVAL and a message; a refused call changes nothing.  What only a rebuild can know is
 * checked there (molecule types against ddcmi_set_molecules, index-named terms / constraint pairs / molecules against the bead count of the upload).  The
 * caller's promise that an array is as long as its count says cannot be checked.  New species or nonbonded parameters under an uploaded state invalidate
 * the forces on the device: the next ddcmi_eval_forces rebuilds the class tables, the beads' tags and the list (ddcmi_step_nglf asks for it). */
/* BOX h (row-major 3x3, orthorhombic) and pbc bitmask: allocGPUBoxInfo */
int ddcmi_set_box(ddcmi_ctx *ctx, const double h[9], int pbc);
/* SPECIES tables: mass, charge (ddcenergy.c:210), LJ type (getCGLJindexbySpecie
 * bioMartini.c:952), molecule type (molecule.c:56) */
int ddcmi_set_species(ddcmi_ctx *ctx, int nspecies, const double *mass, const double *charge,
                      const int *ljtype, const int *moltype);
/* martiniNonBondGPUParms (bioMartiniGPU.h:8): LJ table [nlj*nlj] {sigma,eps,shift},
 * rmax ("cutoff"), keR = ke/epsilon_r, krf, crf (bioMartini.c:1234-1245) */
int ddcmi_set_nonbonded(ddcmi_ctx *ctx, int nlj, const double *sigma, const double *eps, const double *shift,
                        double rmax, double keR, double krf, double crf);
/* reOrgPairs inputs (bioMartini.c:1392-1485): per molecule type nSpecies and the
 * bpairList (atmgrp codes) of its ownership residue.  nmoltype = 0: no MOLECULECLASS */
int ddcmi_set_molecules(ddcmi_ctx *ctx, int nmoltype, const int *mol_nspecies, const int *bpair_off,
                        const int *bpairI, const int *bpairJ);
/* martiniBondGPUParms (bioMartiniGPU.h:9): bonded terms as lists over CALLER-order
 * atom indices.  angle func 1/2/10 = resAngleSorted/resAngleCosineSorted/
 * resAngleRestrainSorted; tors func 1/2 = resTorsionSorted/resImproperSorted
 * (bioCharmmCovalentEnergiesSorted.c).  excludePotentialTerm: bioCharmmParms.h:25-28 */
int ddcmi_set_bonded(ddcmi_ctx *ctx,
                     int nbond, const int *bond_ij, const double *bond_kb, const double *bond_b0,
                     int nangle, const int *angle_ijk, const int *angle_func, const double *angle_k, const double *angle_t0,
                     int ntors, const int *tors_ijkl, const int *tors_func, const int *tors_n, const double *tors_k, const double *tors_delta,
                     int excludePotentialTerm);
/* The same terms with atoms named by gid (gid_type, gid.h), for decomposed runs where a
 * bead's index does not survive migration: every rank passes the GLOBAL term list.  At
 * each list rebuild a rank finds the owned or halo copy of every term atom; it evaluates
 * the terms of the atoms it owns and keeps those atoms' forces, so no force is sent back,
 * and a term's energy and virial are booked by the owner of its first atom: the sums over
 * ranks count every term once.  This replaces the whole-molecule ownership of
 * ddcRuleMartini (bioMartiniRule.c:64-85): a partner further than rmax+deltaR from the
 * owning domain is an error at the rebuild. */
int ddcmi_set_bonded_gid(ddcmi_ctx *ctx,
                         int nbond, const uint64_t *bond_gid, const double *bond_kb, const double *bond_b0,
                         int nangle, const uint64_t *angle_gid, const int *angle_func, const double *angle_k, const double *angle_t0,
                         int ntors, const uint64_t *tors_gid, const int *tors_func, const int *tors_n, const double *tors_k, const double *tors_delta,
                         int excludePotentialTerm);
/* POTENTIAL type=RESTRAINT (restraint.c:259-361; restraintGPU.cu): harmonic position restraints on the
 * beads with the listed gids.  fc[3n] = fcx fcy fcz switches, r0[3n] = x0 y0 z0 as fractions of the box,
 * kb[n]; origin 0: the box is centred on the origin (r0*L - L/2).  Energy lands in DDCMI_E_RESTRAINT
 * and in DDCMI_E_TOTAL.  n = 0 removes them. */
int ddcmi_set_restraints(ddcmi_ctx *ctx, int n, const uint64_t *gid, const int *fc, const double *r0, const double *kb, int origin);
/* NEIGHBOR deltaR (neighbor.c:49) and DDC updateRate (ddc.c:96).  updateRate = 0: rebuild when
 * neighborCheck (neighbor.c:117-208) finds 2*max displacement >= deltaR (one host round trip per step) */
int ddcmi_set_neighbor(ddcmi_ctx *ctx, double deltaR, int updateRate);
/* GROUP objects (group.c:48-90): type DDCMI_FREE / DDCMI_BERENDSEN{Teq,tau,interval} /
 * DDCMI_LANGEVIN{Teq,tau} (langevin.c:92-128, constant Teq, vcm = 0).  Teq in internal energy units
 * (kB = 1).  The Langevin noise is a counter-based normal stream keyed by (seed, gid, loop): the
 * reference's per-particle LCG64 states are not reproduced -- statistical parity only. */
int ddcmi_set_groups(ddcmi_ctx *ctx, int ngroup, const int *type, const double *Teq, const double *tau, const int *interval);
/* LANGEVIN groups, the rest of langevin_velocityUpdate (langevin.c:92-128): `vcm`, the velocity the friction relaxes towards
 * (:106,167; v = vcm + a (v - vcm) + c f + d g), per group, [3 ngroup], internal units; after ddcmi_set_groups (which resets it
 * to zero).  And Teq as a function of time (`Teq_dynamics = EXPLICIT_TIME`, :49,84-85): langevin_Update re-evaluates the group's
 * equation once per step on the host; here the host does the same between calls of ddcmi_step_nglf with ddcmi_set_group_temperature
 * (the equation grammar is simutil's eq_parse, which the reference tree does not hold: the deck loader takes constants). */
int ddcmi_set_group_vcm(ddcmi_ctx *ctx, int ngroup, const double *vcm);
int ddcmi_set_group_temperature(ddcmi_ctx *ctx, int group, double Teq);
/* INTEGRATOR type=NGLFCONSTRAINT (nglfconstraint.c:510-574): NGLF plus a semi-isotropic Berendsen
 * barostat (changeVolume, :64-84) driven by the molecular pressure of the last force evaluation at the
 * target temperature T: lambda_xy = cbrt(1 + beta dt/tau ((Pxx+Pyy)/2 - P0)), lambda_z likewise from Pzz;
 * box and positions scaled before the FRONT kick.  beta = 0 switches it off.  Costs one host round trip per step
 * (decomposed runs: one all-reduce of the virial and the molecular terms, ddcmi_set_molecule_lists_gid).  Without ddcmi_set_molecule_lists every bead is its own molecule.  Internal units. */
int ddcmi_set_barostat(ddcmi_ctx *ctx, double T, double P0, double beta, double tau);
/* on: ONE scale factor for the three axes, from the mean of Pxx, Pyy, Pzz -- what the reference's GPU integrator
 * NGLFGPULANGEVIN applies (changeVolumeGPUisotropic, molecularPressureGPU.cu:204-239); off (default): changeVolume's
 * semi-isotropic form */
int ddcmi_set_barostat_isotropic(ddcmi_ctx *ctx, int on);
/* current box (it changes under the barostat) */
int ddcmi_get_box(const ddcmi_ctx *ctx, double h[9]);
/* the molecular pressure (xx, yy, zz; molecularPressure.c:57-67) the barostat acted on in the last step */
int ddcmi_get_barostat_pressure(const ddcmi_ctx *ctx, double p[3]);
/* Molecules for the barostat's molecular virial (molecularVirial, molecularPressure.c:23-56):
 * nmol_total = N in the N kB T term; the nmulti molecules of two or more beads are listed as caller-order
 * atom indices, molecule m = mol_atoms[mol_off[m] .. mol_off[m+1]).  One domain. */
int ddcmi_set_molecule_lists(ddcmi_ctx *ctx, long nmol_total, int nmulti, const int *mol_off, const int *mol_atoms);
/* The same with atoms named by gid, for decomposed runs (every rank passes the GLOBAL lists; mol_mass[m] = total mass of
 * molecule m).  Beads migrate one by one, so a molecule may have atoms on several ranks: each rank sums over the atoms it
 * owns; the centre of mass and total force of such SPLIT molecules are completed by an all-reduce of six doubles per split
 * molecule in every step the barostat acts, together with the virial.  No limit on the molecule's extent. */
int ddcmi_set_molecule_lists_gid(ddcmi_ctx *ctx, long nmol_total, int nmulti, const int *mol_off, const uint64_t *mol_atom_gid, const double *mol_mass);
/* nglfconstraint's velocity constraints (velocityConstraintOld/resMoveConsOld, nglfconstraint.c:180-264,
 * 438-455; groups from genConstraint, bioMartini.c:300-445).  Group g holds the pairs
 * [pair_off[g], pair_off[g+1]) in the order the reference sweeps them; pairI/pairJ are caller-order atom
 * indices, dist the constrained lengths (internal units).  Groups must not share atoms.  With groups set,
 * every step runs FRONT kick -> constraint ((r + dt v)^2 = d^2) -> drift -> forces -> BACK kick ->
 * constraint (r.v = 0) -> kinetic terms.  Gauss-Seidel sweeps to |rvab dt| < 1e-12, at most 500, as the
 * reference.  The constraint virial is not booked (the reference computes and drops it).  Caller-order indices: one domain
 * (decomposed runs: ddcmi_set_constraints_gid). */
int ddcmi_set_constraints(ddcmi_ctx *ctx, int ngroups, const int *pair_off, const int *pairI, const int *pairJ, const double *dist);
/* The same groups with atoms named by gid, for decomposed runs (every rank passes the GLOBAL list): at each rebuild a rank
 * finds the owned or halo copy of every group atom; a rank that owns an atom of a group solves the whole group -- positions of
 * the partners from the position halo, their velocities from a velocity halo exchanged before each of the two solves of a
 * step -- and keeps the velocities of the atoms it owns.  A partner further than rmax+deltaR from the owning domain is an
 * error at the rebuild. */
int ddcmi_set_constraints_gid(ddcmi_ctx *ctx, int ngroups, const int *pair_off, const uint64_t *pairI, const uint64_t *pairJ, const double *dist);
/* largest sweep count of any group since the last reset, and how many group solves hit the 500-sweep cap */
int ddcmi_constraint_stats(ddcmi_ctx *ctx, int *max_sweeps, int *unconverged, int reset);
/* RANDOM seed (random.c:44-60) for the Langevin noise */
int ddcmi_set_random(ddcmi_ctx *ctx, uint64_t seed);
/* RANDOM type LCG64 (lcg64.c, random.c:135-160): Langevin groups draw from the reference's own per-particle streams -- three
 * unit normals per half kick by gasdev3d over lcg64_2 -- instead of the counter-based one.  n LCG64_PARM records
 * {state, multID, prime} (lcg64.h:8-12) in the caller order of the last ddcmi_upload_state, which they must follow (and, in a
 * decomposed run, precede the first list build); what a particle without a random field gets is lcg64_default's business
 * (deck.c restates it for the atoms reader).  The records travel with their beads: through the sorts of a rebuild and, in a
 * decomposed run, inside the migration records (particleRegisterinfo of random->parmsArray, random.c:72-76).  n = 0: back to
 * the counter-based stream.  ddcmi_get_random_lcg64 returns the advanced states (collection_write.c:157-161) in caller
 * order -- in the order of ddcmi_download_particles for a decomposed run, whose beads have no caller order left. [sync] */
int ddcmi_set_random_lcg64(ddcmi_ctx *ctx, int n, const uint64_t *state, const uint32_t *multID, const uint32_t *prime);
int ddcmi_get_random_lcg64(ddcmi_ctx *ctx, int n, uint64_t *state, uint32_t *multID, uint32_t *prime);
/* SIMULATE loop/time (simulate.c:146,155) */
int ddcmi_set_clock(ddcmi_ctx *ctx, int64_t loop, double time);

/* ---- state ----------------------------------------------------------------- */
/* allocSendGPUState + sendGPUState + sendForceVelocityToGPU (gpuMemUtils.h):
 * nlocal particles in caller order.  v may be NULL (zero). [sync] */
int ddcmi_upload_state(ddcmi_ctx *ctx, int nlocal,
                       const double *rx, const double *ry, const double *rz,
                       const double *vx, const double *vy, const double *vz,
                       const uint64_t *gid, const int *species, const int *group);
/* sendGPUState between list rebuilds (gpuMemUtils.cu; martiniGPU1 with a CPU integrator, bioMartini.cu:157): new
 * positions (and velocities, if not NULL) of the SAME particles in caller order; the neighbour list stays valid.
 * A position that was wrapped into the box on the host is taken at its periodic image nearest to where the bead
 * was (the device keeps positions continuous between rebuilds). */
int ddcmi_upload_positions(ddcmi_ctx *ctx, const double *rx, const double *ry, const double *rz,
                           const double *vx, const double *vy, const double *vz);
/* sendHostState / sendForceVelocityToHost / sendPosnToHost: any pointer may be
 * NULL; positions come back wrapped into the box like backInBox_fast
 * (nglf.c:90). [sync] */
int ddcmi_download_state(ddcmi_ctx *ctx, int mask,
                         double *rx, double *ry, double *rz,
                         double *vx, double *vy, double *vz,
                         double *fx, double *fy, double *fz);
int ddcmi_nlocal(const ddcmi_ctx *ctx);

/* ---- hot path -------------------------------------------------------------- */
/* constructList (nlistGPU.cu:1459; hook ddcUpdateAll.c:136-139): wrap, bin-sort,
 * image atoms, full neighbour list within rmax+deltaR, exclusion split. [sync] */
int ddcmi_build_list(ddcmi_ctx *ctx);
/* martiniGPU1 (bioMartini.cu:146-171) = zeroGPUForceEnergyBuffers + charmmPairGPU
 * + charmmConvalentGPU: forces on the device, energies[DDCMI_NE], virial[6].
 * Builds the list first when none exists.  [sync] */
int ddcmi_eval_forces(ddcmi_ctx *ctx, double *energies, double *virial);
/* nglfGPU (nglfGPU.cu:511; CPU contract nglf.c:67-112): nsteps velocity-Verlet
 * steps fully on the device incl. rebuild every updateRate loops and
 * kinetic_terms each step.  Not synchronising except at rebuilds. */
int ddcmi_step_nglf(ddcmi_ctx *ctx, double dt, int nsteps);
/* sendForceEnergyToHost + kineticGPU: energies/virial of the last force
 * evaluation, rk and tion[6] of the last kinetic_terms. [sync] */
int ddcmi_get_energies(ddcmi_ctx *ctx, double *energies, double *virial, double *rk, double *tion);
/* kinetic_terms (energy.c:48-163) on the current velocities. [sync] */
int ddcmi_kinetic(ddcmi_ctx *ctx, double *rk, double *tion);
/* the per-group (by_species = 0) or per-species (1) copies of kinetic_terms and the thermal flux, energy.c:104-147:
 * out[12 c + k] for class c = {rk, tion xx yy zz xy xz yz, mass, number, J x y z} over this rank's beads, J = sum (K + U) v - S v / 2
 * with the per-atom U and S this path leaves at zero (ddcenergy.c:152-154, bioMartini.c:1111-1120); the classes' J add up to
 * e->thermal_flux.  nclass must be the context's group / species count.  For print steps. [sync] */
int ddcmi_kinetic_detail(ddcmi_ctx *ctx, int by_species, int nclass, double *out);
/* eval_energyInfo group branch (energyInfo.c:118-141): refresh the per-group
 * temperatures Berendsen reads; returns them in Tgroup[ngroup] if not NULL.  With an
 * RCCL communicator the sums are all-reduced first (collective call). [sync] */
int ddcmi_group_temperatures(ddcmi_ctx *ctx, double *Tgroup);
int ddcmi_get_clock(const ddcmi_ctx *ctx, int64_t *loop, double *time);
int ddcmi_sync(ddcmi_ctx *ctx);

/* ---- introspection / measurement ------------------------------------------- */
/* list statistics of the last build: stats[0]=stored full-list entries,
 * [1]=excluded-list entries, [2]=ELL width, [3]=image (halo) atoms, [4]=cells,
 * [5]=rebuild count */
int ddcmi_list_stats(const ddcmi_ctx *ctx, int64_t stats[8]);
/* the per-step halo exchange of a decomposed run, as laid out at the last rebuild (ddcSendRecvTables, ddcSendRecv.c:41):
 * stats[0]=beads this rank sends per step, [1]=beads it receives, [2]/[3]=messages sent/received (one per peer),
 * [4]=RCCL version code (ncclGetVersion), [5]=transport (0 none, 1 RCCL, 2 host-staged TCP, 3 RCCL loopback), [6]=ranks, [7]=rank */
int ddcmi_comm_stats(const ddcmi_ctx *ctx, int64_t stats[8]);
/* the same exchange peer by peer: returns the number of peers (one message each way per step) and fills up to cap of them --
 * peer rank, beads sent to it and received from it per step (x 24 B on the wire; 2x2x2 bricks: seven peers over seven xGMI links) */
int ddcmi_comm_peer_stats(const ddcmi_ctx *ctx, int cap, int *peer, int64_t *send_beads, int64_t *recv_beads);
/* Copy the full neighbour list out as CSR over caller-order indices (image atoms
 * are mapped back to their source atom). start[nlocal+1]; j may be NULL to query
 * the size (returned through *nentries).  which: 0 kept, 1 excluded. [sync] */
int ddcmi_get_list(ddcmi_ctx *ctx, int which, int *start, int *j, int64_t *nentries);
/* HIP-event timing of the nonbonded kernel on the context's stream:
 * enable, run, then read {launch count, total ms}.  [sync] on read */
int ddcmi_timing_enable(ddcmi_ctx *ctx, int on);
int ddcmi_timing_read(ddcmi_ctx *ctx, int64_t *launches, double *total_ms, int reset);
/* of what the last ddcmi_timing_read returned: the launches (and their time) whose epilogue was the integrator's pass -- between
 * print steps ddcmi_step_nglf runs BACK kick, kinetic_terms, FRONT kick and drift (nglf.c:74-104) inside the pair kernel when
 * the force is complete at the end of the list walk (no bonded terms, restraints, constraints, barostat, charges) */
int ddcmi_timing_fused(ddcmi_ctx *ctx, int64_t *launches, double *total_ms);
/* native stream handle (hipStream_t) for callers that time with their own events */
void *ddcmi_stream(ddcmi_ctx *ctx);

/* ---- multi-GPU: spatial decomposition over RCCL ----------------------------
 * replaces ddc_init / ddcAssignment / ddcSendRecvTables / ddcUpdate (ddc.c,
 * ddcAssignment.c, ddcSendRecv.c, ddcUpdate.c).  The caller distributes the
 * 128-byte id produced on rank 0 (MPI_Bcast in ddcMD, torch.distributed in
 * bench.py).  With a full list no force return (ddcUpdateForce) is needed. */
int ddcmi_comm_unique_id(char id[128]);
int ddcmi_comm_init(ddcmi_ctx *ctx, int rank, int nranks, const char id[128], int px, int py, int pz);
/* energyInfo.c:9-63 allreduce(): sum the 24-double ETYPE block across ranks */
int ddcmi_comm_allreduce_sum(ddcmi_ctx *ctx, double *values, int n);
/* Preflight of a fresh communicator (RCCL or host transport), right behind ddcmi_comm_init[_host] and before any timing: ONE grouped
 * exchange of a known pattern along every direction the brick plan names (the peers, matching and grouping of ddcUpdate's halo
 * exchange, ddcUpdate.c:56-85 / ddcSendRecv.c:126-225), one 24-double sum all-reduce (energyInfo.c:37) and one int all-gather (the
 * rebuild's count round), each verified on the receiver.  Collective.  On a mismatch, or when timeout_s (<= 0: 60 s) passes without
 * completion, the call returns DDCMI_ECOMM on EVERY rank and ddcmi_last_error says which stage, peer rank and direction failed;
 * after a timeout the RCCL communicator has been aborted (start a fresh process for another transport).  The all-gather also carries a
 * hash of what every rank must have been given alike (box, cut-offs, neighbour settings, LJ table, species, molecule tables, term counts,
 * groups, barostat, process grid: set them BEFORE the preflight): a rank set up from another deck ends the launch with DDCMI_EINVAL on
 * every rank, named ([4]; [5] = -1).
 * report (may be NULL): [0] distinct peer ranks, [1] directions exchanged, [2] bytes per direction message, [3] 1 if a peer was
 * named, [4] that peer, [5] its direction code, [6] stages verified (3 = all), [7] elapsed microseconds, [8..15] the peer ranks. [sync] */
int ddcmi_comm_preflight(ddcmi_ctx *ctx, double timeout_s, int64_t report[16]);
/* ---- process rendezvous without MPI (host/rdzv.c) ----------------------------
 * ddcMD takes rank, size, MPI_Bcast, MPI_Barrier and MPI_Allreduce from its MPI launcher
 * (ddcMD.c:93-139; energyInfo.c:9-63).  A one-process-per-GPU launch without MPI hands each
 * process RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT only; these calls turn that into a
 * full mesh of TCP streams.  Rank 0 listens on (addr, port); with port <= 0 (the launcher
 * itself occupies MASTER_PORT, as torch.distributed.run does) rank 0 binds an ephemeral port
 * and publishes it in port_file, which the other ranks poll.  All calls are collective,
 * blocking, and fail with DDCMI_ECOMM after timeout_s (<= 0: 300 s) instead of hanging. */
typedef struct ddcmi_rdzv ddcmi_rdzv;
int ddcmi_rdzv_create(ddcmi_rdzv **out, int rank, int world, const char *addr, int port, const char *port_file, double timeout_s);
void ddcmi_rdzv_destroy(ddcmi_rdzv *h);
/* leave the job at once (a rank that failed where its peers already wait for it): every stream is shut down, the peers'
 * transfers fail with "peer closed" instead of waiting out their timeout.  The handle stays valid for ddcmi_rdzv_destroy. */
void ddcmi_rdzv_abort(ddcmi_rdzv *h);
const char *ddcmi_rdzv_last_error(const ddcmi_rdzv *h);      /* h may be NULL: last create error */
int ddcmi_rdzv_rank(const ddcmi_rdzv *h);
int ddcmi_rdzv_world(const ddcmi_rdzv *h);
int ddcmi_rdzv_bcast(ddcmi_rdzv *h, void *buf, size_t nbytes, int root);                 /* MPI_Bcast: the 128-byte RCCL id */
int ddcmi_rdzv_barrier(ddcmi_rdzv *h);                                                   /* MPI_Barrier */
int ddcmi_rdzv_allreduce_f64(ddcmi_rdzv *h, double *v, int n, int op);                   /* op 0 sum (rank order), 1 max */
int ddcmi_rdzv_allgather(ddcmi_rdzv *h, const void *send, void *recv, size_t nbytes);    /* nbytes per rank */
/* grouped point-to-point exchange, matched like ncclSend/ncclRecv inside one group: the k-th message
 * sent to peer p is the k-th message p receives from this rank */
int ddcmi_rdzv_exchange(ddcmi_rdzv *h, int nsend, const int *send_peer, const void *const *send_buf, const size_t *send_bytes,
                        int nrecv, const int *recv_peer, void *const *recv_buf, const size_t *recv_bytes);
/* Decomposition over the HOST transport: the same migration / halo / all-reduce protocol as
 * ddcmi_comm_init, with every message staged through pinned host memory and carried by the
 * rendezvous' TCP streams instead of RCCL.  For ranks that share one GPU (RCCL refuses two ranks
 * on a device: tests on a single-GPU box) and for nodes without working GPU peer access; rank and
 * size come from the rendezvous, which must outlive the context. */
int ddcmi_comm_init_host(ddcmi_ctx *ctx, ddcmi_rdzv *rdzv, int px, int py, int pz);
/* Call order with decomposition: ddcmi_set_box -> ddcmi_comm_init -> set_* ->
 * ddcmi_upload_state with THIS rank's local beads (any beads inside the box are
 * accepted; the first rebuild migrates them to their owners, ddcAssignment.c) */
int ddcmi_domain_bounds(const ddcmi_ctx *ctx, double lo[3], double hi[3]);
/* current local beads in device order, identified by gid (beads migrate between
 * ranks; ddcMD identifies them by label).  Pointers may be NULL. [sync] */
int ddcmi_download_particles(ddcmi_ctx *ctx, int cap, int *nout, uint64_t *gid, int *species,
                             double *rx, double *ry, double *rz, double *vx, double *vy, double *vz,
                             double *fx, double *fy, double *fz);

#ifdef __cplusplus
}
#endif
#endif
