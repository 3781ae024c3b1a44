/* ddcmi_test.h -- TEST-ONLY entry points of the device code.  NOT part of the drop-in boundary: libddcmi.so (what a ddcMD
 * maintainer links with -lddcmi) exports include/ddcmi.h and nothing else.  The functions below are exported by a second link
 * of the very same objects, libddcmi_test.so (ddcmd_amd/csrc/Makefile: same kernels, wider version script), which only
 * tests/ and tools/ load:
 *   ddcmi_group_*              in-process emulation of a px*py*pz decomposition on ONE GPU (the multi-domain path without RCCL)
 *   ddcmi_plan_*               the host logic of the halo exchange / domain directions, callable without a GPU (world-2 CPU tests)
 *   ddcmi_debug_branch_census  which rarely taken dihedral branches a test's geometry drives */
#ifndef DDCMI_TEST_H
#define DDCMI_TEST_H
#include "ddcmi.h"
#ifdef __cplusplus
extern "C" {
#endif
/* census of the rarely taken branches of the dihedral code since the last reset (bioCharmmCovalentEnergiesSorted.c:649-683,
 * 793-810): [0] torsion series (|sin phi| <= 1e-8), of these [1] delta < 1 deg, [2] delta > 179 deg, [3] any other delta;
 * [4] improper series; [5] improper difference wrapped by 2 pi; [6] cos phi clamped.  Counted per evaluation (a term is
 * evaluated once per atom it has).  Test aid: shows that a test's geometry really drives those branches. [sync] */
int ddcmi_debug_branch_census(unsigned long long out[8], int reset);
/* Host logic of the halo exchange (ddcSendRecvTables, ddcSendRecv.c:126-225), callable without a GPU.
 * ddcmi_plan_recv_counts: from the all-gathered per-direction send counts all_counts[nranks][27], what
 * this rank receives: recv_cnt[c] = what the rank in my direction opp(c) sends along ITS direction c.
 * ddcmi_plan_halo_layout: buffer layout (in beads) and message list of the per-step exchange -- remote
 * segments ordered by (peer rank, direction code) on both sides, so that each peer pair exchanges ONE
 * message: send_off/recv_off[28] = offset of direction c's segment (send: my direction; receive: the
 * SENDER's direction); msgs[0] = number of send messages, then {peer, offset, count} triples;
 * msgr likewise for the receives (room for 1 + 3*27 ints each).  loopback != 0: a single rank whose
 * periodic neighbours are itself exchanges with itself through the transport (test facility). */
int ddcmi_plan_recv_counts(int px, int py, int pz, int rank, int pbc, int loopback, const int *all_counts, int *recv_cnt);
int ddcmi_plan_halo_layout(int px, int py, int pz, int rank, int pbc, int loopback, const int send_cnt[27], const int recv_cnt[27],
                           int send_off[28], int recv_off[28], int *msgs, int *msgr);
/* host logic of the decomposition (domain.c:61-208 for a cubic lattice of domain
 * centres): destination rank and periodic shift of the 26 neighbour directions,
 * code = (dx+1)+3(dy+1)+9(dz+1); dest[27], shift[27*3]; dest = -1 where the box is open */
int ddcmi_plan_directions(int px, int py, int pz, int rank, int pbc, int *dest, int *shift);
/* in-process emulation of a px*py*pz decomposition (several contexts on one
 * device, halo/migration traffic by device copies): lets the whole multi-domain
 * path run on a single GPU.  Contexts of a group are driven only through these. */
int ddcmi_group_create(ddcmi_ctx **ctxs, int n, int px, int py, int pz);
int ddcmi_group_destroy(ddcmi_ctx **ctxs, int n);
int ddcmi_group_eval_forces(ddcmi_ctx **ctxs, int n);
int ddcmi_group_step_nglf(ddcmi_ctx **ctxs, int n, double dt, int nsteps);
/* ddcmi_group_temperatures for an in-process group (sums over its domains) */
int ddcmi_group_temperatures_all(ddcmi_ctx **ctxs, int n, double *Tgroup);
/* the lean step (a single domain of FREE beads without bonded terms: one launch per step, the second stage of its energy / virial /
 * kinetic sums formed for all pending steps at once): the sums of the steps of the last such launch, 32 doubles per step --
 * {lj, ele, virial xx yy zz xy xz yz} as the full list counts them (twice), {rk, tion xx yy zz xy xz yz}, 0, the bonded kernels'
 * {e_bond, e_angle, e_tors, e_impr, virial xx yy zz xy xz yz}, zeros.  Forms what is pending first.  sums: room for 32 steps. */
int ddcmi_debug_lean_history(ddcmi_ctx *ctx, int *nsteps, double *sums);
/* the displacement bound of the shell-limited walk: the word the reduction launches add to, and the lean steps' words since the rebuild (largest |v|^2 of each; ring[32]) */
int ddcmi_debug_disp(ddcmi_ctx *ctx, double *disp, float *ring, int *nring);

/* roctx ranges opened so far (DDCMI_ROCTX=1: MDSTEP, DDCENERGY, CHARMM_NONBOND ... named like ptiming.h's regions; 0 without the variable) */
long ddcmi_debug_roctx_ranges(void);

#ifdef __cplusplus
}
#endif
#endif
