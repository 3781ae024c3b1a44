"""The device's constraint solve (k_constrain through ddcmi_step_nglf) against the reference's own linear solver: solve.c,
compiled where it lies into oracle/_ref (make -C oracle ref) and run in a child process (tests/ref_probe.py)."""
import os
import numpy as np
import pytest

import ddcmd_amd
from test_ref_pinned import REF, make_constraint_system, groups_of, res_move_cons

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not os.path.exists(REF), reason="oracle/_ref not built (make -C oracle ref needs /root/reference)")]


def constraint_setup(sysd, dt):
    """a Setup of the constraint molecules alone: one species per atom of every topology, one LJ type with eps = 0 (the pair
    kernel runs and finds no force), no bonded terms -- a step is kick, FRONT solve, drift, BACK solve and nothing else"""
    from ddcmd_amd.deck import Setup, GROUP_FREE
    s = Setup()
    tops = sysd["tops"]
    nt, nsp = len(tops), int(sysd["sp_off"][-1])
    s.dt, s.deltaR, s.updateRate, s.rmax, s.rcoulomb = dt, 5.0, 20, 20.0, 20.0
    s.nlj, s.sigma, s.eps, s.shift = 1, np.ones(1), np.zeros(1), np.zeros(1)
    s.nspecies = nsp
    s.species_name = ["T%02dxA%02d" % (t, a) for t, (na, _) in enumerate(tops) for a in range(na)]
    s.mass, s.charge = sysd["mass"].copy(), np.zeros(nsp)
    s.ljtype = np.zeros(nsp, np.int32)
    s.moltype = s.resitype = np.concatenate([np.full(na, t, np.int32) for t, (na, _) in enumerate(tops)])
    s.atomoffset = np.concatenate([np.arange(na, dtype=np.int32) for na, _ in tops])
    s.nmoltype = s.nresi = nt
    s.mol_nspecies = s.resi_natoms = np.array([na for na, _ in tops], np.int32)
    s.bpair_off = np.zeros(nt + 1, np.int32)
    s.bond_off = s.angle_off = s.tors_off = np.zeros(nt + 1, np.int32)
    s.cons_off = np.concatenate(([0], np.cumsum([len(pr) for _, pr in tops]))).astype(np.int32)
    s.consI = np.array([a for _, pr in tops for a, _b in pr], np.int32)
    s.consJ = np.array([b for _, pr in tops for _a, b in pr], np.int32)
    s.cons_grp = np.zeros(s.consI.size, np.int32)
    s.cons_r0 = np.concatenate(sysd["r0"])
    s.nresicons = nt
    s.ngroup, s.group_name = 1, ["group"]
    s.group_type, s.group_Teq, s.group_tau, s.group_interval = np.array([GROUP_FREE], np.int32), np.zeros(1), np.zeros(1), np.ones(1, np.int32)
    L = sysd["box"]
    s.h = np.array([L, 0, 0, 0, L, 0, 0, 0, L], dtype=np.float64)
    s.pbc = 7
    s.natoms = len(sysd["gid"])
    s.rx, s.ry, s.rz = (np.ascontiguousarray(sysd["r"][:, c]) for c in range(3))
    s.vx, s.vy, s.vz = (np.ascontiguousarray(sysd["v"][:, c]) for c in range(3))
    s.gid, s.species, s.group = sysd["gid"].copy(), sysd["species"].copy(), np.zeros(s.natoms, np.int32)
    return s


@pytest.mark.parametrize("seed", [3, 4])
def test_device_constraint_step_against_the_references_linear_solver(seed):
    """nglfconstraint.c:538-571 on molecules that feel no force: FRONT solve at r0, drift, BACK solve at r1.  The device sweeps
    Gauss-Seidel (resMoveConsOld, :180-264); the check is the reference's direct form (resMoveCons :266-312 +
    solveConstraintMatrix :139-175) with every linear system solved by the reference's own solve() (solve.c).  Groups of
    1..12 pairs: chains, rings, a triangle with a tail, a star; molecules across the periodic faces."""
    from ddcmd_amd.martini import MartiniHIP
    dt = 20.0
    sysd = make_constraint_system(seed, box=120.0, copies=4)
    s = constraint_setup(sysd, dt)
    m = MartiniHIP(s, constraints=True)
    e, _ = m.eval_forces()
    assert e["total"] == 0.0
    m.step(1)
    st = m.download()
    sweeps, bad = m.constraint_stats()
    assert bad == 0 and 1 < sweeps < 500
    m.close()
    # the reference's direct form
    g0 = groups_of(sysd, sysd["r"], sysd["v"])
    assert 2 <= res_move_cons(0, g0, dt) <= 30
    v1 = sysd["v"].copy()
    for g in g0:
        v1[g["idx"]] = g["V"]
    r1 = sysd["r"] + dt * v1
    g1 = groups_of(sysd, r1, v1)
    assert res_move_cons(1, g1, dt) == 1
    v2 = v1.copy()
    for g in g1:
        v2[g["idx"]] = g["V"]
    vdev = np.stack(st["v"], axis=1)
    rdev = np.stack(st["r"], axis=1)
    dr = rdev - r1
    dr -= sysd["box"] * np.rint(dr / sysd["box"])
    assert np.abs(vdev - v2).max() < 5e-10 * np.abs(sysd["v"]).max()
    assert np.abs(dr).max() < 1e-9
    for g in g1:      # every constrained pair has its length and no relative velocity along it
        for ab, (a, b) in enumerate(g["pairs"]):
            d = rdev[g["idx"][a]] - rdev[g["idx"][b]]
            d -= sysd["box"] * np.rint(d / sysd["box"])
            assert abs(np.linalg.norm(d) / g["dist"][ab] - 1.0) < 1e-10
            assert abs(np.dot(d, vdev[g["idx"][a]] - vdev[g["idx"][b]])) * dt / g["dist"][ab] ** 2 < 1e-10
