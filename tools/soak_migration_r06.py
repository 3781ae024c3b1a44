#!/usr/bin/env python3
"""Soak of the decomposed rebuild under strong density changes (round 6, after the keep[] bug): a cube of liquid water in a box three times its size drifts
diagonally, one brick of a 2x2x2 (or other) grid per few rebuild periods, so every domain's bead count swings between zero and most of the system and
every array of the migration / halo path grows at some rebuild.  After every period: the bead set is whole, and ONE domain evaluating the gathered state gives the same forces and sums (1e-10).
   python3 tools/soak_migration_r06.py [periods] [n] [grid, e.g. 2,2,2] [water | water_langevin | lipid | lipid_npt]"""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ddcmd_amd.synth import make_water_setup
from ddcmd_amd.martini import MartiniGroup, MartiniHIP

periods = int(sys.argv[1]) if len(sys.argv) > 1 else 30
n = int(sys.argv[2]) if len(sys.argv) > 2 else 14
grid = tuple(int(x) for x in sys.argv[3].split(",")) if len(sys.argv) > 3 else (2, 2, 2)
workload = sys.argv[4] if len(sys.argv) > 4 else "water"
NPT = workload == "lipid_npt"      # the full nglfconstraint step: constraint groups across the bricks' faces, the barostat on the molecular pressure moving the box
if workload in ("lipid", "lipid_npt"):
    # the relaxed bilayer patch tiled n x n x 1 (charges, bonds, angles, dihedrals by gid: the gid -> slot tables and the term localisation swing with the beads),
    # vacuum above and below (box x 3 in z): the membrane drifts through the z bricks and slides along x and y
    from ddcmd_amd.deck import load_deck
    from ddcmd_amd.synth import replicate_setup
    deck = os.path.join(ROOT, "tests", "golden", "lipid_deck")
    extra = None
    if NPT:
        extra = ("TSTM RESIPARMS { constraintList = TSTM_cl0 TSTM_cl1; } TSTM_cl0 CONSLISTPARMS { constraintSubList = TSTM_c0 TSTM_c1 TSTM_c2; } "
                 "TSTM_cl1 CONSLISTPARMS { constraintSubList = TSTM_c3; } TSTM_c0 CONSPARMS { atomI=0; atomJ=1; func=1; r0=0.40 nm; } "
                 "TSTM_c1 CONSPARMS { atomI=1; atomJ=2; func=1; r0=0.40 nm; } TSTM_c2 CONSPARMS { atomI=0; atomJ=2; func=1; r0=0.655 nm; } "
                 "TSTM_c3 CONSPARMS { atomI=3; atomJ=4; func=1; r0=0.40 nm; } DPPC RESIPARMS { constraintList = DPPC_cl0; } "
                 "DPPC_cl0 CONSLISTPARMS { constraintSubList = DPPC_c0; } DPPC_c0 CONSPARMS { atomI=2; atomJ=3; func=1; r0=0.37 nm; } ")
    s = replicate_setup(load_deck(os.path.join(deck, "object_nvt.data"), restart_file=os.path.join(deck, "relaxed", "restart"), extra_objects=extra), (n, n, 1))
    if NPT:
        from ddcmd_amd.deck import units_convert
        s.npt_T, s.npt_P0 = units_convert(310.0, "K"), units_convert(1.0, "bar")
        s.npt_beta, s.npt_tau = units_convert(3.0e-4, "1/bar") * 20.0, units_convert(1.0, "ps")
    s.h = np.array(s.h, dtype=np.float64)
    bricks = np.array([s.h[0], s.h[4], 3.0 * s.h[8]]) / np.array(grid)
    s.h[8] *= 3.0
    drift = np.array([0.37, 0.23, 0.31]) * bricks / (int(s.updateRate) * s.dt)
else:
    s = make_water_setup(n, temperature_K=300.0)
    if workload == "water_langevin":      # LANGEVIN on the counter-based stream keyed by gid: the same noise on any decomposition
        from ddcmd_amd.deck import units_convert
        s.group_type = np.array([2], np.int32); s.group_Teq = np.array([units_convert(300.0, "K")]); s.group_tau = np.array([units_convert(1.0, "ps")])
        s.rng_seed = 20261003
    L = s.h[0]
    s.h = np.array(s.h, dtype=np.float64) * 3.0
    # 0.37 / 0.23 / 0.31 of a brick (1.5 L) per rebuild period along x / y / z: incommensurate, so the cube meets the brick faces in ever new ways
    drift = np.array([0.37, 0.23, 0.31]) * 1.5 * L / (int(s.updateRate) * s.dt)
period = int(s.updateRate)
s.vx = np.asarray(s.vx) + drift[0]; s.vy = np.asarray(s.vy) + drift[1]; s.vz = np.asarray(s.vz) + drift[2]
one = MartiniHIP(s)      # (evaluates gathered states: forces and sums need neither the constraint groups nor the barostat)
one.eval_forces()
traj = MartiniHIP(s, constraints=NPT)      # the same run on one domain, stepped alongside for the first periods (thermostat and all): before the two trajectories part as any two do
traj.eval_forces()
TRAJ_PERIODS = 10
class OneDomain(object):
    """grid 1,1,1: the plain one-domain context under the same checks (the lean step, images staged from their owners, the shell-limited walk under a
    large uniform drift) -- against a second context that evaluates the downloaded state afresh"""
    def __init__(self, s):
        self.m, self.n, self.s = MartiniHIP(s), 1, s
        self.step, self.energies, self.eval_forces = self.m.step, self.m.energies, self.m.eval_forces
    def gather(self):
        d = self.m.download()
        return {"gid": np.asarray(self.s.gid), "nlocal": [self.s.natoms], "r": d["r"], "v": d["v"], "f": d["f"]}
g = OneDomain(s) if grid == (1, 1, 1) else MartiniGroup(s, grid, constraints=NPT)
g.eval_forces()
gid0 = np.sort(np.asarray(s.gid))
assert np.array_equal(gid0, np.asarray(s.gid))      # (caller order = gid order: the gathered state uploads as it is)
worst = worst_f = worst_t = 0.0
seen_min, seen_max = [10 ** 9] * g.n, [0] * g.n
for p in range(periods):
    g.step(period)
    st = g.gather()
    assert sum(st["nlocal"]) == s.natoms and np.array_equal(st["gid"], gid0), (p, st["nlocal"])
    for r, c in enumerate(st["nlocal"]):
        seen_min[r] = min(seen_min[r], c); seen_max[r] = max(seen_max[r], c)
    # the decomposed run's state on ONE domain: the same forces and sums, exactly (no trajectory between the two to part)
    if NPT:      # the barostat has moved the box: the evaluating context takes the bricks' current one
        hb = np.array(s.h, dtype=np.float64); hb[[0, 4, 8]] = g.ranks[0].box()
        one._chk(one.lib.ddcmi_set_box(one.ctx, hb.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), int(s.pbc)))
    one.upload(st["r"][0], st["r"][1], st["r"][2], st["v"][0], st["v"][1], st["v"][2])
    ea, va = one.eval_forces()
    d = one.download()
    eb, vb, rkb, _ = g.energies()
    fmax = max(np.abs(d["f"][c]).max() for c in range(3))
    ferr = max(np.abs(st["f"][c] - d["f"][c]).max() for c in range(3)) / fmax
    err = max(max(abs(eb[k] - ea[k]) / max(abs(ea[k]), 1e-3 * abs(ea["total"])) for k in ("lj", "ele", "bond", "angle", "tors", "impr")), np.abs(vb - va).max() / np.abs(va).max())
    rk_np = 0.5 * float(np.sum(np.asarray(s.mass)[np.asarray(s.species)] * (st["v"][0] ** 2 + st["v"][1] ** 2 + st["v"][2] ** 2)))      # (caller order = gid order)
    err = max(err, abs(rkb - rk_np) / rk_np)
    terr = 0.0
    if p < TRAJ_PERIODS:
        traj.step(period)
        et, vt, rkt, _ = traj.energies()
        terr = max(abs(eb["total"] - et["total"]) / abs(et["total"]), abs(rkb - rkt) / rkt)
        assert terr < 1e-6, (p, terr)
        worst_t = max(worst_t, terr)
    worst, worst_f = max(worst, err), max(worst_f, ferr)
    print("period %3d  beads per domain %s  e_lj %.10g  against one domain: sums %.1e, forces %.1e%s" % (p + 1, st["nlocal"], eb["lj"], err, ferr, ", trajectory %.1e" % terr if p < TRAJ_PERIODS else ""), flush=True)
    if not (err < 1e-10 and ferr < 1e-10):
        print("   one domain:", {k: "%.12g" % v for k, v in ea.items()}, "virial", va)
        print("   the bricks:", {k: "%.12g" % v for k, v in eb.items()}, "virial", vb)
    assert err < 1e-10 and ferr < 1e-10, (p, err, ferr)
print("%d periods of %d steps, %d beads on %s bricks: every domain between %s and %s beads; against one domain evaluating the same state: sums %.1e, forces %.1e; against the one-domain run over the first %d periods: %.1e"
      % (periods, period, s.natoms, "x".join(map(str, grid)), seen_min, seen_max, worst, worst_f, TRAJ_PERIODS, worst_t))
