"""GPU tests (-m gpu) of the spatial decomposition: px*py*pz domains emulated in
one process on one GPU (device copies instead of RCCL) must reproduce the
single-domain run and the oracle: migration, halo tables, per-step halo refresh."""
import numpy as np
import pytest

import pyoracle
import ddcmd_amd
from ddcmd_amd.synth import make_water_setup
from conftest import rel_force_err
from ddcmd_amd.deck import units_convert

pytestmark = pytest.mark.gpu
TOL = 1e-6
TIGHT = 1e-10


@pytest.mark.parametrize("grid", [(2, 1, 1), (2, 2, 1), (2, 2, 2), (1, 1, 2)])
def test_decomposed_forces_match_oracle(grid):
    from ddcmd_amd.martini import MartiniGroup
    s = make_water_setup(15)            # 13.5k beads, box 122 A: 61 A bricks
    o = pyoracle.Oracle(s)
    e0, v0 = o.forces()
    g = MartiniGroup(s, grid)
    e, vir = g.eval_forces()
    st = g.gather()
    assert np.array_equal(st["gid"], np.sort(s.gid))
    assert sum(st["nlocal"]) == s.natoms
    assert rel_force_err(st["f"], (o.fx, o.fy, o.fz)) < TIGHT
    assert abs(e["lj"] - e0["lj"]) < TIGHT * abs(e0["lj"])
    assert np.abs(vir - v0).max() < TIGHT * np.abs(v0).max()
    g.close()


def test_decomposed_trajectory_with_migration():
    """45 steps across two rebuilds: beads migrate between the 8 domains; state and energies follow the oracle"""
    from ddcmd_amd.martini import MartiniGroup
    s = make_water_setup(15)
    o = pyoracle.Oracle(s)
    o.forces()
    g = MartiniGroup(s, (2, 2, 2))
    g.eval_forces()
    n0 = list(g.gather()["nlocal"])
    for block in range(3):
        eo, vo, rko, _ = o.step(15)
        g.step(15)
        e, vir, rk, _ = g.energies()
        assert abs(e["total"] - eo["total"]) < TOL * abs(eo["total"]), block
        assert abs(rk - rko) < TOL * rko
        assert np.abs(vir - vo).max() < TOL * np.abs(vo).max()
    st = g.gather()
    assert sum(st["nlocal"]) == s.natoms and np.array_equal(st["gid"], np.sort(s.gid))
    assert st["nlocal"] != n0                     # ownership really changed
    assert rel_force_err(st["f"], (o.fx, o.fy, o.fz)) < TOL
    L = s.h[0]
    for c, ref in enumerate((o.rx, o.ry, o.rz)):
        dr = st["r"][c] - ref
        dr -= L * np.rint(dr / L)
        assert np.abs(dr).max() < 1e-8
    for c, ref in enumerate((o.vx, o.vy, o.vz)):
        assert np.abs(st["v"][c] - ref).max() < 1e-8 * np.abs(ref).max()
    g.close()


def test_decomposed_langevin_on_migrating_lcg64_streams():
    """RANDOM type LCG64 under a decomposition: the particles' stream records ride in the migration records (random.c:72-76
    registers parmsArray with the particle exchange) and through every sort, so 8 domains draw what one domain draws --
    45 steps across two rebuilds with beads changing owners follow the oracle, and every stream ends in the oracle's state"""
    from ddcmd_amd.martini import MartiniGroup
    from ddcmd_amd.deck import units_convert
    s = make_water_setup(15)
    s.group_type = np.array([2], np.int32)
    s.group_Teq = np.array([units_convert(310.0, "K")])
    s.group_tau = np.array([units_convert(0.3, "ps")])
    s.lcg64 = pyoracle.lcg64_default(s.gid)
    o = pyoracle.Oracle(s)
    assert o.lcg is not None
    o.forces()
    g = MartiniGroup(s, (2, 2, 2))
    g.eval_forces()
    st0 = g.gather()
    by_gid = np.argsort(s.gid, kind="stable")
    assert (st0["lcg64"] == s.lcg64[by_gid]).all()
    for block in range(3):
        eo, vo, rko, _ = o.step(15)
        g.step(15)
        e, vir, rk, _ = g.energies()
        assert abs(e["total"] - eo["total"]) < TOL * abs(eo["total"]), block
        assert abs(rk - rko) < TOL * rko, block
    st = g.gather()
    assert st["nlocal"] != st0["nlocal"]                     # ownership really changed
    assert (st["lcg64"]["state"] == o.lcg["state"][by_gid]).all()
    assert (st["lcg64"]["prime"] == s.lcg64["prime"][by_gid]).all() and (st["lcg64"]["multID"] == s.lcg64["multID"][by_gid]).all()
    for c, ref in enumerate((o.vx, o.vy, o.vz)):
        assert np.abs(st["v"][c] - ref[by_gid]).max() < 1e-8 * np.abs(ref).max()
    g.close()


def test_decomposed_charged_system():
    """charges: per-rank self term and reaction field across domain faces"""
    from ddcmd_amd.martini import MartiniGroup
    s = make_water_setup(12)
    s.nspecies = 4
    s.species_name = ["WxW", "WFxWF", "QPxQP", "QMxQM"]
    s.mass = np.array([s.mass[0]] * 4)
    s.charge = np.array([0.0, 0.0, 1.0, -1.0])
    s.ljtype = np.array([1, 0, 1, 0], np.int32)
    s.moltype = np.array([0, 1, 2, 3], np.int32)
    s.resitype = np.array([0, 1, 0, 1], np.int32)
    s.atomoffset = np.zeros(4, np.int32)
    s.nmoltype = 4
    s.mol_nspecies = np.ones(4, np.int32)
    s.bpair_off = np.zeros(5, np.int32)
    pick = np.random.RandomState(3).rand(s.natoms)
    s.species = np.where(pick < 0.1, 2, np.where(pick < 0.2, 3, s.species)).astype(np.int32)
    o = pyoracle.Oracle(s)
    e0, v0 = o.forces()
    g = MartiniGroup(s, (2, 2, 1))
    e, vir = g.eval_forces()
    st = g.gather()
    assert rel_force_err(st["f"], (o.fx, o.fy, o.fz)) < TIGHT
    assert abs(e["ele"] - e0["ele"]) < TIGHT * abs(e0["ele"])
    assert abs(e["lj"] - e0["lj"]) < TIGHT * abs(e0["lj"])
    g.close()


import os
LIPID_DECK = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "lipid_deck", "object.data")
KINDS = ("lj", "ele", "bond", "angle", "tors", "impr", "total")


def test_bonded_terms_by_gid_single_domain():
    """terms named by gid (the decomposed-run interface) on one domain == caller-order indices == oracle"""
    from ddcmd_amd.martini import MartiniHIP
    from ddcmd_amd.deck import load_deck
    s = load_deck(LIPID_DECK)
    o = pyoracle.Oracle(s)
    e0, v0 = o.forces()
    m = MartiniHIP(s, bonded_by_gid=True)
    e, vir = m.eval_forces()
    d = m.download()
    assert rel_force_err(d["f"], (o.fx, o.fy, o.fz)) < 1e-9
    for k in KINDS:
        assert abs(e[k] - e0[k]) < 1e-9 * max(abs(e0[k]), 1e-12), k
    assert np.abs(vir - v0).max() < 1e-9 * np.abs(v0).max()
    m.close()


@pytest.mark.parametrize("grid", [(1, 1, 2), (2, 1, 1), (1, 2, 2), (2, 2, 2)])
def test_decomposed_lipid_deck_all_terms(grid):
    """bonds/angles/dihedrals/impropers that straddle domain faces: each rank evaluates the terms
    touching its beads from halo positions, forces only on its own beads, energy/virial weighted"""
    from ddcmd_amd.martini import MartiniGroup
    from ddcmd_amd.deck import load_deck
    s = load_deck(LIPID_DECK)
    o = pyoracle.Oracle(s)
    e0, v0 = o.forces()
    g = MartiniGroup(s, grid)
    g.eval_forces()
    e, vir, _, _ = g.energies()
    st = g.gather()
    assert sum(st["nlocal"]) == s.natoms
    assert rel_force_err(st["f"], (o.fx, o.fy, o.fz)) < 1e-9
    for k in KINDS:
        assert abs(e[k] - e0[k]) < 1e-9 * max(abs(e0[k]), 1e-12), k
    assert np.abs(vir - v0).max() < 1e-9 * np.abs(v0).max()
    g.close()


def test_decomposed_lipid_deck_steps():
    """30 NGLF steps (3 rebuilds, lipids migrate bead by bead) on 2x2x2 domains follow the oracle"""
    from ddcmd_amd.martini import MartiniGroup
    from ddcmd_amd.deck import load_deck
    s = load_deck(LIPID_DECK)
    o = pyoracle.Oracle(s)
    o.forces()
    g = MartiniGroup(s, (2, 2, 2))
    g.eval_forces()
    for block in range(3):
        eo, vo, rko, _ = o.step(10)
        g.step(10)
        e, vir, rk, _ = g.energies()
        for k in ("lj", "ele", "bond", "angle", "tors", "impr"):
            assert abs(e[k] - eo[k]) < TOL * max(abs(eo[k]), abs(eo["total"]) * 1e-3), (block, k)
        assert abs(rk - rko) < TOL * rko
        assert np.abs(vir - vo).max() < TOL * np.abs(vo).max()
    st = g.gather()
    assert rel_force_err(st["f"], (o.fx, o.fy, o.fz)) < TOL
    g.close()


def test_decomposed_lipid_nvt_berendsen():
    """relaxed 310 K lipid restart, Berendsen group: the group temperature is summed over the domains,
    so 2x2x2 domains scale velocities exactly like one domain and like the oracle"""
    from ddcmd_amd.martini import MartiniGroup
    from ddcmd_amd.deck import load_deck
    deck = os.path.dirname(LIPID_DECK)
    s = load_deck(os.path.join(deck, "object_nvt.data"), restart_file=os.path.join(deck, "relaxed", "restart"))
    assert int(s.group_type[0]) == 1
    o = pyoracle.Oracle(s)
    o.forces()
    o.group_temperature()
    g = MartiniGroup(s, (2, 2, 2))
    g.eval_forces()
    Tg = g.group_temperatures()
    assert abs(Tg[0] - o.groups[0].temperature) < 1e-10 * Tg[0]
    for block in range(3):
        eo, vo, rko, _ = o.step(10)
        g.step(10)
        e, vir, rk, _ = g.energies()
        assert abs(rk - rko) < TOL * rko, block
        assert abs(e["total"] - eo["total"]) < TOL * abs(eo["total"]), block
        o.group_temperature()
        Tg = g.group_temperatures()
        assert abs(Tg[0] - o.groups[0].temperature) < 1e-8 * Tg[0]
    g.close()


def test_decomposed_displacement_triggered_rebuilds():
    """updateRate = 0 on 2x2x1 domains: any domain whose beads moved half the skin triggers the
    rebuild of all (check4updateNeighbor, ddcUpdateAll.c:56); the trajectory follows the oracle"""
    from ddcmd_amd.martini import MartiniGroup
    from ddcmd_amd.deck import units_convert
    s = make_water_setup(15)
    s.updateRate = 0
    s.deltaR = units_convert(2.0, "Angstrom")
    o = pyoracle.Oracle(s)
    o.forces()
    g = MartiniGroup(s, (2, 2, 1))
    g.eval_forces()
    for block in range(4):
        eo, vo, rko, _ = o.step(10)
        g.step(10)
        e, vir, rk, _ = g.energies()
        assert abs(e["total"] - eo["total"]) < TOL * abs(eo["total"]), block
        assert abs(rk - rko) < TOL * rko
    assert g.ranks[0].list_stats()["rebuilds"] >= 3
    g.close()


def test_decomposed_restraints():
    """restrained beads are located on whichever domain owns them after each migration"""
    from ddcmd_amd.martini import MartiniGroup
    from ddcmd_amd.deck import load_deck
    x = ("system SYSTEM { potential = martini restraintPot; } restraintPot POTENTIAL { type = RESTRAINT; parmfile = restraint.data; }")
    s = load_deck(LIPID_DECK, extra_objects=x)
    o = pyoracle.Oracle(s)
    e0, v0 = o.forces()
    g = MartiniGroup(s, (2, 2, 2))
    g.eval_forces()
    e, vir, _, _ = g.energies()
    st = g.gather()
    assert rel_force_err(st["f"], (o.fx, o.fy, o.fz)) < 1e-9
    assert abs(e["restraint"] - e0["restraint"]) < 1e-10 * e0["restraint"]
    assert np.abs(vir - v0).max() < 1e-9 * np.abs(v0).max()
    eo, vo, rko, _ = o.step(20)
    g.step(20)
    e, vir, rk, _ = g.energies()
    assert abs(e["restraint"] - eo["restraint"]) < TOL * eo["restraint"]
    assert abs(e["total"] - eo["total"]) < TOL * abs(eo["total"])
    g.close()


def test_index_based_npt_and_constraints_are_refused_on_several_domains(monkeypatch):
    """the barostat / velocity constraints given by caller-order indices and set BEFORE the decomposition must not slip
    through: each rank would scale its own box from its local virial.  The gid forms (what MartiniRank uses)
    carry the cross-domain sums and are accepted: test_decomposed_barostat_... below"""
    import ctypes
    from ddcmd_amd.martini import MartiniHIP, MartiniRank, _declare_domains
    s = make_water_setup(10)
    s.npt_T, s.npt_P0, s.npt_beta, s.npt_tau = 1e-3, 0.0, 1e-3, 1000.0
    a, b = MartiniHIP(s, upload=False, test_api=True), MartiniHIP(s, upload=False, test_api=True)          # molecule lists by index (contexts of libddcmi_test.so: groups are test API)
    _declare_domains(a.lib)
    arr = (ctypes.c_void_p * 2)(a.ctx, b.ctx)
    assert a.lib.ddcmi_group_create(arr, 2, 2, 1, 1) == -4                   # DDCMI_EUNSUPPORTED
    assert b"named by gid" in a.lib.ddcmi_last_error(a.ctx)
    a.lib.ddcmi_group_destroy(arr, 2)
    a.close(); b.close()
    # RCCL path, one rank in loopback mode is still ONE domain: accepted
    monkeypatch.setenv("DDCMI_RCCL_LOOPBACK", "1")
    m = MartiniRank(s, np.arange(s.natoms))
    buf = ctypes.create_string_buffer(128)
    assert m.lib.ddcmi_comm_unique_id(buf) == 0
    m.comm_init(0, 1, buf.raw, (1, 1, 1))
    m.close()


def _relaxed_lipid(extra=None):
    from ddcmd_amd.deck import load_deck
    deck = os.path.dirname(LIPID_DECK)
    return load_deck(os.path.join(deck, "object_nvt.data"), restart_file=os.path.join(deck, "relaxed", "restart"), extra_objects=extra)


def _check_constrained_state(s, st, o, pi_gid, pj_gid, dd, box, tag):
    """st: gathered state ordered by gid; constrained pairs keep their length and have no relative velocity along the pair"""
    order = np.argsort(np.asarray(s.gid, dtype=np.uint64), kind="stable")          # caller order -> gid order
    rank_of = np.empty(s.natoms, np.int64)
    rank_of[order] = np.arange(s.natoms)
    pi, pj = rank_of[pi_gid], rank_of[pj_gid]
    r, v = st["r"], st["v"]
    d = np.stack([r[c][pi] - r[c][pj] for c in range(3)], axis=1)
    d -= box * np.rint(d / box)
    assert np.abs(np.sqrt((d * d).sum(axis=1)) / dd - 1.0).max() < 1e-10, tag
    w = np.stack([v[c][pi] - v[c][pj] for c in range(3)], axis=1)
    assert np.abs((d * w).sum(axis=1) * s.dt / dd ** 2).max() < 1e-10, tag
    for c, (ro, vo_) in enumerate(((o.rx, o.vx), (o.ry, o.vy), (o.rz, o.vz))):
        dr = r[c] - ro[order]
        dr -= box[c] * np.rint(dr / box[c])
        assert np.abs(dr).max() < 1e-7 * box[c], (tag, c)
        assert np.abs(v[c] - vo_[order]).max() < 1e-6 * np.abs(vo_).max(), (tag, c)


@pytest.mark.parametrize("grid", [(2, 1, 1), (1, 2, 2), (2, 2, 2)])
def test_decomposed_velocity_constraints_match_oracle(grid):
    """nglfconstraint's velocity constraints across domain faces (nglfconstraint.c:180-264, 510-574): groups named by gid,
    every rank that owns an atom of a group solves the whole group with the partners' positions from the position halo and
    their velocities from the velocity halo (exchanged before the FRONT and before the BACK solve), and keeps its own atoms"""
    from ddcmd_amd.martini import MartiniGroup, expand_constraints
    from test_oracle import CONSTRAINT_X
    s = _relaxed_lipid(CONSTRAINT_X)
    po, pi, pj, dd = expand_constraints(s)
    o = pyoracle.Oracle(s, constraints=True)
    o.forces()
    g = MartiniGroup(s, grid, constraints=True)
    g.eval_forces()
    o.group_temperature()
    g.group_temperatures()
    box = s.box
    for block in range(4):
        eo, vo, rko, tio = o.step(5)
        g.step(5 if block % 2 else 1)
        if block % 2 == 0:
            g.step(4)
        o.group_temperature()
        g.group_temperatures()
        e, vir, rk, tion = g.energies()
        assert abs(e["total"] - eo["total"]) < TOL * abs(eo["total"]), block
        assert abs(rk - rko) < TOL * rko, block
        assert np.abs(vir - vo).max() < TOL * np.abs(vo).max(), block
        assert np.abs(tion - tio).max() < TOL * np.abs(tio).max(), block
        st = g.gather()
        assert np.array_equal(st["gid"], np.sort(s.gid))
        _check_constrained_state(s, st, o, pi, pj, dd, box, block)
    worst = [r.constraint_stats() for r in g.ranks]
    assert all(bad == 0 for _, bad in worst) and max(sw for sw, _ in worst) > 1
    g.close()


@pytest.mark.parametrize("grid", [(2, 1, 1), (1, 1, 2), (2, 2, 2)])
def test_decomposed_barostat_with_molecular_virial_and_constraints(grid):
    """the full nglfconstraint step on several domains: the barostat acts on the MOLECULAR pressure, whose intramolecular
    term needs every molecule's centre of mass and total force -- lipids reach across domain faces, so the ranks' partial
    sums of such split molecules are all-reduced with the virial every step (molecularPressure.c:22-67); every domain
    scales its box with the same factors"""
    from ddcmd_amd.martini import MartiniGroup
    from test_oracle import CONSTRAINT_X
    s = _relaxed_lipid(CONSTRAINT_X)
    T = units_convert(310.0, "K")
    P0 = units_convert(1.0, "bar")
    beta = units_convert(3.0e-4, "1/bar") * 20.0
    tau = units_convert(1.0, "ps")
    s.npt_T, s.npt_P0, s.npt_beta, s.npt_tau = T, P0, beta, tau
    o = pyoracle.Oracle(s, constraints=True)
    o.forces()
    g = MartiniGroup(s, grid, constraints=True)
    g.eval_forces()
    o.group_temperature()
    g.group_temperatures()
    L0 = g.ranks[0].box().copy()
    nsplit = None
    for block in range(3):
        eo, vo, rko, _ = o.step_npt(5, T, P0, beta, tau, molecular=True)
        g.step(5 if block % 2 else 2)
        if block % 2 == 0:
            g.step(3)
        o.group_temperature()
        g.group_temperatures()
        e, vir, rk, _ = g.energies()
        for r in g.ranks:                # every domain holds the same pressure and the same box
            assert np.abs(r.barostat_pressure() - o.pmol).max() < 1e-8 * np.abs(o.pmol).max(), block
            assert np.abs(r.box() - o.box).max() < 1e-10 * o.box.max(), block
        assert abs(e["total"] - eo["total"]) < TOL * abs(eo["total"]), block
        assert abs(rk - rko) < TOL * rko
        assert np.abs(vir - vo).max() < TOL * np.abs(vo).max()
    assert np.abs(g.ranks[0].box() - L0).max() > 1e-5 * L0.max()
    g.close()


@pytest.mark.parametrize("vscale", [3.0, 0.02])
def test_decomposed_rows_end_at_the_last_shell_that_can_matter(monkeypatch, vscale):
    """the shell-limited walk of the pair kernel was a single-domain feature.  A decomposed rank now bounds its pair
    distances by D_own + max(D_own, H): D_own = sum over the steps of max |dt v| of its OWNED beads, H = the largest distance of a
    RECEIVED halo bead from its place at the rebuild, measured by the halo update where the bead arrives.  2x2x2 emulated domains
    with the walk cut short and with the full walk (DDCMI_NO_SHELL_SKIP) agree bit for bit over two rebuild periods with migration,
    hot (large D) and nearly frozen (half of every row skipped), and follow the oracle."""
    from ddcmd_amd.martini import MartiniGroup
    s = make_water_setup(15)
    s.vx, s.vy, s.vz = (np.asarray(v) * vscale for v in (s.vx, s.vy, s.vz))
    monkeypatch.delenv("DDCMI_NO_SHELL_SKIP", raising=False)
    a = MartiniGroup(s, (2, 2, 2))
    monkeypatch.setenv("DDCMI_NO_SHELL_SKIP", "1")
    b = MartiniGroup(s, (2, 2, 2))
    monkeypatch.delenv("DDCMI_NO_SHELL_SKIP", raising=False)
    o = pyoracle.Oracle(s)
    o.forces()
    a.eval_forces(); b.eval_forces()
    for block, n in enumerate((17, 20, 8)):
        a.step(n); b.step(n)
        eo, vo, rko, _ = o.step(n)
        sa, sb = a.gather(), b.gather()
        assert np.array_equal(sa["gid"], sb["gid"])
        for k in ("r", "v", "f"):
            for c in range(3):
                assert np.array_equal(sa[k][c], sb[k][c]), (block, k, c)
        ea, _, rka, _ = a.energies()
        eb, _, rkb, _ = b.energies()
        assert ea["total"] == eb["total"] and rka == rkb
        assert abs(ea["total"] - eo["total"]) < TOL * abs(eo["total"]) and abs(rka - rko) < TOL * max(rko, 1e-300)
    a.close(); b.close()


def test_a_domain_grows_its_arrays_in_the_middle_of_a_migration():
    """2x1x1 bricks, free faces in y.  Three groups of water beads more than a list radius apart in y: a slab (half the beads) that fills the LEFT brick and
    drifts one brick to the right per rebuild period; a thin layer at the top of the RIGHT brick that drifts with it (so it LEAVES the right brick while
    the slab arrives); a thin layer at the bottom of the right brick that stays.  At the first rebuild after the start the right domain -- sized for its
    two thin layers plus headroom -- receives several times its bead count, loses some and keeps some: its arrays must grow between the phase that
    decides who leaves (keep[]) and the phase that compacts by that decision.  Until round 6 keep[] was grown WITHOUT its contents (tools/fuzz_abi.py
    met it as a GPU memory fault).  Uniform drifts change no pair distance inside a group, so the run equals the same system on one domain."""
    from ddcmd_amd.martini import MartiniGroup, MartiniHIP
    s = make_water_setup(16, temperature_K=300.0)
    L = s.h[0]
    y = np.asarray(s.ry)
    slab, top, bottom = np.abs(y) < 0.25 * L, y > 0.44 * L, y < -0.44 * L
    sel = np.flatnonzero(slab | top | bottom)
    for a in ("rx", "ry", "rz", "vx", "vy", "vz", "species", "group", "gid"):
        setattr(s, a, np.ascontiguousarray(np.asarray(getattr(s, a))[sel]))
    s.natoms = int(sel.size)
    slab, top, bottom = slab[sel], top[sel], bottom[sel]
    assert (0.25 + 0.06) * L + (s.rmax + s.deltaR) < 0.44 * L          # the groups do not see each other
    s.h = np.array(s.h, dtype=np.float64)
    s.h[0] = 2.0 * L                                                   # bricks of the old box's width
    s.pbc = 5                                                          # free faces in y
    s.rx = np.asarray(s.rx) - 0.5 * L                                  # everything in the left brick [-L, 0) ...
    s.rx[top | bottom] += L                                            # ... but for the two thin layers
    period = int(s.updateRate)
    s.vx = np.asarray(s.vx).copy()
    s.vx[slab | top] += L / (period * s.dt)                            # one brick per rebuild period
    one = MartiniHIP(s)
    e1, _ = one.eval_forces()
    g = MartiniGroup(s, (2, 1, 1))
    e0, _ = g.eval_forces()
    n0 = list(g.gather()["nlocal"])
    assert n0 == [int(slab.sum()), int(top.sum() + bottom.sum())] and n0[0] + n0[1] // 2 > 1.3 * n0[1] + 5000, n0
    assert abs(e0["lj"] - e1["lj"]) < TIGHT * abs(e1["lj"])
    for block in range(2):
        one.step(period + 2)
        g.step(period + 2)                                # across the rebuild at which the slab and the top layer have changed bricks
        st = g.gather()
        assert sum(st["nlocal"]) == s.natoms and np.array_equal(st["gid"], np.sort(s.gid)), (block, st["nlocal"])
        ea, va, rka, _ = one.energies()
        eb, vb, rkb, _ = g.energies()
        assert abs(eb["lj"] - ea["lj"]) < 1e-9 * abs(ea["lj"]) and abs(rkb - rka) < 1e-9 * rka, block
        assert np.abs(vb - va).max() < 1e-8 * np.abs(va).max(), block
        if block == 0:
            assert abs(st["nlocal"][1] - (int(slab.sum()) + int(bottom.sum()))) < 0.02 * s.natoms, st["nlocal"]      # slab + the layer that stayed
    d = one.download()
    order = np.argsort(np.asarray(s.gid), kind="stable")
    for c in range(3):
        dv = st["v"][c] - d["v"][c][order]
        assert np.abs(dv).max() < 1e-9 * np.abs(d["v"][c]).max()
    one.close()
    g.close()


def test_a_drifting_cube_of_water_swings_every_domain_between_empty_and_full():
    """tools/soak_migration_r06.py, short: a cube of liquid in a box three times its size drifts diagonally across 2x2x2 bricks; after every rebuild
    period the bead set is whole and ONE domain evaluating the gathered state gives the same forces and sums (1e-10; measured 1e-15)"""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "soak_migration_r06.py"), "16", "12", "2,2,2"], cwd=root, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "16 periods of" in r.stdout, r.stdout[-1500:] + r.stderr[-1500:]


def test_a_drifting_membrane_empties_and_fills_the_domains_and_their_bonded_sums():
    """the same with the relaxed bilayer patch (tiled 2x2, vacuum above and below, Berendsen thermostat) drifting through the z bricks: the gid -> slot tables
    and the term localisation swing with the beads.  Until round 6 a domain that had just been EMPTIED kept reporting the bonded energies and virial of its
    last bonded launch (the run's bond energy 14 % too large in period 5 of this very run; forces were never affected): the sums of every period equal ONE
    domain evaluating the gathered state, and the first ten periods equal the one-domain run step for step"""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "soak_migration_r06.py"), "14", "2", "2,2,2", "lipid"], cwd=root, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "14 periods of" in r.stdout, r.stdout[-1500:] + r.stderr[-1500:]
