"""GPU tests (-m gpu): edge cases of the boundary (open boundaries, slabs, non-cubic
boxes, vacuum regions, beads on the box faces) against the oracle, and
size-independent properties at BASELINE.json's full sizes (1 M and 4 M beads)."""
import numpy as np
import pytest

import pyoracle
import ddcmd_amd
from ddcmd_amd.deck import units_convert
from ddcmd_amd.synth import make_water_setup
from conftest import rel_force_err

pytestmark = pytest.mark.gpu
TIGHT = 1e-10
TOL = 1e-6


def _compare(s, steps=0):
    from ddcmd_amd.martini import MartiniHIP
    o = pyoracle.Oracle(s)
    e0, v0 = o.forces()
    m = MartiniHIP(s)
    e, vir = m.eval_forces()
    f = m.download()["f"]
    assert rel_force_err(f, (o.fx, o.fy, o.fz)) < TIGHT
    assert abs(e["total"] - e0["total"]) < TIGHT * max(abs(e0["total"]), 1e-12)
    assert np.abs(vir - v0).max() < TIGHT * max(np.abs(v0).max(), 1e-12)
    if steps:
        eo, vo, rko, _ = o.step(steps)
        m.step(steps)
        e, vir, rk, _ = m.energies()
        assert abs(e["total"] - eo["total"]) < TOL * abs(eo["total"])
        assert abs(rk - rko) < TOL * max(rko, 1e-12)
    m.close()


@pytest.mark.parametrize("pbc", [0, 3, 4, 5, 6])
def test_open_and_mixed_boundaries(pbc):
    """pbc bitmask (box.c:56-67): open axes get no images; the oracle reduces only periodic axes"""
    s = make_water_setup(8)
    s.pbc = pbc
    # open axes: widen the box so there is vacuum around the beads and nothing leaves it
    s.h = s.h.copy()
    for a in range(3):
        if not (pbc >> a) & 1:
            s.h[4 * a] *= 1.4
    _compare(s, steps=25)


def test_non_cubic_box_and_vacuum():
    """orthorhombic 1 : 1.5 : 2.5 box, water only in a slab: most cells and whole tiles are empty"""
    s = make_water_setup(8)
    L = s.h[0]
    s.h = np.array([L, 0, 0, 0, 1.5 * L, 0, 0, 0, 2.5 * L])
    s.ry = s.ry * 1.0
    _compare(s, steps=25)


def test_beads_on_box_faces_and_outside():
    """positions exactly on +-L/2 and slightly outside the box (callers hand over unwrapped beads)"""
    s = make_water_setup(8)
    L = s.h[0]
    s.rx = s.rx.copy(); s.ry = s.ry.copy(); s.rz = s.rz.copy()
    s.rx[0] = 0.5 * L
    s.ry[1] = -0.5 * L
    s.rz[2] = 0.5 * L + 0.37          # one lattice reduction brings it back (backInBox_fast)
    s.rx[3] = -0.5 * L - 0.11
    o = pyoracle.Oracle(s)
    o.L.orc_back_in_box(__import__("ctypes").byref(o.p), o.n, pyoracle._d(o.rx), pyoracle._d(o.ry), pyoracle._d(o.rz))
    e0, v0 = o.forces()
    from ddcmd_amd.martini import MartiniHIP
    m = MartiniHIP(s)
    e, vir = m.eval_forces()
    f = m.download()["f"]
    assert rel_force_err(f, (o.fx, o.fy, o.fz)) < TIGHT
    assert abs(e["total"] - e0["total"]) < TIGHT * abs(e0["total"])
    m.close()


def test_single_bead_and_isolated_beads():
    """no neighbours at all: zero forces and energies, nothing crashes"""
    from ddcmd_amd.martini import MartiniHIP
    s = make_water_setup(8)
    keep = np.array([0, 17, 300])          # far apart on the lattice? make sure: move them
    for k in ("rx", "ry", "rz", "vx", "vy", "vz"):
        setattr(s, k, getattr(s, k)[keep].copy())
    s.rx[:] = [-40.0, 0.0, 40.0]; s.ry[:] = 0.0; s.rz[:] = 0.0      # 21 A apart, also through the periodic faces
    s.gid = s.gid[keep].copy(); s.species = s.species[keep].copy(); s.group = s.group[keep].copy()
    s.natoms = 3
    m = MartiniHIP(s)
    e, vir = m.eval_forces()
    f = m.download()["f"]
    assert e["total"] == 0.0 and np.abs(vir).max() == 0.0 and max(np.abs(c).max() for c in f) == 0.0
    m.step(3)
    assert m.list_stats()["entries"] == 0
    m.close()


@pytest.mark.parametrize("n,nsteps", [(64, 40), (102, 40)])
def test_full_size_properties(n, nsteps):
    """1.05 M and 4.24 M beads (the headline box, bench.py HEADLINE_N): Newton's third law, symmetric list, energy conservation over two
    rebuilds, kinetic tensor trace = 2 rk -- properties that need no oracle run"""
    from ddcmd_amd.martini import MartiniHIP
    s = make_water_setup(n)
    m = MartiniHIP(s)
    e0, vir0 = m.eval_forces()
    rk0, tion0 = m.kinetic()
    f = m.download(4)["f"]
    fmax = max(np.abs(c).max() for c in f)
    for c in f:
        assert abs(c.sum()) < 1e-9 * fmax * np.sqrt(s.natoms)
    st = m.list_stats()
    assert st["entries"] % 2 == 0
    dens = s.natoms / s.volume
    expect = 4.0 / 3.0 * np.pi * (s.rmax + s.deltaR) ** 3 * dens
    assert abs(st["entries"] / s.natoms / expect - 1) < 0.07          # ~128 stored neighbours per bead (134 on the initial lattice)
    assert abs((tion0[0] + tion0[1] + tion0[2]) - 2 * rk0) < 1e-12 * rk0
    etot0 = e0["total"] + rk0
    m.step(nsteps)
    e1, vir1, rk1, tion1 = m.energies()
    assert m.list_stats()["rebuilds"] == 1 + nsteps // 20
    # velocity-Verlet at dt = 20 fs from a relaxing lattice: the O(dt^2) energy error stays below 2 % of E_kin
    assert abs(e1["total"] + rk1 - etot0) < 0.02 * rk1
    T = 2 * rk1 / (3 * s.natoms) / units_convert(1.0, "K")
    assert 150.0 < T < 450.0
    # momentum conservation: centre-of-mass velocity stays zero
    v = m.download(2)["v"]
    mass = s.mass[s.species]
    for c in v:
        assert abs(np.sum(mass * c)) < 1e-9 * np.sum(mass * np.abs(c))
    m.close()


def test_decomposition_consistent_at_1M():
    """1.05 M beads on 8 emulated domains == single domain (energies, ownership) after 25 steps"""
    from ddcmd_amd.martini import MartiniHIP, MartiniGroup
    s = make_water_setup(64)
    m = MartiniHIP(s)
    m.eval_forces()
    m.step(25)
    e1, v1, rk1, _ = m.energies()
    m.close()
    g = MartiniGroup(s, (2, 2, 2))
    g.eval_forces()
    g.step(25)
    e8, v8, rk8, _ = g.energies()
    assert abs(e8["total"] - e1["total"]) < 1e-9 * abs(e1["total"])
    assert abs(rk8 - rk1) < 1e-9 * rk1
    assert np.abs(v8 - v1).max() < 1e-9 * np.abs(v1).max()
    assert sum(int(r.lib.ddcmi_nlocal(r.ctx)) for r in g.ranks) == s.natoms
    g.close()


@pytest.fixture(scope="module")
def water_4M_single():
    """the headline box (FCC n = 102: 4 244 832 beads) after 45 steps on one domain: energies, kinetic energy, virial"""
    from ddcmd_amd.martini import MartiniHIP
    s = make_water_setup(102)
    m = MartiniHIP(s)
    m.eval_forces()
    m.step(45)
    e1, v1, rk1, _ = m.energies()
    m.close()
    return s, e1, v1, rk1


@pytest.mark.parametrize("grid", [(2, 1, 1), (2, 2, 1), (2, 2, 2)])
def test_bench_decompositions_at_4M(water_4M_single, grid):
    """the bricks bench.py --gpus 2/4/8 uses, on the 4.24 M-bead headline box itself (emulated domains on one GPU):
    same energies, kinetic energy and virial as the single domain after 45 steps (3 rebuilds with migration),
    no bead lost -- the buffer sizes, halo tables and migration of the multi-GPU runs at their real sizes"""
    from ddcmd_amd.martini import MartiniGroup
    s, e1, v1, rk1 = water_4M_single
    g = MartiniGroup(s, grid)
    g.eval_forces()
    g.step(45)
    e, v, rk, _ = g.energies()
    assert abs(e["total"] - e1["total"]) < 1e-9 * abs(e1["total"])
    assert abs(rk - rk1) < 1e-9 * rk1
    assert np.abs(v - v1).max() < 1e-9 * np.abs(v1).max()
    assert sum(int(r.lib.ddcmi_nlocal(r.ctx)) for r in g.ranks) == s.natoms
    g.close()


def _copies_vs_oracle(reps, decomposed, base=25):
    """The 62.5 k-bead water box (BASELINE configs[1], the size the oracle runs every step in seconds; base = 17: 19.7 k beads, whose
    6x6x6 tiling is the n = 102 headline box) tiled reps times is the same
    periodic system at the size the bench times: EVERY copy of EVERY bead must feel the force the oracle computes for the small
    box, energies and virial scale with the copy count, and after 25 steps -- across the rebuild at step 20 -- every copy sits
    where the oracle's bead sits, with its velocity."""
    import pyoracle
    from ddcmd_amd.synth import replicate_setup
    from ddcmd_amd.martini import MartiniHIP, MartiniGroup
    s0 = make_water_setup(base)
    o = pyoracle.Oracle(s0)
    e0, v0 = o.forces()
    ref = np.stack([o.fx, o.fy, o.fz])[:, None, :].copy()
    ncopy = reps[0] * reps[1] * reps[2]
    s = replicate_setup(s0, reps)
    assert s.natoms == s0.natoms * ncopy
    m = MartiniGroup(s, (2, 2, 2)) if decomposed else MartiniHIP(s)
    e, vir = m.eval_forces()
    d = m.gather() if decomposed else m.download()
    f = np.stack(d["f"]).reshape(3, ncopy, s0.natoms)
    assert np.abs(f - ref).max() < 1e-10 * np.abs(ref).max()          # per bead, per copy
    assert abs(e["lj"] - ncopy * e0["lj"]) < 1e-10 * ncopy * abs(e0["lj"])
    assert abs(e["total"] - ncopy * e0["total"]) < 1e-10 * ncopy * abs(e0["total"])
    assert np.abs(vir - ncopy * v0).max() < 1e-10 * ncopy * np.abs(v0).max()
    eo, vo, rko, _ = o.step(25)
    m.step(25)
    e2, vir2, rk, _ = m.energies()
    assert abs(rk - ncopy * rko) < 1e-9 * ncopy * rko
    assert abs(e2["total"] - ncopy * eo["total"]) < 1e-9 * ncopy * abs(eo["total"])
    assert np.abs(vir2 - ncopy * vo).max() < 1e-9 * ncopy * np.abs(vo).max()
    d = m.gather() if decomposed else m.download()
    v = np.stack(d["v"]).reshape(3, ncopy, s0.natoms)
    vref = np.stack([o.vx, o.vy, o.vz])[:, None, :]
    assert np.abs(v - vref).max() < 1e-8 * np.abs(vref).max()
    f = np.stack(d["f"]).reshape(3, ncopy, s0.natoms)
    fref = np.stack([o.fx, o.fy, o.fz])[:, None, :]
    assert np.abs(f - fref).max() < 1e-7 * np.abs(fref).max()
    # positions: copy (ix, iy, iz) is the oracle's bead shifted by whole small boxes
    L0 = np.array([s0.h[0], s0.h[4], s0.h[8]])
    r = np.stack(d["r"]).reshape(3, ncopy, s0.natoms)
    for c, oref in enumerate((o.rx, o.ry, o.rz)):
        dr = r[c] - oref[None, :] - L0[c] * (0.5 - 0.5 * reps[c])      # (replicate_setup centres the tiled box on 0 again)
        dr -= L0[c] * np.rint(dr / L0[c])
        assert np.abs(dr).max() < 1e-9 * L0[c]
    if not decomposed:
        assert m.list_stats()["rebuilds"] == 2
    m.close()


@pytest.mark.parametrize("reps,base", [((6, 6, 6), 17), ((4, 2, 2), 25)])
def test_water_at_bench_sizes_per_bead_against_the_oracle(reps, base):
    """4 244 832 beads (the headline config: the n = 17 box tiled 6x6x6 has exactly the n = 102 box's edge and bead count) and 1.0 M beads (configs[2]) on one domain"""
    _copies_vs_oracle(reps, False, base)


def test_water_4M_on_eight_domains_per_bead_against_the_oracle():
    """the same at the headline size on the 2x2x2 bricks of bench.py --gpus 8 (emulated domains on one GPU)"""
    _copies_vs_oracle((6, 6, 6), True, 17)


def test_lipid_bilayer_2M_beads_periodic_copies():
    """BASELINE config 5 size: the lipid deck tiled 12x12x6 (2.04M beads, ~0.9M bonded terms).
    A periodic box repeated is the same system: every copy of a bead must feel the force the
    oracle computes for the original 2363-bead deck, and energies scale with the copy count."""
    import os
    import pyoracle
    from ddcmd_amd.deck import load_deck
    from ddcmd_amd.synth import replicate_setup
    from ddcmd_amd.martini import MartiniHIP
    from conftest import rel_force_err
    deck = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "lipid_deck")
    # the 310 K restart (make_lipid_relaxed.py): lipids there wrap around the box edges, so the
    # tiling has to keep molecules whole; Berendsen group as in BASELINE config 5
    s0 = load_deck(os.path.join(deck, "object_nvt.data"), restart_file=os.path.join(deck, "relaxed", "restart"))
    o = pyoracle.Oracle(s0)
    e0, v0 = o.forces()
    reps = (12, 12, 6)
    ncopy = reps[0] * reps[1] * reps[2]
    s = replicate_setup(s0, reps)
    assert s.natoms == 2363 * ncopy
    m = MartiniHIP(s)
    e, vir = m.eval_forces()
    d = m.download()
    f = np.stack(d["f"]).reshape(3, ncopy, s0.natoms)
    ref = np.stack([o.fx, o.fy, o.fz])[:, None, :]
    assert np.abs(f - ref).max() < 1e-8 * np.abs(ref).max()
    for k in ("lj", "ele", "bond", "angle", "tors", "impr", "total"):
        assert abs(e[k] - ncopy * e0[k]) < 1e-9 * ncopy * max(abs(e0[k]), abs(e0["total"]) * 1e-6), k
    assert np.abs(vir - ncopy * v0).max() < 1e-9 * ncopy * np.abs(v0).max()
    # 20 steps (two rebuilds) with the Berendsen group active track the oracle of the small deck
    o.group_temperature()
    m.group_temperatures()
    eo, vo, rko, _ = o.step(20)
    m.step(20)
    e2, _, rk, _ = m.energies()
    assert abs(rk - ncopy * rko) < 1e-6 * ncopy * rko
    assert abs(e2["total"] - ncopy * eo["total"]) < 1e-6 * ncopy * abs(eo["total"])
    m.close()


def test_lipid_bilayer_2M_on_eight_domains_per_copy_against_the_oracle():
    """BASELINE config 5 in its decomposed form at full size (VERDICT r5 #1a): the 12x12x6 tiling (2.04 M beads, 1.17 M bonds,
    0.85 M angles, dihedrals, charges) on the 2x2x2 bricks of an 8-GPU run (emulated domains on one GPU: 255 k beads per brick,
    lipids straddling every face, terms named by gid, the gid -> slot table at 0.9 M terms per rank).  EVERY copy of EVERY bead
    must feel the force the oracle computes for the 2363-bead deck, energies by kind and the virial scale with the copy count;
    then 20 Berendsen steps -- two rebuilds with bead-by-bead migration of lipids across the faces (the reference moves whole
    residues with their centre atom, bioMartiniRule.c:64-205; results are the same) -- track the oracle
    of the deck, and no bead is lost."""
    import os
    import pyoracle
    from ddcmd_amd.deck import load_deck
    from ddcmd_amd.synth import replicate_setup
    from ddcmd_amd.martini import MartiniGroup
    deck = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "lipid_deck")
    s0 = load_deck(os.path.join(deck, "object_nvt.data"), restart_file=os.path.join(deck, "relaxed", "restart"))
    o = pyoracle.Oracle(s0)
    e0, v0 = o.forces()
    reps = (12, 12, 6)
    ncopy = reps[0] * reps[1] * reps[2]
    s = replicate_setup(s0, reps)
    assert s.natoms == 2363 * ncopy
    g = MartiniGroup(s, (2, 2, 2))
    e, vir = g.eval_forces()
    st = g.gather()          # ordered by gid: copy k's beads are one block (molecule ids offset by k * nmol), inside it the deck's beads by gid
    assert sum(st["nlocal"]) == s.natoms and min(st["nlocal"]) > 0.9 * s.natoms / 8
    by_gid = np.argsort(np.asarray(s0.gid, dtype=np.uint64), kind="stable")
    f = np.stack(st["f"]).reshape(3, ncopy, s0.natoms)
    ref = np.stack([o.fx, o.fy, o.fz])[:, None, by_gid]
    assert np.abs(f - ref).max() < 1e-8 * np.abs(ref).max()
    for k in ("lj", "ele", "bond", "angle", "tors", "impr", "total"):
        assert abs(e[k] - ncopy * e0[k]) < 1e-9 * ncopy * max(abs(e0[k]), abs(e0["total"]) * 1e-6), k
    assert np.abs(vir - ncopy * v0).max() < 1e-9 * ncopy * np.abs(v0).max()
    own0 = np.sort(g.ranks[0].download_particles()["gid"])
    o.group_temperature()
    Tg = g.group_temperatures()
    assert abs(Tg[0] - o.groups[0].temperature) < 1e-9 * Tg[0]
    eo, vo, rko, _ = o.step(20)
    g.step(20)
    e2, vir2, rk, _ = g.energies()
    assert abs(rk - ncopy * rko) < 1e-6 * ncopy * rko
    assert abs(e2["total"] - ncopy * eo["total"]) < 1e-6 * ncopy * abs(eo["total"])
    for k in ("lj", "ele", "bond", "angle", "tors", "impr"):
        assert abs(e2[k] - ncopy * eo[k]) < 1e-6 * ncopy * max(abs(eo[k]), abs(eo["total"]) * 1e-3), k
    assert np.abs(vir2 - ncopy * vo).max() < 1e-6 * ncopy * np.abs(vo).max()
    st = g.gather()
    assert sum(st["nlocal"]) == s.natoms
    # beads really changed owners (the brick faces lie on copy boundaries, so by symmetry as many enter as leave: compare WHO is there)
    own1 = np.sort(g.ranks[0].download_particles()["gid"])
    moved = np.setdiff1d(own1, own0).size
    assert moved > 100, moved
    assert np.array_equal(st["gid"], np.sort(np.asarray(s.gid, dtype=np.uint64)))
    v = np.stack(st["v"]).reshape(3, ncopy, s0.natoms)
    vref = np.stack([o.vx, o.vy, o.vz])[:, None, by_gid]
    assert np.abs(v - vref).max() < 1e-7 * np.abs(vref).max()
    f = np.stack(st["f"]).reshape(3, ncopy, s0.natoms)
    fref = np.stack([o.fx, o.fy, o.fz])[:, None, by_gid]
    assert np.abs(f - fref).max() < 1e-6 * np.abs(fref).max()
    g.close()


def test_tiled_bilayer_follows_the_oracle_for_a_thousand_steps():
    """long-trajectory parity: exact periodic copies move in lockstep, so a 4x4x2 tiling of the lipid deck (75.6 k beads, every
    bonded kind, Berendsen, 100 list rebuilds) must reproduce the oracle's run of the single deck -- the same thermostat
    history, the same temperature excursions -- until chaos separates them (1e-14 at the start grows to ~1e-4 by step 1500)"""
    import os
    from ddcmd_amd.deck import load_deck
    from ddcmd_amd.synth import replicate_setup
    from ddcmd_amd.martini import MartiniHIP
    deck = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "lipid_deck")
    s0 = load_deck(os.path.join(deck, "object_nvt.data"), restart_file=os.path.join(deck, "relaxed", "restart"))
    reps = (4, 4, 2)
    ncopy = reps[0] * reps[1] * reps[2]
    o = pyoracle.Oracle(s0)
    o.forces(); o.group_temperature()
    m = MartiniHIP(replicate_setup(s0, reps))
    m.eval_forces(); m.group_temperatures()
    for blk, tol in ((1, 1e-7), (2, 2e-5)):
        for _ in range(25):          # the Berendsen group reads the temperature published at the print cadence: every 20 steps here
            eo, _, rko, _ = o.step(20); o.group_temperature()
            m.step(20); m.group_temperatures()
        e, _, rk, _ = m.energies()
        assert abs(rk - ncopy * rko) < tol * ncopy * rko, (blk, rk / ncopy, rko)
        assert abs(e["total"] - ncopy * eo["total"]) < tol * ncopy * abs(eo["total"]), (blk, e["total"] / ncopy, eo["total"])
    m.close()


@pytest.mark.parametrize("ntypes", [12, 20, 26])
def test_many_lj_types(ntypes):
    """9..16 LJ types: packed entries without the shift bit (the shifted-copy flag rides in bit 0 of the partner's staged z);
    > 16 types: bare 16-bit slots, the partner's class in bits 1-8 of its staged z.  Charged beads on top.  Every pair of types
    has its own random sigma and eps here: 78 / 210 distinct table entries (two-level table where it saves a workgroup per CU) and,
    at 26 types, 351 -- more than the level table's byte indexes: the direct table."""
    from ddcmd_amd.martini import MartiniHIP, MartiniGroup
    s = make_water_setup(10)
    rs = np.random.RandomState(7)
    s.nspecies = ntypes
    s.species_name = ["S%dxA" % k for k in range(ntypes)]
    s.mass = np.full(ntypes, s.mass[0])
    s.charge = np.where(np.arange(ntypes) % 5 == 0, 0.5, np.where(np.arange(ntypes) % 5 == 1, -0.5, 0.0))
    s.ljtype = np.arange(ntypes, dtype=np.int32)
    s.moltype = np.arange(ntypes, dtype=np.int32)
    s.resitype = np.zeros(ntypes, np.int32)
    s.atomoffset = np.zeros(ntypes, np.int32)
    s.nmoltype = ntypes
    s.mol_nspecies = np.ones(ntypes, np.int32)
    s.bpair_off = np.zeros(ntypes + 1, np.int32)
    s.nlj = ntypes
    sig0, eps0 = s.sigma[0], s.eps[0]
    sg = sig0 * (0.9 + 0.2 * rs.rand(ntypes, ntypes)); sg = 0.5 * (sg + sg.T)
    ep = eps0 * (0.6 + 0.8 * rs.rand(ntypes, ntypes)); ep = 0.5 * (ep + ep.T)
    from ddcmd_amd.synth import lj_shift
    s.sigma, s.eps = sg.ravel(), ep.ravel()
    s.shift = np.array([lj_shift(a, b, s.rmax) for a, b in zip(s.sigma, s.eps)])
    s.species = rs.randint(0, ntypes, s.natoms).astype(np.int32)
    o = pyoracle.Oracle(s)
    e0, v0 = o.forces()
    for make in (lambda: MartiniHIP(s), lambda: MartiniGroup(s, (2, 1, 1))):
        m = make()
        m.eval_forces()
        e, vir = m.energies()[:2]
        f = m.download()["f"] if hasattr(m, "download") else m.gather()["f"]
        assert rel_force_err(f, (o.fx, o.fy, o.fz)) < 1e-10
        for k in ("lj", "ele", "total"):
            assert abs(e[k] - e0[k]) < 1e-10 * abs(e0[k]), (ntypes, k)
        assert np.abs(vir - v0).max() < 1e-10 * np.abs(v0).max()
        m.close()


def test_lipid_bilayer_2M_beads_with_forty_bead_types():
    """VERDICT r4 #3, the type-count cliff: martiniLJ_parms builds an nspecies^2 table (bioMartini.c:868-950) and a real Martini
    deck has ~40 bead types, where the decks here have 2-6.  The 2.04 M-bead bilayer tiling with every LJ type split into copies
    of itself -- 40 LJ types, 48 (type, charge) classes, every bead's copy drawn at random (relabel_types) -- is the same physics
    bead for bead: every copy of every bead must feel the force the oracle computes for the ORIGINAL 2363-bead deck.  The direct
    class-pair table would be 74 KB of LDS (one workgroup per CU, or none beside a denser neighbourhood); the kernel keeps one byte
    per class pair + the dozen distinct entries instead (k_nonbond<..., LVL>) and two workgroups per CU."""
    import os
    import pyoracle
    from ddcmd_amd.deck import load_deck
    from ddcmd_amd.synth import replicate_setup, relabel_types
    from ddcmd_amd.martini import MartiniHIP
    deck = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "lipid_deck")
    s0 = load_deck(os.path.join(deck, "object_nvt.data"), restart_file=os.path.join(deck, "relaxed", "restart"))
    o = pyoracle.Oracle(s0)
    e0, v0 = o.forces()
    reps = (12, 12, 6)
    ncopy = reps[0] * reps[1] * reps[2]
    s = relabel_types(replicate_setup(s0, reps), 40)
    assert s.nlj == 40 and s.natoms == 2363 * ncopy
    classes = set(zip(s.ljtype[s.species].tolist(), s.charge[s.species].tolist()))
    assert len(classes) >= 44
    m = MartiniHIP(s)
    e, vir = m.eval_forces()
    d = m.download()
    f = np.stack(d["f"]).reshape(3, ncopy, s0.natoms)
    ref = np.stack([o.fx, o.fy, o.fz])[:, None, :]
    assert np.abs(f - ref).max() < 1e-8 * np.abs(ref).max()
    for k in ("lj", "ele", "bond", "angle", "tors", "impr", "total"):
        assert abs(e[k] - ncopy * e0[k]) < 1e-9 * ncopy * max(abs(e0[k]), abs(e0["total"]) * 1e-6), k
    assert np.abs(vir - ncopy * v0).max() < 1e-9 * ncopy * np.abs(v0).max()
    o.group_temperature()
    m.group_temperatures()
    eo, vo, rko, _ = o.step(20)
    m.step(20)
    e2, _, rk, _ = m.energies()
    assert abs(rk - ncopy * rko) < 1e-6 * ncopy * rko
    assert abs(e2["total"] - ncopy * eo["total"]) < 1e-6 * ncopy * abs(eo["total"])
    m.close()


@pytest.mark.parametrize("workload", ["water", "lipid", "types12", "lipid40"])
def test_pair_table_in_two_levels_equals_the_direct_table(monkeypatch, workload):
    """k_nonbond<..., LVL> against the direct class-pair table on systems where both fit (DDCMI_FORCE_LEVEL_TABLE=1 takes the level
    form regardless): the same entries reach the same arithmetic, so forces, energies and a 25-step trajectory agree in every bit --
    packed entries with and without the shift bit, bare entries, charges, bonded terms, the fused step"""
    import os
    from ddcmd_amd.deck import load_deck
    from ddcmd_amd.synth import relabel_types
    from ddcmd_amd.martini import MartiniHIP
    deck = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "lipid_deck")
    if workload == "water":
        s = make_water_setup(12)
    elif workload == "lipid":
        s = load_deck(os.path.join(deck, "object_nvt.data"), restart_file=os.path.join(deck, "relaxed", "restart"))
    elif workload == "lipid40":
        s = relabel_types(load_deck(os.path.join(deck, "object_nvt.data"), restart_file=os.path.join(deck, "relaxed", "restart")), 40)
    else:
        s = relabel_types(make_water_setup(12), 12)
    out = []
    for force in (False, True):
        if force:
            monkeypatch.setenv("DDCMI_FORCE_LEVEL_TABLE", "1")
        else:
            monkeypatch.delenv("DDCMI_FORCE_LEVEL_TABLE", raising=False)
        m = MartiniHIP(s)
        e, vir = m.eval_forces()
        f0 = np.stack(m.download()["f"])
        m.group_temperatures()
        m.step(25)
        e2, vir2, rk, _ = m.energies()
        d = m.download()
        out.append((e["total"], e["lj"], e["ele"], vir.copy(), f0, e2["total"], rk, np.stack(d["r"]), np.stack(d["v"])))
        m.close()
    monkeypatch.delenv("DDCMI_FORCE_LEVEL_TABLE", raising=False)
    a, b = out
    for x, y in zip(a, b):
        assert np.array_equal(np.asarray(x), np.asarray(y))
    if workload in ("lipid", "lipid40"):
        o = pyoracle.Oracle(s)
        eo, _ = o.forces()
        assert abs(a[0] - eo["total"]) < 1e-10 * abs(eo["total"])
