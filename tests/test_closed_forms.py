"""CPU: the oracle against numbers that do NOT come from the oracle -- energies of a few beads written straight from
the reference's formulas (tests/closed_forms.py) and their central differences: every bonded term kind of the lipid
deck's test molecule, and an analytic Lennard-Jones + reaction-field pair.  (The same checks run against the device
in tests/test_gpu_branches.py.)"""
import copy
import os
import numpy as np
import pytest

import pyoracle
import closed_forms as cf
from ddcmd_amd.deck import load_deck, units_convert
from ddcmd_amd.synth import make_water_setup

LIPID_DECK = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "lipid_deck", "object.data")
KINDS = ("bond", "angle", "tors", "impr")


def tstm_molecules(s):
    """[(bead indices in residue order)] of the deck's 5-bead test molecules, and their residue type"""
    rt = int(np.flatnonzero(np.asarray(s.resi_natoms) == 5)[0])
    idx = np.flatnonzero(s.resitype[s.species] == rt)
    order = idx[np.argsort(s.gid[idx], kind="stable")]
    return [order[5 * m:5 * m + 5] for m in range(order.size // 5)], rt


def conformers(x0, rng):
    """a gently distorted copy and a strongly distorted one of a 5-bead molecule (bohr)"""
    A = units_convert(1.0, "Angstrom")
    return [x0 + rng.uniform(-0.4, 0.4, x0.shape) * A, x0 + rng.uniform(-1.5, 1.5, x0.shape) * A]


def test_bonded_terms_against_closed_forms():
    s = load_deck(LIPID_DECK)
    s.excludePotentialTerm = 128            # bonded terms only (bioCharmmParms.h:25-28)
    mols, rt = tstm_molecules(s)
    assert len(mols) == 8
    rng = np.random.default_rng(11)
    o = pyoracle.Oracle(s)
    e0, _ = o.forces()
    m = mols[3]
    x0 = np.stack([o.rx[m], o.ry[m], o.rz[m]], axis=1)
    c0 = cf.molecule_terms_E(s, x0, rt)
    for x1 in conformers(x0, rng):
        o.rx[m], o.ry[m], o.rz[m] = x1[:, 0], x1[:, 1], x1[:, 2]
        e1, _ = o.forces()
        c1 = cf.molecule_terms_E(s, x1, rt)
        for k in KINDS:
            # the change of the system's energy of this kind is the change of this molecule's closed form
            assert abs((e1[k] - e0[k]) - (c1[k] - c0[k])) < 1e-11 * max(abs(e0[k]), 1e-3), k
        f_ref = cf.fd_forces(lambda x: sum(cf.molecule_terms_E(s, x, rt).values()), x1)
        f = np.stack([o.fx[m], o.fy[m], o.fz[m]], axis=1)
        assert np.abs(f - f_ref).max() < 2e-7 * np.abs(f_ref).max()


def charged_pair_setup(r_A, q=(1.0, -0.5)):
    """two charged beads at distance r along (1, 2, 2)/3 in a 32.5 A box"""
    s = make_water_setup(4)
    s.rmax = units_convert(9.0, "Angstrom")
    s.deltaR = units_convert(2.0, "Angstrom")
    from ddcmd_amd.synth import lj_shift
    s.shift = lj_shift(s.sigma, s.eps, s.rmax)
    s.charge = np.array(q, dtype=np.float64)
    # reaction field of a conducting medium (bioMartini.c:1234-1245), eps_r = 15
    lib_ke = 2.0 * units_convert(1.0, "Ry") * 1.0      # ke = e^2 / 4 pi eps0 = 2 Ry bohr in these units
    s.keR = lib_ke / 15.0
    s.krf = 0.5 / s.rmax ** 3
    s.crf = 1.5 / s.rmax
    r = units_convert(r_A, "Angstrom")
    d = np.array([1.0, 2.0, 2.0]) / 3.0
    keep = 2
    for name in ("rx", "ry", "rz", "vx", "vy", "vz"):
        setattr(s, name, np.zeros(keep))
    s.rx, s.ry, s.rz = np.array([0.0, -r * d[0]]) + 1.0, np.array([0.0, -r * d[1]]) - 2.0, np.array([0.0, -r * d[2]]) + 0.5
    s.gid = np.array([0, 1 << 32], dtype=np.uint64)
    s.species = np.array([0, 1], dtype=np.int32)
    s.group = np.zeros(2, np.int32)
    s.natoms = 2
    return s, r, d


@pytest.mark.parametrize("r_A", [4.3, 5.2, 8.9, 9.5])
def test_lj_reaction_field_pair_is_analytic(r_A):
    """one pair: energies, the force along the pair and the virial from the closed form (r = 9.5 A lies beyond the cutoff:
    only the self term -1/2 sum q^2 ke/eps_r crf remains, bioMartini.c:1030-1035)"""
    s, r, d = charged_pair_setup(r_A)
    o = pyoracle.Oracle(s)
    e, vir = o.forces()
    k = int(s.ljtype[1] + s.nlj * s.ljtype[0])
    kq = s.keR * s.charge[0] * s.charge[1]
    inside = r < s.rmax
    e_lj = 4 * s.eps[k] * ((s.sigma[k] / r) ** 12 - (s.sigma[k] / r) ** 6) + s.shift[k] if inside else 0.0
    e_pair = cf.lj_rf_pair_E(r, s.sigma[k], s.eps[k], s.shift[k], kq, s.krf, s.crf, s.rmax)
    e_self = -0.5 * s.keR * s.crf * float(np.sum(s.charge[s.species] ** 2))
    assert abs(e["lj"] - e_lj) < 1e-13 * max(abs(e_lj), 1e-6)
    assert abs(e["ele"] - ((e_pair - e_lj) + e_self)) < 1e-13 * abs(e_self)
    F = cf.lj_rf_pair_F(r, s.sigma[k], s.eps[k], kq, s.krf, s.rmax)
    f0 = np.array([o.fx[0], o.fy[0], o.fz[0]])
    assert np.abs(f0 - F * d).max() < 1e-12 * max(abs(F), 1e-9)
    assert np.abs(f0 + np.array([o.fx[1], o.fy[1], o.fz[1]])).max() < 1e-18 + 1e-15 * abs(F)
    # virial = f (x) d (bioMartini.c:1098-1103): xx yy zz xy xz yz
    dd = r * d
    want = F * np.array([d[0] * dd[0], d[1] * dd[1], d[2] * dd[2], d[0] * dd[1], d[0] * dd[2], d[1] * dd[2]])
    assert np.abs(vir - want).max() < 1e-12 * max(np.abs(want).max(), 1e-9)


def test_oracle_dihedral_series_branches():
    """planar dihedrals (|sin phi| <= 1e-8) drive the oracle's series arms (its census says so); at phi = 0 with
    delta = 0 / psi0 = 0 the series is the true derivative: forces equal central differences of the closed forms"""
    import ctypes
    from closed_forms import planar_chain
    s = load_deck(LIPID_DECK)
    s.excludePotentialTerm = 128
    mols, rt = tstm_molecules(s)
    t0 = int(s.tors_off[rt])
    s.tors_delta = np.array(s.tors_delta, dtype=np.float64)
    s.tors_delta[t0], s.tors_delta[t0 + 1], s.tors_delta[t0 + 2] = 0.0, 0.0, np.pi
    A = units_convert(1.0, "Angstrom")
    o = pyoracle.Oracle(s)
    shapes = [(7.5, 50.0, False, 0.0), (8.47, 40.0, False, 0.3), (9.0, 60.0, False, 0.0), (8.0, 35.0, False, 1.1),
              (7.5, 50.0, True, 0.0), (8.47, 40.0, True, 0.3), (9.0, 60.0, True, 0.0), (8.0, 65.0, True, 0.7)]
    for m, (chord, alpha, trans, tilt) in zip(mols, shapes):
        centre = np.array([o.rx[m].mean(), o.ry[m].mean(), o.rz[m].mean()])
        x = planar_chain(5, chord * A, alpha, trans, centre, tilt)
        o.rx[m], o.ry[m], o.rz[m] = x[:, 0], x[:, 1], x[:, 2]
    out = (ctypes.c_long * 8)()
    o.L.orc_branch_census(out, 1)
    o.forces()
    o.L.orc_branch_census(out, 1)
    c = list(out)
    assert c[0] > 0 and c[1] > 0 and c[2] > 0 and c[4] > 0, c
    for m in mols[:4]:
        x = np.stack([o.rx[m], o.ry[m], o.rz[m]], axis=1)
        fd = cf.fd_forces(lambda y: sum(cf.molecule_terms_E(s, y, rt).values()), x, h=1e-4)
        f = np.stack([o.fx[m], o.fy[m], o.fz[m]], axis=1)
        assert np.abs(f - fd).max() < 1e-5 * np.abs(fd).max()
