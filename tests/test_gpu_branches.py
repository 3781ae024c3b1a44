"""GPU tests (-m gpu) that harden parity where the oracle alone cannot (Next #5):
  * the dihedral code's series branches (|sin phi| <= 1e-8; delta ~ 0, ~ pi, other; improper) and the 2 pi wrap of
    the improper difference, driven on purpose -- the device's branch census proves they ran -- against the oracle,
    and against closed forms where the reference's series is the true derivative (phi = 0 with delta = 0 or pi);
  * the device's forces against central differences of the DEVICE's own energy (forcetest.c:10-88 method) --
    independent of the oracle;
  * the device's virial against -dE/dV under uniform scaling (testPressure, masters.c:134-202);
  * the closed forms of tests/closed_forms.py (bonded term kinds, analytic LJ + reaction-field pair)."""
import ctypes
import os
import numpy as np
import pytest

import pyoracle
import closed_forms as cf
from closed_forms import planar_chain
from ddcmd_amd.deck import load_deck, units_convert
from ddcmd_amd.martini import MartiniHIP
from conftest import rel_force_err
from test_closed_forms import tstm_molecules, conformers, charged_pair_setup, LIPID_DECK, KINDS

pytestmark = pytest.mark.gpu


def census(m, reset=True):
    out = (ctypes.c_ulonglong * 8)()
    m.lib.ddcmi_debug_branch_census.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
    assert m.lib.ddcmi_debug_branch_census(out, 1 if reset else 0) == 0
    return [int(x) for x in out]


def test_dihedral_series_branches_are_driven_and_match():
    s = load_deck(LIPID_DECK)
    s.excludePotentialTerm = 128                       # bonded terms only
    mols, rt = tstm_molecules(s)
    t0 = int(s.tors_off[rt])
    # the test molecule's dihedrals: d0 proper n=2, d1 improper, d2 proper n=3 -> delta = 0, psi0 = 0, delta = pi
    assert [int(f) for f in s.tors_func[t0:t0 + 3]] == [1, 2, 1]
    s.tors_delta = np.array(s.tors_delta, dtype=np.float64)
    s.tors_delta[t0], s.tors_delta[t0 + 1], s.tors_delta[t0 + 2] = 0.0, 0.0, np.pi
    A = units_convert(1.0, "Angstrom")
    L = s.box
    o = pyoracle.Oracle(s)
    # molecules 0-3 planar all-cis (phi = 0), 4-7 planar all-trans (phi = pi): several sizes and orientations, because
    # whether cos phi lands exactly on +-1 (the series branch) or one ulp inside (the plain branch, same value) is a
    # matter of rounding; long bonds keep bioDihedralFast's 1e-12 regulariser below one ulp of g1 g2
    shapes = [(7.5, 50.0, False, 0.0), (8.47, 40.0, False, 0.3), (9.0, 60.0, False, 0.0), (8.0, 35.0, False, 1.1),
              (7.5, 50.0, True, 0.0), (8.47, 40.0, True, 0.3), (9.0, 60.0, True, 0.0), (8.0, 65.0, True, 0.7)]
    for m, (chord, alpha, trans, tilt) in zip(mols, shapes):
        centre = np.array([o.rx[m].mean(), o.ry[m].mean(), o.rz[m].mean()])
        x = planar_chain(5, chord * A, alpha, trans, centre, tilt)
        o.rx[m], o.ry[m], o.rz[m] = x[:, 0], x[:, 1], x[:, 2]
    s.rx, s.ry, s.rz = o.rx.copy(), o.ry.copy(), o.rz.copy()
    e0, v0 = o.forces()
    dev = MartiniHIP(s, test_api=True)
    census(dev)
    e, vir = dev.eval_forces()
    c = census(dev)
    d = dev.download()
    assert c[0] > 0 and c[1] > 0 and c[2] > 0 and c[4] > 0, "series branches not reached: %s" % c
    # (planar dihedrals are the singular point of bioDihedralFast -- at phi = pi the truncated series is a quotient of
    # two nearly cancelling sums -- so the last digits are amplified: 1e-7 here against 1e-9 on ordinary geometries)
    assert rel_force_err(d["f"], (o.fx, o.fy, o.fz)) < 1e-7
    for k in KINDS:          # (acos next to +-1: one ulp of cos phi is 1e-8 of phi)
        assert abs(e[k] - e0[k]) < 1e-7 * max(abs(e0[k]), 1e-12), k
    assert np.abs(vir - v0).max() < 1e-7 * np.abs(v0).max()
    # phi = 0 with delta = 0 (proper n=2) and psi0 = 0 (improper): there the series IS the derivative -- the cis
    # molecules' forces equal central differences of the closed-form energies
    for m in mols[:4]:
        x = np.stack([s.rx[m], s.ry[m], s.rz[m]], axis=1)
        fd = cf.fd_forces(lambda y: sum(cf.molecule_terms_E(s, y, rt).values()), x, h=1e-4)
        f = np.stack([d["f"][0][m], d["f"][1][m], d["f"][2][m]], axis=1)
        assert np.abs(f - fd).max() < 1e-5 * np.abs(fd).max()
    # a third parameter set: delta neither 0 nor pi in the series branch (the reference's "else" arm), and an improper
    # whose difference psi - psi0 must be wrapped by 2 pi
    s.tors_delta[t0], s.tors_delta[t0 + 1] = 0.6, 3.0
    for m in mols[4:]:                                  # trans molecules: psi = +-pi, psi0 = 3.0 -> |psi - psi0| may exceed pi
        s.rz[m] = s.rz[m] + np.array([0.0, 0.02, -0.03, 0.05, -0.02]) * A      # slightly out of plane, both signs of psi occur
    s.ry[mols[5]] = 2 * s.ry[mols[5]].mean() - s.ry[mols[5]]                    # mirror image: the other sign
    o2 = pyoracle.Oracle(s)
    e0, v0 = o2.forces()
    dev.close()
    dev = MartiniHIP(s, test_api=True)
    census(dev)
    e, vir = dev.eval_forces()
    c = census(dev)
    d = dev.download()
    assert c[3] > 0 and c[5] > 0, "other-delta series arm / improper wrap not reached: %s" % c
    assert rel_force_err(d["f"], (o2.fx, o2.fy, o2.fz)) < 1e-7
    for k in KINDS:
        assert abs(e[k] - e0[k]) < 1e-7 * max(abs(e0[k]), 1e-12), k
    dev.close()


def _fd_check(s, atoms, h, tol):
    dev = MartiniHIP(s, test_api=True)
    dev.eval_forces()
    f0 = np.stack(dev.download()["f"])
    fmax = np.abs(f0).max()
    pos = [np.array(s.rx, dtype=np.float64), np.array(s.ry, dtype=np.float64), np.array(s.rz, dtype=np.float64)]
    worst = 0.0
    for i in atoms:
        for c in range(3):
            keep = pos[c][i]
            pos[c][i] = keep + h
            dev.upload(pos[0], pos[1], pos[2], s.vx, s.vy, s.vz)
            ep = dev.eval_forces()[0]["total"]
            pos[c][i] = keep - h
            dev.upload(pos[0], pos[1], pos[2], s.vx, s.vy, s.vz)
            em = dev.eval_forces()[0]["total"]
            pos[c][i] = keep
            worst = max(worst, abs(-(ep - em) / (2 * h) - f0[c, i]) / fmax)
    dev.close()
    assert worst < tol, worst


def test_device_forces_are_minus_the_gradient_of_the_device_energy():
    """forcetest.c:10-88 on the HIP path itself, full lipid potential (LJ, reaction field incl. excluded pairs, bonds,
    three angle kinds, proper + improper dihedrals): no oracle involved"""
    s = load_deck(LIPID_DECK)
    tst = np.flatnonzero(np.array([s.species_name[k].startswith("TSTM") for k in s.species]))[:10]
    lip = np.array([0, 1, 2, 3, 7, 11, 300, 1700])      # lipid beads (charged head, glycerol, tails) and two waters
    _fd_check(s, np.concatenate((tst, lip)), 1e-4, 2e-7)


def test_device_virial_is_the_volume_derivative_of_the_device_energy():
    """testPressure (masters.c:134-202): under r -> lambda r, h -> lambda h: dE/dlambda = -tr(virial).  With the deck's shifted LJ
    and the conducting reaction field both pair energies vanish at the cutoff, so no surface term remains"""
    for name, s in (("water", None), ("lipid", load_deck(LIPID_DECK))):
        if s is None:
            from ddcmd_amd.synth import make_water_setup
            s = make_water_setup(6)
        dev = MartiniHIP(s, test_api=True)
        e, vir = dev.eval_forces()
        dev.close()
        tr = vir[0] + vir[1] + vir[2]

        def scaled(lam):
            import copy
            s2 = copy.copy(s)
            s2.h = np.asarray(s.h) * lam
            s2.rx, s2.ry, s2.rz = np.asarray(s.rx) * lam, np.asarray(s.ry) * lam, np.asarray(s.rz) * lam
            d2 = MartiniHIP(s2, test_api=True)
            en = d2.eval_forces()[0]["total"]
            d2.close()
            return en
        dl = 2e-6
        dEdl = (scaled(1 + dl) - scaled(1 - dl)) / (2 * dl)
        assert abs(dEdl + tr) < 2e-6 * abs(tr), (name, dEdl, tr)


def test_device_against_closed_forms():
    """the CPU test of tests/test_closed_forms.py with the device in the oracle's place"""
    s = load_deck(LIPID_DECK)
    s.excludePotentialTerm = 128
    mols, rt = tstm_molecules(s)
    rng = np.random.default_rng(11)
    dev = MartiniHIP(s, test_api=True)
    e0, _ = dev.eval_forces()
    m = mols[3]
    x0 = np.stack([s.rx[m], s.ry[m], s.rz[m]], axis=1)
    c0 = cf.molecule_terms_E(s, x0, rt)
    pos = [np.array(s.rx), np.array(s.ry), np.array(s.rz)]
    for x1 in conformers(x0, rng):
        for c in range(3):
            pos[c][m] = x1[:, c]
        dev.upload(pos[0], pos[1], pos[2], s.vx, s.vy, s.vz)
        e1, _ = dev.eval_forces()
        c1 = cf.molecule_terms_E(s, x1, rt)
        for k in KINDS:
            assert abs((e1[k] - e0[k]) - (c1[k] - c0[k])) < 1e-10 * max(abs(e0[k]), 1e-3), k
        f_ref = cf.fd_forces(lambda x: sum(cf.molecule_terms_E(s, x, rt).values()), x1)
        f = np.stack([dev.download()["f"][c][m] for c in range(3)], axis=1)
        assert np.abs(f - f_ref).max() < 2e-7 * np.abs(f_ref).max()
    dev.close()
    for r_A in (4.3, 5.2, 8.9, 9.5):
        s, r, d = charged_pair_setup(r_A)
        dev = MartiniHIP(s, test_api=True)
        e, vir = dev.eval_forces()
        f = dev.download()["f"]
        dev.close()
        k = int(s.ljtype[1] + s.nlj * s.ljtype[0])
        kq = s.keR * s.charge[0] * s.charge[1]
        e_lj = 4 * s.eps[k] * ((s.sigma[k] / r) ** 12 - (s.sigma[k] / r) ** 6) + s.shift[k] if r < s.rmax else 0.0
        e_pair = cf.lj_rf_pair_E(r, s.sigma[k], s.eps[k], s.shift[k], kq, s.krf, s.crf, s.rmax)
        e_self = -0.5 * s.keR * s.crf * float(np.sum(s.charge[s.species] ** 2))
        assert abs(e["lj"] - e_lj) < 1e-12 * max(abs(e_lj), 1e-6)
        assert abs(e["ele"] - ((e_pair - e_lj) + e_self)) < 1e-12 * abs(e_self)
        F = cf.lj_rf_pair_F(r, s.sigma[k], s.eps[k], kq, s.krf, s.rmax)
        assert np.abs(np.array([f[0][0], f[1][0], f[2][0]]) - F * d).max() < 1e-11 * max(abs(F), 1e-9)
        dd = r * d
        want = F * np.array([d[0] * dd[0], d[1] * dd[1], d[2] * dd[2], d[0] * dd[1], d[0] * dd[2], d[1] * dd[2]])
        assert np.abs(vir - want).max() < 1e-11 * max(np.abs(want).max(), 1e-9)
