"""GPU tests (-m gpu) of the spatial decomposition: px*py*pz domains emulated in
one process on one GPU (device copies instead of RCCL) must reproduce the
single-domain run and the oracle: migration, halo tables, per-step halo refresh."""
import numpy as np
import pytest

import pyoracle
import ddcmd_amd
from ddcmd_amd.synth import make_water_setup
from conftest import rel_force_err

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
