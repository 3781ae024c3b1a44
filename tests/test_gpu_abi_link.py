"""GPU test (-m gpu): the C-ABI driven from a C program that itself defines ddcMD's own function names (nglf, ddcenergy,
object_get, units_convert ... with the reference's signatures) and links with -lddcmi alone -- what wiring libddcmi.so into
ddcMD looks like to the linker.  Forces, energies, virial and three NGLF steps of a 4000-bead water box
against the oracle."""
import os
import subprocess
import numpy as np
import pytest

import pyoracle
import ddcmd_amd
from ddcmd_amd import _lib

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_c_abi_beside_ddcmds_own_names_matches_the_oracle(tmp_path):
    s = ddcmd_amd.make_water_setup(10)
    f64 = lambda a: np.ascontiguousarray(a, dtype=np.float64)
    i32 = lambda a: np.ascontiguousarray(a, dtype=np.int32)
    path = str(tmp_path / "setup.bin")
    with open(path, "wb") as f:
        f.write(i32([s.natoms, s.nspecies, s.nlj, s.pbc]).tobytes())
        f.write(f64(list(np.asarray(s.h).ravel()) + [s.rmax, s.keR, s.krf, s.crf, s.deltaR, s.dt, 0.0]).tobytes())
        f.write(f64(s.mass).tobytes()); f.write(f64(s.charge).tobytes()); f.write(i32(s.ljtype).tobytes()); f.write(i32(s.moltype).tobytes())
        for a in (s.sigma, s.eps, s.shift, s.rx, s.ry, s.rz, s.vx, s.vy, s.vz):
            f.write(f64(a).tobytes())
        f.write(np.ascontiguousarray(s.gid, dtype=np.uint64).tobytes()); f.write(i32(s.species).tobytes()); f.write(i32(s.group).tobytes())
    exe = str(tmp_path / "link_with_ddcmd_names")
    libdir = os.path.dirname(_lib.LIB_PATH)
    subprocess.check_call(["gcc", "-std=gnu99", "-Wall", "-Wextra", "-Werror", "-I" + os.path.join(ROOT, "include"), "-o", exe,
                           os.path.join(ROOT, "tests", "abi", "link_with_ddcmd_names.c"), "-L" + libdir, "-lddcmi",
                           "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"])
    out = subprocess.run([exe, path], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, (out.stdout, out.stderr[-2000:])
    rec = {l.split()[0]: [float(x) for x in l.split()[1:]] for l in out.stdout.splitlines() if l[:1] in "EVF"}
    assert "stubs_called_by_library 0" in out.stdout
    o = pyoracle.Oracle(s)
    e, vir = o.forces()
    assert abs(rec["E0"][0] - e["lj"]) < 1e-10 * abs(e["lj"]) and abs(rec["E0"][1] - e["total"]) < 1e-10 * abs(e["total"])
    assert np.abs(np.array(rec["V0"]) - np.asarray(vir)).max() < 1e-9 * np.abs(vir).max()
    n = s.natoms
    fmax = max(o.fx.max(), o.fy.max(), o.fz.max())
    assert np.abs(np.array(rec["F0"][:4]) - np.array([o.fx[0], o.fy[n // 2], o.fz[n - 1], fmax])).max() < 1e-9 * fmax
    assert abs(rec["F0"][4]) < 1e-8 * fmax * n ** 0.5
    e3, _, rk3, _ = o.step(3)
    assert abs(rec["E3"][0] - e3["total"]) < 1e-8 * abs(e3["total"]) and abs(rec["E3"][1] - rk3) < 1e-8 * rk3
