#!/usr/bin/env python3
"""Relax tests/golden/lipid_deck (a hand-placed bilayer patch that heats to ~1900 K when
run NVE) into a 310 K starting point for the lipid benchmark workload:

  * tests/golden/lipid_deck/object_nvt.data  = object.data with the GROUP switched to
    BERENDSEN (Teq 310 K, tau 1 ps): BASELINE config 5's thermostat;
  * tests/golden/lipid_deck/relaxed/restart + snapshot.mem/atoms#000000 = the state after
    STEPS steps of the CPU oracle under a tight Berendsen coupling (tau 100 fs).

Everything is produced by our own oracle (oracle/ddc_oracle.c); nothing is read from
the reference.  Run from the repo root:  python tests/golden/make_lipid_relaxed.py
"""
import os
import sys
import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import pyoracle  # noqa: E402
from ddcmd_amd.deck import load_deck, units_convert, GROUP_BERENDSEN  # noqa: E402

DECK = os.path.join(ROOT, "tests", "golden", "lipid_deck")
STEPS = 4000


def main():
    s = load_deck(os.path.join(DECK, "object.data"))
    s.group_type = np.array([GROUP_BERENDSEN], np.int32)
    s.group_Teq = np.array([units_convert(310.0, "K")])
    s.group_tau = np.array([units_convert(100.0, "fs")])
    s.group_interval = np.array([1], np.int32)
    o = pyoracle.Oracle(s)
    o.forces()
    o.group_temperature()
    for b in range(STEPS // 10):
        # the group temperature Berendsen scales with is the one eval_energyInfo last
        # published (printinfo cadence in simulateMaster): refresh it every 10 steps
        e, v, rk, _ = o.step(10)
        o.group_temperature()
        if (b + 1) % 50 == 0:
            T = 2.0 * rk / (3.0 * s.natoms) / units_convert(1.0, "K")
            print("step %5d  epot %.6f  T %.1f K" % ((b + 1) * 10, e["total"], T))
    A = units_convert(1.0, None, "Angstrom")
    out = os.path.join(DECK, "relaxed")
    os.makedirs(os.path.join(out, "snapshot.mem"), exist_ok=True)
    L = [s.h[0] * A, s.h[4] * A, s.h[8] * A]
    h = "h=     %.3f 0.0 0.0\n       0.0 %.3f 0.0\n       0.0 0.0 %.3f ;" % tuple(L)
    n = s.natoms
    open(os.path.join(out, "restart"), "w").write(
        "simulate SIMULATE { loop=0; time=0.000000 ;}\nbox BOX {\n%s\n}\ncollection COLLECTION { mode=VARRECORDASCII; size=%d; files=relaxed/snapshot.mem/atoms#;}\n" % (h, n))   # paths are relative to the run directory (where object_nvt.data lives)
    # wrap back into the box like backInBox (the oracle keeps unwrapped coordinates between rebuilds)
    r = np.stack([o.rx, o.ry, o.rz], 1) * A
    r -= np.array(L) * np.rint(r / np.array(L))
    v = np.stack([o.vx, o.vy, o.vz], 1) * A
    with open(os.path.join(out, "snapshot.mem", "atoms#000000"), "w") as f:
        f.write("particle FILEHEADER {type=MULTILINE; datatype=VARRECORDASCII; checksum=NONE;\nloop=0; time=0.000000;\n"
                "nfiles=1; nrecord=%d; nfields=10;\nfield_names=id class type group rx ry rz vx vy vz;\n"
                "field_types=u s s s f f f f f f;\n%s\ngroups = group ;\ntypes = ATOM ;\n} \n\n" % (n, h))
        for i in range(n):
            f.write("%14d ATOM %10s group %21.13e %21.13e %21.13e %21.13e %21.13e %21.13e\n" %
                    (int(s.gid[i]), s.species_name[int(s.species[i])], r[i, 0], r[i, 1], r[i, 2], v[i, 0], v[i, 1], v[i, 2]))
    obj = open(os.path.join(DECK, "object.data")).read()
    assert obj.count("group GROUP { type = FREE; }") == 1
    obj = obj.replace("group GROUP { type = FREE; }", "group GROUP { type = BERENDSEN; Teq = 310 K; tau = 1000 fs; }")
    open(os.path.join(DECK, "object_nvt.data"), "w").write(obj)
    print("wrote", out)


if __name__ == "__main__":
    main()
