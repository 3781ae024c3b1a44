#!/usr/bin/env python3
"""Soak of the 2.04 M-bead bilayer at the cadence of the reference's shipped decks -- dt = 20 fs, updateRate = 20
(examples/waterbox/object.data:10,35; examples/object/object.data:13,41) -- for 20 000 steps (0.4 ns): temperature, energies, the longest
bond and the hottest bead every 2000 steps; at the end the forces from the aged list against a fresh one.   python3 tools/lipid_soak_r05.py [steps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench
from ddcmd_amd.martini import MartiniHIP
from ddcmd_amd import units_convert
nsteps = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
s, name, _, _ = bench.build_setup("lipid", None, "12,12,6")
s.dt = units_convert(20.0, "fs"); s.updateRate = 20
K, kj, A = units_convert(1.0, "K"), units_convert(1.0, "kJ*mol^-1"), units_convert(1.0, "Angstrom")
m = MartiniHIP(s)
e, _ = m.eval_forces()
bij = m.terms["bond_ij"].reshape(-1, 2)
box = s.box
print("%s: dt 20 fs, rebuild every 20 steps, Berendsen 310 K (temperature published every 20 steps)" % name)
t0 = time.perf_counter()
done = 0
while done < nsteps:
    for _ in range(100):
        m.group_temperatures(); m.step(20)
    done += 2000
    m.sync()
    e, vir, rk, _ = m.energies()
    d = m.download()
    r, v = np.stack(d["r"], 1), np.stack(d["v"], 1)
    db = r[bij[:, 0]] - r[bij[:, 1]]
    db -= box * np.rint(db / box)
    bl = np.sqrt((db * db).sum(1)) / A
    vmax = np.sqrt((v * v).sum(1)).max() * units_convert(1.0, None, "Angstrom/fs") if False else np.sqrt((v * v).sum(1)).max() / A
    T = m.group_temperatures()[0] / K
    print("step %6d  T %.2f K  Epot %.6e  Ekin %.6e kJ/mol  longest bond %.3f A  fastest bead %.4f A/fs  finite %s  (%.1f s)" %
          (done, T, e["total"] / kj, rk / kj, bl.max(), vmax, bool(np.isfinite(r).all()), time.perf_counter() - t0))
m.step(19)
m.eval_forces(); fa = m.download()["f"]
m.build_list(); m.eval_forces(); fb = m.download()["f"]
err = max(float(np.abs(fa[c] - fb[c]).max()) for c in range(3)) / max(float(np.abs(fb[c]).max()) for c in range(3))
print("after %d steps: forces from the 19-step-old list vs a fresh list: max |dF|/max|F| = %.1e; rebuilds %d" % (done + 19, err, m.list_stats()["rebuilds"]))
m.close()
