#!/usr/bin/env python3
"""Tuning aid: k_nonbond (the plain launch, fresh list) on ONE equilibrated state shared by every build under test -- ablated or mocked
kernels that compute nonsense still run, and every build sees the same beads:
   python3 tools/time_nonbond_state.py <lattice> [launches]                 (tree library: makes /tmp/ddcmi_state_<n>_<density>.npz if absent)
   DDCMI_LIB=tuning/libddcmi_x.so python3 tools/time_nonbond_state.py ...   (reads it)
DDCMI_BENCH_DENSITY_SCALE scales the water density (LDS room for experiments)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ddcmd_amd
from ddcmd_amd.martini import MartiniHIP
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dens = float(os.environ.get("DDCMI_BENCH_DENSITY_SCALE", "1.0"))
fn = "/tmp/ddcmi_state_%d_%g.npz" % (n, dens)
s = ddcmd_amd.make_water_setup(n, density_scale=dens)
if not os.path.exists(fn):
    if os.environ.get("DDCMI_LIB"):
        sys.exit("run the tree library first: it makes " + fn)
    m = MartiniHIP(s)
    m.eval_forces(); m.step(220)
    d = m.download()
    np.savez(fn, r=np.stack(d["r"]), v=np.stack(d["v"]))
    m.close()
z = np.load(fn)
s.rx, s.ry, s.rz = z["r"]
s.vx, s.vy, s.vz = z["v"]
m = MartiniHIP(s)
for _ in range(3):
    m.eval_forces()
m.timing(True)
for _ in range(reps):
    m.eval_forces()
launches, ms = m.timing_read()
st = m.list_stats()
print("%-28s lattice %d density x%g: k_nonbond (plain, fresh list) %.4f ms  entries/bead %.1f" % (os.path.basename(os.environ.get("DDCMI_LIB", "tree")), n, dens, ms / launches, st["entries"] / s.natoms))
