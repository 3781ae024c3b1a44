#!/usr/bin/env python3
"""Tuning aid: k_nonbond alone on a fixed state (no integration, so ablated kernels that compute nonsense still run):
   [DDCMI_LIB=...] python3 tools/time_nonbond.py <lattice> [launches]   ->  ms per launch (HIP events on the library's stream)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ddcmd_amd
from ddcmd_amd.martini import MartiniHIP
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
s = ddcmd_amd.make_water_setup(n)
m = MartiniHIP(s)
if not os.environ.get("DDCMI_LIB"):
    m.eval_forces(); m.step(40)       # an equilibrated-ish state when the kernel is the real one
for _ in range(3):
    m.eval_forces()
m.timing(True)
for _ in range(reps):
    m.eval_forces()
launches, ms = m.timing_read()
print("%-28s lattice %d: k_nonbond %.4f ms" % (os.path.basename(os.environ.get("DDCMI_LIB", "tree")), n, ms / launches))
