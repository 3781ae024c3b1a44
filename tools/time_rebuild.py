#!/usr/bin/env python3
"""Tuning aid: wall time of ddcmi_build_list on a fixed state: [DDCMI_LIB=...] python3 tools/time_rebuild.py <lattice> [reps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ddcmd_amd
from ddcmd_amd.martini import MartiniHIP
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
wl = os.environ.get("WORKLOAD", "water")
if wl == "water":
    s = ddcmd_amd.make_water_setup(n)
else:
    from ddcmd_amd.deck import load_deck
    from ddcmd_amd.synth import replicate_setup
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    deck = os.path.join(root, "tests", "golden", "lipid_deck")
    s = replicate_setup(load_deck(os.path.join(deck, "object_nvt.data"), restart_file=os.path.join(deck, "relaxed", "restart")), (12, 12, 6))
m = MartiniHIP(s)
m.eval_forces()
for _ in range(2):
    m.build_list()
m.sync()
t0 = time.perf_counter()
for _ in range(reps):
    m.build_list()
m.sync()
el = (time.perf_counter() - t0) / reps
st = m.list_stats()
print("%-24s %s %d beads: rebuild %.3f ms wall, %.1f entries/bead" % (os.path.basename(os.environ.get("DDCMI_LIB", "tree")), wl, s.natoms, el * 1e3, st["entries"] / s.natoms))
