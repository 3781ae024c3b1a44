#!/usr/bin/env python3
"""Soak at the headline sizes: 4.0 M-bead water, NVE, 2000 steps (100 rebuilds): energy drift; 2.04 M-bead bilayer,
Berendsen, 3000 steps (300 rebuilds): temperature.  python tools/soak.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ddcmd_amd
from ddcmd_amd.martini import MartiniHIP
from ddcmd_amd.deck import load_deck
from ddcmd_amd.synth import replicate_setup

K = ddcmd_amd.units_convert(1.0, None, "K")
s = ddcmd_amd.make_water_setup(100)
m = MartiniHIP(s); m.eval_forces(); m.step(200)
e, _, rk, _ = m.energies(); e0 = e["total"] + rk
t0 = time.time()
for blk in range(4):
    m.step(500)
    e, _, rk, _ = m.energies()
    print("water 4.0M step %5d: E %.9g drift/E0 %+.2e T %.1f K" % (200 + 500 * (blk + 1), e["total"] + rk, (e["total"] + rk - e0) / abs(e0), K * 2.0 * rk / (3.0 * s.natoms)), flush=True)
print("  %.3f ms/step wall" % ((time.time() - t0) / 2000 * 1e3))
m.close()
deck = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "lipid_deck")
s = replicate_setup(load_deck(os.path.join(deck, "object_nvt.data"), restart_file=os.path.join(deck, "relaxed", "restart")), (12, 12, 6))
m = MartiniHIP(s); m.eval_forces(); m.group_temperatures()
t0 = time.time()
for blk in range(6):
    for _ in range(25):
        m.step(20); T = m.group_temperatures()
    e, _, rk, _ = m.energies()
    print("lipid 2.04M step %5d: Epot/N %.6f T %.2f K" % (500 * (blk + 1), e["total"] / s.natoms, K * float(T[0])), flush=True)
print("  %.3f ms/step wall" % ((time.time() - t0) / 3000 * 1e3))
m.close()
