#!/usr/bin/env python3
"""Long-run sanity: many rebuild periods at full size.  Water (NVE): total energy drift; lipid bilayer
(Berendsen 310 K): temperature stays at the target.  python tools/long_run_check.py [steps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ddcmd_amd
from ddcmd_amd.martini import MartiniHIP
from ddcmd_amd.deck import load_deck
from ddcmd_amd.synth import replicate_setup

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
s = ddcmd_amd.make_water_setup(64)                      # 1.05 M beads, NVE
m = MartiniHIP(s); m.eval_forces()
m.step(200)
e, _, rk, _ = m.energies(); e0 = e["total"] + rk
for blk in range(4):
    m.step(steps // 4)
    e, _, rk, _ = m.energies()
    print("water  %8d beads, step %6d: E = %.9g  drift/E0 = %+.2e  T = %.1f K" % (s.natoms, 200 + (blk + 1) * (steps // 4), e["total"] + rk, (e["total"] + rk - e0) / abs(e0),
          ddcmd_amd.units_convert(2.0 * rk / (3.0 * s.natoms), None, "K")))
m.close()
deck = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "lipid_deck")
s = replicate_setup(load_deck(os.path.join(deck, "object_nvt.data"), restart_file=os.path.join(deck, "relaxed", "restart")), (8, 8, 4))
m = MartiniHIP(s); m.eval_forces(); m.group_temperatures()
for blk in range(8):
    for _ in range(steps // 8 // 20):
        m.step(20)
        T = m.group_temperatures()      # what eval_energyInfo publishes at ddcMD's print cadence; a stale value makes Berendsen over-correct
    e, _, rk, _ = m.energies()
    print("lipid  %8d beads, step %6d: Epot = %.9g  T = %.1f K" % (s.natoms, (blk + 1) * (steps // 8), e["total"], ddcmd_amd.units_convert(float(T[0]), None, "K")))
m.close()
