#!/usr/bin/env python3
"""Debugging aid: long NVT runs of the lipid deck at several tilings; prints the group temperature."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ddcmd_amd
from ddcmd_amd.martini import MartiniHIP
from ddcmd_amd.deck import load_deck
from ddcmd_amd.synth import replicate_setup

deck = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "lipid_deck")
base = load_deck(os.path.join(deck, "object_nvt.data"), restart_file=os.path.join(deck, "relaxed", "restart"))
K = ddcmd_amd.units_convert(1.0, None, "K")
reps_list = [tuple(int(x) for x in a.split(",")) for a in sys.argv[2:]] or [(1, 1, 1), (2, 2, 1), (4, 4, 2)]
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
for reps in reps_list:
    s = replicate_setup(base, reps) if reps != (1, 1, 1) else base
    m = MartiniHIP(s); m.eval_forces(); m.group_temperatures()
    try:
        for blk in range(steps // 200):
            for _ in range(10):
                m.step(20); T = m.group_temperatures()
            e, _, rk, _ = m.energies()
            st = m.download()
            v2 = st["v"][0] ** 2 + st["v"][1] ** 2 + st["v"][2] ** 2
            ihot = int(np.argmax(v2))
            vmax = float(np.sqrt(v2[ihot]))
            hot = "%s gid %x" % (s.species_name[int(s.species[ihot])], int(s.gid[ihot]))
            print("reps %s step %5d Epot/N %.6f T %.1f  max|v| %.3e  bond %.4f angle %.4f tors %.5f impr %.5f  fastest: %s" % (reps, (blk + 1) * 200, e["total"] / s.natoms, K * float(T[0]), vmax, e["bond"] / s.natoms, e["angle"] / s.natoms, e["tors"] / s.natoms, e["impr"] / s.natoms, hot), flush=True)
    except Exception as ex:
        print("reps", reps, "FAILED:", ex)
    m.close()
