#!/usr/bin/env python3
"""Tuning aid: cost of the tail of the list in k_nonbond.  Times the pair kernel on the 4M water box
with the contract's 4 A skin and with shorter skins (same state, cutoff unchanged): the difference is
what the entries between the radii cost -- entries no bead of a wave accepts."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ddcmd_amd
from ddcmd_amd.martini import MartiniHIP

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
for skin in (4.0, 3.0, 2.0, 1.0):
    s = ddcmd_amd.make_water_setup(n, skin_A=skin, update_rate=20)
    m = MartiniHIP(s)
    m.eval_forces()
    m.step(3)
    m.timing(True)
    m.step(10)
    m.sync()
    launches, ms = m.timing_read()
    st = m.list_stats()
    print("skin %.1f A: list entries/bead %.1f  k_nonbond %.3f ms" % (skin, st["entries"] / s.natoms, ms / max(launches, 1)))
    m.close()
