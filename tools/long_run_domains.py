#!/usr/bin/env python3
"""Long NVT run of the tiled lipid deck on px*py*pz emulated domains (one GPU): beads and whole lipids cross
domain faces many times; the temperature must stay at the target and no bead may be lost.
python tools/long_run_domains.py [steps] [reps] [grid]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ddcmd_amd
from ddcmd_amd.martini import MartiniGroup
from ddcmd_amd.deck import load_deck
from ddcmd_amd.synth import replicate_setup

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
reps = tuple(int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "4,4,2").split(","))
grid = tuple(int(x) for x in (sys.argv[3] if len(sys.argv) > 3 else "2,2,2").split(","))
deck = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "lipid_deck")
s = replicate_setup(load_deck(os.path.join(deck, "object_nvt.data"), restart_file=os.path.join(deck, "relaxed", "restart")), reps)
K = ddcmd_amd.units_convert(1.0, None, "K")
g = MartiniGroup(s, grid)
g.eval_forces(); g.group_temperatures()
n0 = [int(r.lib.ddcmi_nlocal(r.ctx)) for r in g.ranks]
for blk in range(steps // 500):
    for _ in range(25):
        g.step(20); T = g.group_temperatures()
    e, _, rk, _ = g.energies()
    nl = [int(r.lib.ddcmi_nlocal(r.ctx)) for r in g.ranks]
    assert sum(nl) == s.natoms, "beads lost: %s" % nl
    print("%s beads on %s domains, step %5d: Epot/N %.6f T %.2f K  beads per domain %s (start %s)" % (s.natoms, grid, (blk + 1) * 500, e["total"] / s.natoms, K * float(T[0]), nl, n0), flush=True)
