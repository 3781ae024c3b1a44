#!/usr/bin/env python3
"""A long NVE run (round 6): the lean step, the shell-limited walk and the displacement bound over thousands of rebuild periods -- a pair that the walk
skipped although it had come inside the cut-off would show as a jump of the total energy; the integrator's own drift (20 fs steps) is smooth.
   python3 tools/long_nve_r06.py [steps] [n] [water | lipid] [grid, e.g. 2,2,2: an in-process group]      # prints E_total per block, the drift per ns and the largest block-to-block jump"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ddcmd_amd.synth import make_water_setup
from ddcmd_amd.martini import MartiniHIP, MartiniGroup
from ddcmd_amd.deck import load_deck, units_convert
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
n = int(sys.argv[2]) if len(sys.argv) > 2 else 12
workload = sys.argv[3] if len(sys.argv) > 3 else "water"
if workload == "lipid":
    deck = os.path.join(ROOT, "tests", "golden", "lipid_deck")
    s = load_deck(os.path.join(deck, "object_nvt.data"), restart_file=os.path.join(deck, "relaxed", "restart"))
    s.group_type = np.zeros(s.ngroup, np.int32)      # FREE: NVE
else:
    s = make_water_setup(n, temperature_K=300.0)
block = 1000
grid = tuple(int(x) for x in sys.argv[4].split(",")) if len(sys.argv) > 4 else None
m = MartiniGroup(s, grid) if grid else MartiniHIP(s)
m.eval_forces()
m.step(2000)      # (settle: the lattice start melts)
E = []
t0 = time.time()
for b in range(steps // block):
    m.step(block)
    e, vir, rk, _ = m.energies()
    E.append(e["total"] + rk)
    if b % 10 == 9:
        print("step %7d  E_total %.10g  E_pot %.10g  E_kin %.8g  T %.2f K" % ((b + 1) * block, E[-1], e["total"], rk, 2.0 * rk / (3.0 * s.natoms) / units_convert(1.0, "K")), flush=True)
E = np.array(E)
cE = units_convert(1.0, None, "kJ/mol")
ns = steps * s.dt * 1e-6
slope = np.polyfit(np.arange(E.size), E, 1)[0] * E.size
jumps = np.abs(np.diff(E))
print(("%d steps of %g fs (%.2f ns), %d beads" + (" on %s bricks" % "x".join(map(str, grid)) if grid else "") + ", %d rebuilds, %.1f s: E_total %.8g -> %.8g; drift %.3g kJ/mol per bead per ns; largest block-to-block change %.3g x the mean one")
      % (steps, s.dt, ns, s.natoms, (m.ranks[0] if grid else m).list_stats()["rebuilds"], time.time() - t0, E[0], E[-1], cE * slope / s.natoms / ns, jumps.max() / jumps.mean()))
