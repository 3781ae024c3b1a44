#!/usr/bin/env python3
"""VERDICT r3 item 4, the cheap decisive half: would a THIRD k_nonbond workgroup per CU pay?  Three per CU need <= 53 KB of LDS each and
<= 84 VGPRs.  The 4 A-skin neighbourhoods (3140 staged beads x 24 B) cannot get there without a new data layout -- but a 1 A skin
shrinks the cells (8 -> 6.5 A) and the staged set to ~1700 beads = 41 KB with the SAME layout, so the occupancy question can be asked of
the unchanged kernel: library A = run-time LDS layout, 2 workgroups per CU, 128-VGPR budget (4 waves / SIMD); library C = the same with
__launch_bounds__(512, 6) (84-VGPR budget) and 3 workgroups per CU.   [DDCMI_LIB=...] python3 tools/occupancy_experiment.py [lattice]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["DDCMI_NO_FUSED_STEP"] = "1"
import ddcmd_amd
from ddcmd_amd.martini import MartiniHIP

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
for skin in (4.0, 1.0):
    s = ddcmd_amd.make_water_setup(n, skin_A=skin, update_rate=5)
    m = MartiniHIP(s)
    m.eval_forces()
    m.step(10)
    m.timing(True)
    m.step(10)
    m.sync()
    launches, ms = m.timing_read()
    st = m.list_stats()
    print("%-22s skin %.1f A: list entries/bead %6.1f  k_nonbond %.4f ms" % (os.path.basename(os.environ.get("DDCMI_LIB", "tree")), skin, st["entries"] / s.natoms, ms / max(launches, 1)))
    m.close()
