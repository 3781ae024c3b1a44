#!/usr/bin/env python3
"""The PCIe-inclusive rate of the boundary's host-buffer mode (INTEGRATOR.uses_gpu = 0: the reference's CPU integrator with the
potential on the accelerator): every step hands positions over from host arrays and takes forces back.
   python3 tools/pcie_rate.py [lattice] [steps]   ->  ms per step and atom-steps/s with the transfers inside the timed region"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ddcmd_amd
from ddcmd_amd.martini import MartiniHIP
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
s = ddcmd_amd.make_water_setup(n)
m = MartiniHIP(s)
m.eval_forces()
m.step(40)
d = m.download()
r = [np.ascontiguousarray(a) for a in d["r"]]
v = [np.ascontiguousarray(a) for a in d["v"]]
m.sync()
t0 = time.perf_counter()
for k in range(steps):
    m.upload_positions(r, v)            # sendGPUState
    m.eval_forces()                     # martiniHIP (energies + virial to the host)
    f = m.download(4)["f"]              # sendForceEnergyToHost
m.sync()
el = (time.perf_counter() - t0) / steps
print("%d beads: %.3f ms per step with positions+velocities up and forces down every step = %.3g atom-steps/s (device-resident: see bench.py)" % (s.natoms, el * 1e3, s.natoms / el))
