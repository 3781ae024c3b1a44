#!/usr/bin/env python3
"""The 2.04 M-bead bilayer box at the cadence of the reference's shipped example decks (dt = 20 fs, updateRate = 20; examples/object/object.data:13,41)
beside the lipid deck's own (dt = 10 fs, updateRate = 10): ms/step, temperature, and -- at the end of a rebuild period -- the forces from the list as
it stands against the forces right after a fresh rebuild (no pair may have entered the cut-off unseen).   python3 tools/lipid_cadence.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench
from ddcmd_amd.martini import MartiniHIP
from ddcmd_amd import units_convert
for dt_fs, rate in ((10.0, 10), (20.0, 20)):
    s, name, _, _ = bench.build_setup("lipid", None, "12,12,6")
    s.dt = units_convert(dt_fs, "fs"); s.updateRate = rate
    m = MartiniHIP(s)
    m.eval_forces()
    for _ in range(10):
        m.group_temperatures(); m.step(rate * 2)
    m.sync()
    t0 = time.perf_counter()
    nst = 0
    for _ in range(10):
        m.group_temperatures(); m.step(rate * 2); nst += rate * 2
    m.sync()
    el = (time.perf_counter() - t0) / nst
    T = m.group_temperatures()
    # the last step of a period, forces from the aged list vs a fresh one
    m.step(rate - 1)
    m.eval_forces(); fa = m.download()["f"]
    m.build_list(); m.eval_forces(); fb = m.download()["f"]
    err = max(float(np.abs(fa[c] - fb[c]).max()) for c in range(3)) / max(float(np.abs(fb[c]).max()) for c in range(3))
    print("dt %4.0f fs, rebuild every %2d steps: %.4f ms/step  (%.1f ns/day)  T %s K  aged list vs fresh list: max |dF|/max|F| = %.1e" %
          (dt_fs, rate, el * 1e3, 86400.0 / el * dt_fs * 1e-6, np.round(np.asarray(T) / units_convert(1.0, "K"), 1), err))
    m.close()
