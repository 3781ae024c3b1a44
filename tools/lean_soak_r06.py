#!/usr/bin/env python3
"""The lean step at the size SOAK_N names (64: 1.05 M beads; 102: the 4.24 M headline box, lean since round 6): water NVE, 4000 steps in calls of 20 (200 rebuilds), once lean (one launch per step: images staged from their
owners, displacement bound kept by the pair kernel, second-stage sums in batches) and once with DDCMI_NO_LEAN_STEP=1 -- potential and kinetic energy after every
1000 steps must agree to the last bit, the energy drift says whether a pair was ever missed.   python3 tools/lean_soak_r05.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ddcmd_amd
from ddcmd_amd.martini import MartiniHIP
K = ddcmd_amd.units_convert(1.0, None, "K")


def run(lean):
    if lean: os.environ.pop("DDCMI_NO_LEAN_STEP", None)
    else: os.environ["DDCMI_NO_LEAN_STEP"] = "1"
    s = ddcmd_amd.make_water_setup(int(os.environ.get("SOAK_N", "64")))
    m = MartiniHIP(s)
    m.eval_forces(); m.step(200)
    e, _, rk, _ = m.energies(); e0 = e["total"] + rk
    out = []
    for blk in range(4):
        for _ in range(50): m.step(20)
        e, _, rk, _ = m.energies()
        out.append((e["total"], rk))
        print("%s step %5d: E %.12g drift/E0 %+.2e T %.1f K rebuilds %d" % ("lean  " if lean else "legacy", 200 + 1000 * (blk + 1), e["total"] + rk, (e["total"] + rk - e0) / abs(e0),
              K * 2.0 * rk / (3.0 * s.natoms), m.list_stats()["rebuilds"]), flush=True)
    m.close()
    return out


a, b = run(True), run(False)
print("lean against a reduction launch per step, E_pot and E_kin after 1000..4000 steps:", ["same bits" if x == y else "DIFFERENT %r %r" % (x, y) for x, y in zip(a, b)])
