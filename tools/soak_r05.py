#!/usr/bin/env python3
"""Round 5 soak at the bench's sizes, on the paths the round added: 4.24 M-bead water NVE for 6000 steps (300 rebuilds: interior-first search, staging
capacity following the measured neighbourhood) -- energy drift, rebuild count, the slowest 20-step window; the 530 k brick through the RCCL loopback for
6000 steps (halo staged from the receive buffer) against the single-domain run of the same box (energies after every 1000 steps); the bilayer under 40 LJ
types for 3000 steps (tags in the staged z, two-level table): temperature.   python3 tools/soak_r05.py"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ddcmd_amd
import bench
from ddcmd_amd.martini import MartiniHIP, MartiniRank
K = ddcmd_amd.units_convert(1.0, None, "K")


def run_water(n, loopback, steps, label):
    s = ddcmd_amd.make_water_setup(n)
    if loopback:
        os.environ["DDCMI_RCCL_LOOPBACK"] = "1"
        m = MartiniRank(s, np.arange(s.natoms))
        buf = ctypes.create_string_buffer(128)
        assert m.lib.ddcmi_comm_unique_id(buf) == 0
        m.comm_init(0, 1, buf.raw, (1, 1, 1)); m.upload_local()
    else:
        os.environ.pop("DDCMI_RCCL_LOOPBACK", None)
        m = MartiniHIP(s)
    m.eval_forces(); m.step(200)
    e, _, rk, _ = m.energies(); e0 = e["total"] + rk
    out = []
    worst = 0.0
    for blk in range(steps // 1000):
        for _ in range(50):
            m.sync(); t0 = time.perf_counter(); m.step(20); m.sync(); worst = max(worst, time.perf_counter() - t0)
        e, _, rk, _ = m.energies()
        out.append((e["total"], rk))
        print("%s step %5d: E %.10g drift/E0 %+.2e T %.1f K rebuilds %d slowest 20-step window %.2f ms" % (label, 200 + 1000 * (blk + 1), e["total"] + rk, (e["total"] + rk - e0) / abs(e0),
              K * 2.0 * rk / (3.0 * s.natoms), m.list_stats()["rebuilds"], worst * 1e3), flush=True)
    m.close()
    return out


run_water(102, False, 6000, "water 4.24M")
a = run_water(51, True, 6000, "brick 530k loopback")
b = run_water(51, False, 6000, "brick 530k one domain")
print("loopback vs one domain, relative difference of E_pot / E_kin after 1000..6000 steps:", ["%.1e / %.1e" % (abs(x[0] - y[0]) / abs(y[0]), abs(x[1] - y[1]) / y[1]) for x, y in zip(a, b)])
s, name, _, _ = bench.build_setup("lipid", None, "12,12,6", 40)
m = MartiniHIP(s); m.eval_forces(); m.group_temperatures()
for blk in range(3):
    for _ in range(50):
        m.step(20); T = m.group_temperatures()
    e, _, rk, _ = m.energies()
    print("%s step %5d: Epot/N %.6f T %.2f K rebuilds %d" % (name, 1000 * (blk + 1), e["total"] / s.natoms, K * float(T[0]), m.list_stats()["rebuilds"]), flush=True)
m.close()
