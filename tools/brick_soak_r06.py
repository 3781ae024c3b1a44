#!/usr/bin/env python3
"""Soak of a decomposed rank through the RCCL loopback: the lean step with the halo staged from the receive buffer (the default) against the step
with a halo update launch and a reduction launch each (DDCMI_NO_DIRECT_HALO=1 DDCMI_NO_LEAN_STEP=1), thousands of steps across hundreds of rebuilds
with migration through the wire: potential and kinetic energy must agree to the last bit at every checkpoint (water: NVE, the drift says whether a
pair was ever missed; the lipid brick: Berendsen, bonded partners out of the receive buffer).   python3 tools/brick_soak_r06.py water|lipid [steps]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench, ddcmd_amd
from ddcmd_amd.martini import MartiniRank
kind = sys.argv[1] if len(sys.argv) > 1 else "water"
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 4000
K = ddcmd_amd.units_convert(1.0, None, "K")


def run(legacy):
    for k in ("DDCMI_NO_DIRECT_HALO", "DDCMI_NO_LEAN_STEP"):
        if legacy: os.environ[k] = "1"
        else: os.environ.pop(k, None)
    os.environ["DDCMI_RCCL_LOOPBACK"] = "1"
    s = bench.build_setup("water", 51, "6,6,3")[0] if kind == "water" else bench.build_setup("lipid", None, "6,6,3")[0]
    m = MartiniRank(s, np.arange(s.natoms))
    buf = ctypes.create_string_buffer(128)
    assert m.lib.ddcmi_comm_unique_id(buf) == 0
    m.comm_init(0, 1, buf.raw, (1, 1, 1))
    m.preflight()
    m.upload_local()
    m.eval_forces()
    thermo = any(int(t) == 1 for t in np.asarray(s.group_type).ravel())
    if thermo: m.group_temperatures()
    if kind == "water": m.step(200)
    e, _, rk, _ = m.energies(); e0 = e["total"] + rk
    out = []
    for blk in range(4):
        for _ in range(nsteps // 80):
            m.step(20)
            if thermo: m.group_temperatures()
        e, _, rk, _ = m.energies()
        out.append((e["total"], rk))
        print("%s %s step %6d: E %.12g drift/E0 %+.2e T %.1f K rebuilds %d owned %d" % (kind, "legacy" if legacy else "lean  ", (blk + 1) * (nsteps // 4), e["total"] + rk,
              (e["total"] + rk - e0) / abs(e0), K * 2.0 * rk / (3.0 * s.natoms), m.list_stats()["rebuilds"], int(m.lib.ddcmi_nlocal(m.ctx))), flush=True)
    m.close()
    return out


a, b = run(False), run(True)
print("lean + direct halo against update + reduction launches:", ["same bits" if x == y else "DIFFERENT %r %r" % (x, y) for x, y in zip(a, b)])
