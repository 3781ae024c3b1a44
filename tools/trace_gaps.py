"""Per-CU timelines of k_nonbond from a -DDDCMI_TRACE_BLOCKS build: how much of a CU's two workgroup slots is
idle between consecutive workgroups (dispatch gaps) and at the end of the launch (tail)?
   DDCMI_LIB=ddcmd_amd/lib/variants/libddcmi_trace.so python3 tools/trace_gaps.py <lattice>"""
import ctypes, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from ddcmd_amd.synth import make_water_setup
from ddcmd_amd.martini import MartiniHIP
from ddcmd_amd._lib import load_library

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
s = make_water_setup(n)
m = MartiniHIP(s)
m.eval_forces()
m.step(45)
m.sync()
lib = load_library()
nb = 65536
buf = np.zeros((nb, 8), dtype=np.uint64)
lib.ddcmi_debug_trace.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert lib.ddcmi_debug_trace(buf.ctypes.data, nb) == 0
b = buf[buf[:, 0] > 0].astype(np.int64)
# keep the last launch only: entries of earlier launches (other grid sizes) are older by at least one step
order = np.argsort(b[:, 0])
b = b[order]
st = b[:, 0] / 100.0
cut = np.flatnonzero(np.diff(st) > 40.0)
if cut.size:
    b = b[cut[-1] + 1:]
t0 = b[:, 0].min()
start, staged, end = (b[:, 0] - t0) / 100.0, (b[:, 1] - t0) / 100.0, (b[:, 2] - t0) / 100.0
xcc, hw, nown = b[:, 4] & 0xf, b[:, 5], b[:, 7]
cu = (hw >> 8) & 0xf; sh = (hw >> 12) & 1; se = (hw >> 13) & 7
key = ((xcc * 8 + se) * 2 + sh) * 16 + cu
T = end.max()
print("lattice %d: %d workgroups in the last launch, %d with beads, launch %.1f us; distinct CUs seen %d" % (n, len(b), int((nown > 0).sum()), T, len(np.unique(key))))
busy = (end - start).sum()
ncu = len(np.unique(key))
print("block-time %.0f us = %.1f us per CU; per CU with 2 slots over %.1f us => slot occupancy %.3f" % (busy, busy / ncu, T, busy / (2 * ncu * T)))
# per CU: union coverage with multiplicity
gaps, tails, firsts, conc = [], [], [], []
for k in np.unique(key):
    sel = key == k
    ev = sorted([(x, 1) for x in start[sel]] + [(x, -1) for x in end[sel]])
    lvl, last, area = 0, 0.0, [0.0, 0.0, 0.0, 0.0]
    for t, d in ev:
        area[min(lvl, 3)] += t - last
        last = t
        lvl += d
    area[0] += T - last
    conc.append(area)
    tails.append(T - end[sel].max())
    firsts.append(start[sel].min())
conc = np.array(conc)
print("per-CU time at 0 / 1 / 2 / 3+ resident workgroups (mean us): %.1f / %.1f / %.1f / %.1f" % tuple(conc.mean(axis=0)))
print("tail idle per CU (mean/max us): %.1f / %.1f ; first start (mean/max): %.1f / %.1f" % (np.mean(tails), np.max(tails), np.mean(firsts), np.max(firsts)))
full = nown > 400
print("full tiles: staging %.1f us, compute %.1f us; blocks per CU mean %.1f" % ((staged - start)[full].mean(), (end - staged)[full].mean(), len(b) / ncu))
# does a block run slower when the CU holds two?  compare blocks in the first round
print("compute us by start decile:", np.round([np.mean((end - staged)[full & (start >= lo) & (start < hi)]) if (full & (start >= lo) & (start < hi)).any() else 0 for lo, hi in zip(np.linspace(0, T, 11)[:-1], np.linspace(0, T, 11)[1:])], 1))
