"""Timeline of k_nonbond workgroups from a -DDDCMI_TRACE_BLOCKS build
(tools/build_variants.sh trace "-DDDCMI_TRACE_BLOCKS"; DDCMI_LIB=.../libddcmi_trace.so)."""
import ctypes, json, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np

from ddcmd_amd.synth import make_water_setup
from ddcmd_amd.martini import MartiniHIP
from ddcmd_amd._lib import load_library

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
s = make_water_setup(n)
m = MartiniHIP(s)
m.eval_forces()
for _ in range(45):
    m.step()
m.sync()
lib = load_library()
nb = 16384
buf = np.zeros((nb, 8), dtype=np.uint64)
lib.ddcmi_debug_trace.argtypes = [ctypes.c_void_p, ctypes.c_int]
rc = lib.ddcmi_debug_trace(buf.ctypes.data, nb)
assert rc == 0, rc
used = buf[:, 0] > 0
b = buf[used].astype(np.int64)
t0 = b[:, 0].min()
start = (b[:, 0] - t0) / 100.0      # us (100 MHz clock)
staged = (b[:, 1] - t0) / 100.0
end = (b[:, 2] - t0) / 100.0
xcc = b[:, 4] & 0xf
nown = b[:, 7]
live = nown > 0
out = {"blocks": int(used.sum()), "kernel_us": float(end.max()),
       "staging_us_mean(full tiles)": float((staged - start)[nown > 400].mean()),
       "compute_us_mean(full tiles)": float((end - staged)[nown > 400].mean()),
       "compute_us_p10_p50_p90_max": [float(x) for x in np.percentile((end - staged)[nown > 400], [10, 50, 90, 100])],
       "per_xcd": []}
for x in range(8):
    sel = xcc == x
    out["per_xcd"].append({"xcc": x, "blocks": int(sel.sum()), "beads": int(nown[sel].sum()), "first_start_us": float(start[sel].min()),
                           "last_end_us": float(end[sel].max()), "busy_block_us": float((end - start)[sel].sum())})
# concurrency over time: how many workgroups are resident in 20 slices of the launch
edges = np.linspace(0, end.max(), 21)
out["resident_blocks_per_slice"] = [int(((start < hi) & (end > lo)).sum()) for lo, hi in zip(edges[:-1], edges[1:])]
mid = 0.5 * (edges[:-1] + edges[1:])
out["resident_at_mid"] = [int(((start <= t) & (end > t)).sum()) for t in mid]
if os.environ.get("TRACE_BINS"):
    bins = [1, 64, 128, 192, 256, 320, 384, 448, 513, 100000]
    for x in range(8):
        sel = xcc == x
        row = []
        for lo, hi in zip(bins[:-1], bins[1:]):
            q = sel & (nown >= lo) & (nown < hi)
            row.append((int(q.sum()), round(float((end - staged)[q].mean()), 1) if q.any() else 0, round(float((staged - start)[q].mean()), 1) if q.any() else 0))
        print(x, row)
elif os.environ.get("TRACE_BRIEF"):
    print([(d["xcc"], d["beads"], round(d["last_end_us"]), round(d["busy_block_us"])) for d in out["per_xcd"]])
else:
    print(json.dumps(out, indent=1))
