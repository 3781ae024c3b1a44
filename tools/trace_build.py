"""Per-workgroup timeline of k_tile_build from a -DDDCMI_TRACE_BLOCKS build (tools/build_variants.sh trace "-DDDCMI_TRACE_BLOCKS"):
   DDCMI_LIB=ddcmd_amd/lib/variants/libddcmi_trace.so python3 tools/trace_build.py <lattice> [loopback]"""
import ctypes, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
if len(sys.argv) > 2: os.environ["DDCMI_RCCL_LOOPBACK"] = "1"
from ddcmd_amd.synth import make_water_setup
from ddcmd_amd.martini import MartiniHIP
from ddcmd_amd._lib import load_library

n = int(sys.argv[1]) if len(sys.argv) > 1 else 50
if os.environ.get("TRACE_WORKLOAD") == "lipid":      # the lipid deck tiled n x n x n/2
    from ddcmd_amd.deck import load_deck
    from ddcmd_amd.synth import replicate_setup
    deck = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "lipid_deck")
    setup = replicate_setup(load_deck(os.path.join(deck, "object_nvt.data"), restart_file=os.path.join(deck, "relaxed", "restart")), (n, n, max(1, n // 2)))
else:
    setup = make_water_setup(n)
m = MartiniHIP(setup)
m.eval_forces()
m.step(45)          # two more rebuilds in a running simulation: the last one is what is read
m.sync()
lib = load_library()
nb = 65536
buf = np.zeros((nb, 8), dtype=np.uint64)
lib.ddcmi_debug_trace_build.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert lib.ddcmi_debug_trace_build(buf.ctypes.data, nb) == 0
b = buf[buf[:, 0] > 0].astype(np.int64)
act = b[b[:, 2] > 0]
t0 = b[:, 0].min()
start, staged, end = (act[:, 0] - t0) / 100.0, (act[:, 1] - t0) / 100.0, (act[:, 2] - t0) / 100.0
cyc = act[:, 3]
nown, nst = act[:, 6], act[:, 7]
hw = act[:, 5]; xcc = act[:, 4] & 0xf
key = ((xcc * 8 + ((hw >> 13) & 7)) * 2 + ((hw >> 12) & 1)) * 16 + ((hw >> 8) & 0xf)
print("lattice %d: %d workgroups, %d with owned beads; span of the launch %.1f us (first start %.1f, last start %.1f)" % (n, len(b), len(act), end.max(), start.min(), start.max()))
dur = end - start
print("active workgroups: duration mean %.1f / median %.1f / max %.1f us; staging mean %.1f us; shader clock %.0f MHz (cycles / wall)" %
      (dur.mean(), np.median(dur), dur.max(), (staged - start).mean(), np.median(cyc / np.maximum(dur, 1e-3))))
full = nown > 400
if full.any(): print("full tiles (%d): duration mean %.1f us, staged beads %.0f, owned %.0f" % (full.sum(), dur[full].mean(), nst[full].mean(), nown[full].mean()))
u, c = np.unique(key, return_counts=True)
print("CUs used %d; active workgroups per CU: mean %.2f max %d" % (len(u), c.mean(), c.max()))
# how long does a workgroup take as a function of how many it shared the CU with?
for lo, hi in ((0, 100), (100, 300), (300, 450), (450, 600)):
    sel = (nown >= lo) & (nown < hi)
    if sel.any(): print("   owned beads %3d-%3d: %4d workgroups, %.1f us mean, %.2f us per owned bead" % (lo, hi, sel.sum(), dur[sel].mean(), (dur[sel] / np.maximum(nown[sel], 1)).mean()))
order = np.argsort(start)
print("start-time deciles (us):", np.round(np.percentile(start, [0, 10, 50, 90, 100]), 1), " end-time deciles:", np.round(np.percentile(end, [0, 10, 50, 90, 100]), 1))
