"""How much of the pair kernel's wave time goes to rows shorter than their wave's longest?  (round 6)
Runs a water box to its liquid state, takes the full list's row lengths (ddcmi_get_list), orders the beads as the device does (tile-major cells,
8x4x4 cells per tile) and forms, per chunk of 64 rows of a tile, the groups of 8 slots each lane walks against the wave's longest row."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, ddcmd_amd
from ddcmd_amd.martini import MartiniHIP
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
s = ddcmd_amd.make_water_setup(n)
m = MartiniHIP(s); m.eval_forces(); m.step(220)      # (a rebuild at step 220: the list is fresh)
start, _ = m.get_list(0)
cnt = np.diff(start)
d = m.download(1)
L = s.h[0]; nc = int(np.floor(L / (0.5 * (s.rmax + s.deltaR)))); c = L / nc
ix = [np.clip(np.floor((np.asarray(d['r'][k]) - L * np.rint(np.asarray(d['r'][k]) / L) + L / 2) / c).astype(np.int64), 0, nc - 1) for k in range(3)]
TC = (8, 4, 4)
T = [-(-nc // TC[k]) for k in range(3)]
tile = ((ix[2] // 4) * T[1] + ix[1] // 4) * T[0] + ix[0] // 8
lc = ((ix[2] % 4) * 4 + ix[1] % 4) * 8 + ix[0] % 8
order = np.lexsort((lc, tile))
tl, ct = tile[order], cnt[order]
grp = (ct + 7) // 8
bounds = np.flatnonzero(np.concatenate(([True], tl[1:] != tl[:-1], [True])))
walked = 0; paid = 0; paid_sorted = 0; lanes = 0
for a, b in zip(bounds[:-1], bounds[1:]):
    g = grp[a:b]
    for k in range(0, len(g), 64):
        w = g[k:k + 64]
        walked += w.sum(); paid += 64 * w.max(); lanes += len(w) * w.max()
    gs = np.sort(g)[::-1]
    for k in range(0, len(gs), 64):
        w = gs[k:k + 64]
        paid_sorted += 64 * w.max()
print("beads %d  entries/bead %.1f  beads/tile %.0f" % (s.natoms, cnt.mean(), s.natoms / (len(bounds) - 1)))
print("lane-groups walked / (64 x the wave's longest row): %.3f   -- of which partial last chunks of the tiles: %.3f" % (walked / paid, lanes / paid))
print("the same with a tile's rows sorted by length before they are cut into waves: %.3f" % (walked / paid_sorted))
m.close()
