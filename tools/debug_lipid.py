import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle"))
import torch  # noqa
import pyoracle
from ddcmd_amd.deck import load_deck
from ddcmd_amd.synth import replicate_setup
from ddcmd_amd.martini import MartiniHIP
deck = "tests/golden/lipid_deck"
s0 = load_deck(os.path.join(deck, "object_nvt.data"), restart_file=os.path.join(deck, "relaxed", "restart"))
o = pyoracle.Oracle(s0); e0, v0 = o.forces()
reps = tuple(int(x) for x in sys.argv[1].split(","))
ncopy = reps[0] * reps[1] * reps[2]
s = replicate_setup(s0, reps)
m = MartiniHIP(s)
e, vir = m.eval_forces()
d = m.download()
f = np.stack(d["f"]).reshape(3, ncopy, s0.natoms)
ref = np.stack([o.fx, o.fy, o.fz])[:, None, :]
err = np.abs(f - ref).max(axis=(0, 2)) / np.abs(ref).max()
print("force err per copy: max", err.max(), "copies bad", int((err > 1e-8).sum()), "of", ncopy, "list stats", m.list_stats())
print({k: (e[k], ncopy * e0[k]) for k in ("lj", "ele", "bond", "angle")})
bad = np.argwhere(np.abs(f - ref) > 1e-6 * np.abs(ref).max())
print("bad entries", len(bad), bad[:10])
m.group_temperatures()
for k in range(12):
    m.step(1)
    ee, _, rk, _ = m.energies()
    print(k, ee["total"] / ncopy, rk / ncopy, m.list_stats()["rebuilds"])
