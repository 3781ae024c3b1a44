#!/usr/bin/env python3
"""Randomised parity of the molecular path: the lipid deck tiled a random number of times, shifted rigidly by
a random vector (so molecules straddle other faces, tiles and domain boundaries), on a random grid of emulated
domains -- step-0 forces, every energy kind, virial and a 20-step trajectory against the CPU oracle.
python tools/fuzz_lipid.py [ncases] [seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import pyoracle
import ddcmd_amd
from ddcmd_amd.deck import load_deck
from ddcmd_amd.synth import replicate_setup
from ddcmd_amd.martini import MartiniHIP, MartiniGroup


def run_cases(ncases, seed, verbose=True):
    rng = np.random.default_rng(seed)
    deck = os.path.join(ROOT, "tests", "golden", "lipid_deck")
    base = load_deck(os.path.join(deck, "object_nvt.data"), restart_file=os.path.join(deck, "relaxed", "restart"))
    worst, worst_t, bad = 0.0, 0.0, 0
    for case in range(ncases):
        reps = tuple(int(x) for x in rng.choice([1, 1, 2], size=3))
        s = replicate_setup(base, reps) if reps != (1, 1, 1) else replicate_setup(base, (1, 1, 1))
        box = np.array([s.h[0], s.h[4], s.h[8]])
        shift = rng.uniform(-0.5, 0.5, 3) * box
        for c, k in enumerate(("rx", "ry", "rz")):
            x = getattr(s, k) + shift[c]
            setattr(s, k, x - box[c] * np.rint(x / box[c]))
        grid = tuple(int(x) for x in rng.choice([1, 1, 2], size=3))
        if any(box[a] / grid[a] < 2.1 * (s.rmax + s.deltaR) for a in range(3)):
            grid = (1, 1, 1)
        o = pyoracle.Oracle(s)
        e0, v0 = o.forces()
        order = np.argsort(s.gid, kind="stable")
        fo = np.stack((o.fx, o.fy, o.fz))[:, order]
        if grid == (1, 1, 1):
            m = MartiniHIP(s); e, vir = m.eval_forces(); fg = np.stack(m.download()["f"])[:, order]
        else:
            m = MartiniGroup(s, grid); e, vir = m.eval_forces(); fg = np.stack(m.gather()["f"])
        err_f = np.abs(fg - fo).max() / np.abs(fo).max()
        err_e = max(abs(e[k] - e0[k]) / max(abs(e0[k]), 1e-9 * abs(e0["total"])) for k in ("lj", "ele", "bond", "angle", "tors", "impr", "total"))
        err_v = np.abs(vir - v0).max() / np.abs(v0).max()
        o.group_temperature(); m.group_temperatures()
        eo, vo, rko, _ = o.step(20)
        m.step(20)
        e2, vir2, rk, _ = m.energies()
        err_t = max(abs(e2["total"] - eo["total"]) / abs(eo["total"]), abs(rk - rko) / rko)
        m.close()
        worst = max(worst, err_f, err_e, err_v); worst_t = max(worst_t, err_t)
        ok = err_f < 1e-9 and err_e < 1e-9 and err_v < 1e-9 and err_t < 1e-6
        bad += not ok
        if verbose:
            print("case %2d reps %s grid %s beads %6d: dF %.1e dE %.1e dVir %.1e | 20 steps %.1e%s" % (case, reps, grid, s.natoms, err_f, err_e, err_v, err_t, "" if ok else "   <-- MISMATCH"), flush=True)
    return worst, worst_t, bad


if __name__ == "__main__":
    w, wt, bad = run_cases(int(sys.argv[1]) if len(sys.argv) > 1 else 12, int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    print("worst step-0 error %.2e, worst 20-step error %.2e, %d mismatching cases" % (w, wt, bad))
