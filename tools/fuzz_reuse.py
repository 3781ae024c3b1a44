#!/usr/bin/env python3
"""One device context reused for a sequence of different systems (sizes growing and shrinking, other boxes and
boundary masks): every upload must behave like a fresh context -- stale capacities, lists, schedules or flags show
as a mismatch with the oracle.  python tools/fuzz_reuse.py [nsystems] [seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import pyoracle
import ddcmd_amd
from ddcmd_amd.martini import MartiniHIP, _d


def run(nsys, seed, verbose=True):
    rng = np.random.default_rng(seed)
    m = None
    worst, bad = 0.0, 0
    for k in range(nsys):
        n = int(rng.integers(6, 14))
        s = ddcmd_amd.make_water_setup(n, seed=int(rng.integers(1, 1 << 30)), temperature_K=310.0)
        L = s.h[0]
        fac = rng.choice([1.0, 1.0, 1.4, 2.0], size=3)
        s.pbc = int(rng.choice([7, 7, 3, 5, 0]))
        s.h = np.array([L * fac[0], 0, 0, 0, L * fac[1], 0, 0, 0, L * fac[2]])
        if rng.random() < 0.5:
            keep = rng.random(s.natoms) < rng.uniform(0.05, 0.9)
            keep[:2] = True
            for a in ("rx", "ry", "rz", "vx", "vy", "vz", "gid", "species", "group"):
                setattr(s, a, np.ascontiguousarray(getattr(s, a)[keep]))
            s.natoms = int(keep.sum())
        o = pyoracle.Oracle(s)
        e0, v0 = o.forces()
        if m is None:
            m = MartiniHIP(s)
        else:                                  # same context: new box, new beads
            m.s = s; m.n = s.natoms
            m._chk(m.lib.ddcmi_set_box(m.ctx, _d(np.ascontiguousarray(s.h, dtype=np.float64)), int(s.pbc)))
            m._chk(m.lib.ddcmi_set_clock(m.ctx, int(s.loop), float(s.time)))
            m.upload(s.rx, s.ry, s.rz, s.vx, s.vy, s.vz)
        e, vir = m.eval_forces()
        fg = np.stack(m.download()["f"]); fo = np.stack((o.fx, o.fy, o.fz))
        err = max(np.abs(fg - fo).max() / max(np.abs(fo).max(), 1e-30), abs(e["total"] - e0["total"]) / max(abs(e0["total"]), 1e-12))
        eo, vo, rko, _ = o.step(23)
        m.step(23)
        e2, _, rk, _ = m.energies()
        err_t = max(abs(e2["total"] - eo["total"]) / max(abs(eo["total"]), 1e-12), abs(rk - rko) / max(rko, 1e-12))
        ok = err < 1e-9 and err_t < 1e-6
        bad += not ok; worst = max(worst, err, err_t)
        if verbose:
            print("system %2d: %6d beads pbc %d box x%.1f x%.1f x%.1f  step-0 %.1e  23 steps %.1e%s" % (k, s.natoms, s.pbc, fac[0], fac[1], fac[2], err, err_t, "" if ok else "   <-- MISMATCH"), flush=True)
    m.close()
    return worst, bad


if __name__ == "__main__":
    w, bad = run(int(sys.argv[1]) if len(sys.argv) > 1 else 20, int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    print("worst %.2e, %d mismatching systems" % (w, bad))
