#!/usr/bin/env python3
"""Randomised edge-case hunt: water boxes of random shape, density pattern (slabs, droplets, vacuum), boundary
mask and size against the CPU oracle -- step-0 forces / energies / virial and a 25-step trajectory with one
rebuild.  python tools/fuzz_parity.py [ncases] [seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import pyoracle
import ddcmd_amd
from ddcmd_amd.martini import MartiniHIP, MartiniGroup, MartiniRank, _declare_domains

def run_cases(ncases, seed, verbose=True, domains=False, loopback=False):
    """returns (worst step-0 error, worst 25-step error, number of mismatching cases)"""
    rng = np.random.default_rng(seed)
    worst = 0.0
    worst_t = 0.0
    bad = 0
    for case in range(ncases):
        n = int(rng.integers(6, 13))
        rcut, skin = float(rng.choice([12.0, 12.0, 9.0, 11.0, 15.0])), float(rng.choice([4.0, 4.0, 2.0, 5.0]))      # list radius <= 20 A: a tile neighbourhood must fit the LDS (DESIGN.md limits)
        if n * 8.12 < 2.05 * (rcut + skin):          # the box must hold two list radii per periodic axis
            rcut, skin = 12.0, 4.0
        s = ddcmd_amd.make_water_setup(n, seed=int(rng.integers(1, 1 << 30)), temperature_K=float(rng.choice([50.0, 310.0])), rcut_A=rcut, skin_A=skin)
        L = s.h[0]
        fac = rng.choice([1.0, 1.0, 1.3, 1.9, 2.6], size=3)
        pbc = int(rng.choice([7, 7, 7, 0, 3, 5, 6, 1]))
        s.updateRate = int(rng.choice([20, 20, 7, 0]))           # 0: rebuilds when neighborCheck asks
        s.pbc = pbc
        s.h = np.array([L * fac[0], 0, 0, 0, L * fac[1], 0, 0, 0, L * fac[2]])
        keep = np.ones(s.natoms, bool)
        kind = rng.choice(["full", "slab", "droplet", "sparse"])
        if kind == "slab":
            keep = np.abs(s.rz) < rng.uniform(0.15, 0.45) * L
        elif kind == "droplet":
            keep = (s.rx ** 2 + s.ry ** 2 + s.rz ** 2) < (rng.uniform(0.2, 0.5) * L) ** 2
        elif kind == "sparse":
            keep = rng.random(s.natoms) < rng.uniform(0.02, 0.5)
        if keep.sum() < 2:
            keep[:2] = True
        for k in ("rx", "ry", "rz", "vx", "vy", "vz", "gid", "species", "group"):
            setattr(s, k, np.ascontiguousarray(getattr(s, k)[keep]))
        s.natoms = int(keep.sum())
        shift = rng.uniform(-0.5, 0.5, 3) * np.array([s.h[0], s.h[4], s.h[8]]) * [(pbc >> a) & 1 for a in range(3)]      # periodic axes: anywhere in the box
        s.rx = s.rx + shift[0]; s.ry = s.ry + shift[1]; s.rz = s.rz + shift[2]
        o = pyoracle.Oracle(s)
        o.L.orc_back_in_box(__import__("ctypes").byref(o.p), o.n, pyoracle._d(o.rx), pyoracle._d(o.ry), pyoracle._d(o.rz))
        e0, v0 = o.forces()
        grid = (1, 1, 1)
        if domains:      # emulated domains, some of them possibly empty; a domain must be at least a list radius wide
            box = np.array([s.h[0], s.h[4], s.h[8]])
            grid = tuple(int(g) if box[a] / g >= 1.05 * (s.rmax + s.deltaR) else 1 for a, g in enumerate(rng.choice([1, 2, 2, 3], size=3)))
        try:
            if loopback:     # the periodic images travel through a 1-rank RCCL communicator (the multi-GPU transport)
                import ctypes
                os.environ["DDCMI_RCCL_LOOPBACK"] = "1"
                m = MartiniRank(s, np.arange(s.natoms))
                _declare_domains(m.lib)
                buf = ctypes.create_string_buffer(128)
                assert m.lib.ddcmi_comm_unique_id(buf) == 0
                m.comm_init(0, 1, buf.raw, (1, 1, 1))
                m.upload_local()
                e, vir = m.eval_forces()
                pp = m.download_particles()
                fg = np.stack(pp["f"])[:, np.argsort(pp["gid"], kind="stable")]
                fo = np.stack((o.fx, o.fy, o.fz))[:, np.argsort(s.gid, kind="stable")]
                grid = "rccl"
            elif grid == (1, 1, 1):
                m = MartiniHIP(s)
                e, vir = m.eval_forces()
                fg = np.stack(m.download()["f"])
                fo = np.stack((o.fx, o.fy, o.fz))
            else:
                m = MartiniGroup(s, grid)
                e, vir = m.eval_forces()
                fg = np.stack(m.gather()["f"])
                fo = np.stack((o.fx, o.fy, o.fz))[:, np.argsort(s.gid, kind="stable")]
        except Exception as ex:
            if "LDS" in str(ex):      # a stated limit: coarse cells of a narrow domain / a long list radius
                if verbose:
                    print("case %2d skipped: %s" % (case, str(ex)[-90:]), flush=True)
                continue
            raise
        scale = max(np.abs(fo).max(), 1e-30)
        err_f = np.abs(fg - fo).max() / scale
        err_e = abs(e["total"] - e0["total"]) / max(abs(e0["total"]), 1e-12)
        eo, vo, rko, _ = o.step(25)
        m.step(25)
        e2, vir2, rk, _ = m.energies()
        err_t = abs(e2["total"] - eo["total"]) / max(abs(eo["total"]), 1e-12)
        err_k = abs(rk - rko) / max(rko, 1e-12)
        m.close()
        worst = max(worst, err_f, err_e)
        worst_t = max(worst_t, err_t, err_k)
        flag = "" if (err_f < 1e-9 and err_e < 1e-9 and err_t < 1e-6 and err_k < 1e-6) else "   <-- MISMATCH"
        bad += bool(flag)
        if verbose:
          print("case %2d n=%d rc %.0f+%.0f beads=%6d grid %s box x%.1f x%.1f x%.1f pbc=%d %-7s dF %.1e dE %.1e | 25 steps dE %.1e dKE %.1e%s" % (
            case, n, rcut, skin, s.natoms, grid, fac[0], fac[1], fac[2], pbc, kind, err_f, err_e, err_t, err_k, flag), flush=True)
    return worst, worst_t, bad


if __name__ == "__main__":
    w, wt, bad = run_cases(int(sys.argv[1]) if len(sys.argv) > 1 else 30, int(sys.argv[2]) if len(sys.argv) > 2 else 1, domains=len(sys.argv) > 3 and sys.argv[3] == "domains", loopback=len(sys.argv) > 3 and sys.argv[3] == "loopback")
    print("worst step-0 error %.2e, worst 25-step error %.2e, %d mismatching cases" % (w, wt, bad))
