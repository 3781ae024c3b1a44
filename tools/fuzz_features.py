#!/usr/bin/env python3
"""Feature combinations on the lipid deck against the CPU oracle: thermostat kind x constraints x barostat
(x restraints), 25 steps each across two rebuilds.  python tools/fuzz_features.py"""
import os, sys, itertools
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import pyoracle
import ddcmd_amd
from ddcmd_amd.deck import load_deck, units_convert
from ddcmd_amd.martini import MartiniHIP


def run(verbose=True):
    from test_oracle import CONSTRAINT_X, RESTRAINT_X
    deck = os.path.join(ROOT, "tests", "golden", "lipid_deck")
    bad = 0
    worst = 0.0
    for thermo, cons, baro, rest in itertools.product(("free", "berendsen", "langevin"), (False, True), (False, True), (False, True)):
        extra = (CONSTRAINT_X if cons else "") + (RESTRAINT_X if rest else "")
        s = load_deck(os.path.join(deck, "object_nvt.data"), restart_file=os.path.join(deck, "relaxed", "restart"), extra_objects=extra or None)
        s.group_type = np.full(s.ngroup, {"free": 0, "berendsen": 1, "langevin": 2}[thermo], np.int32)
        s.group_Teq = np.full(s.ngroup, units_convert(310.0, "K")); s.group_tau = np.full(s.ngroup, units_convert(1.0, "ps"))
        s.rng_seed = 424242
        T, P0, beta, tau = units_convert(310.0, "K"), units_convert(1.0, "bar"), units_convert(3.0e-4, "1/bar") * 20.0, units_convert(1.0, "ps")
        o = pyoracle.Oracle(s, constraints=cons)
        o.forces(); o.group_temperature()
        m = MartiniHIP(s, constraints=cons)
        if baro:
            m.set_barostat(T, P0, beta, tau)
        m.eval_forces(); m.group_temperatures()
        err = 0.0
        for blk in range(5):
            eo, vo, rko, _ = o.step_npt(5, T, P0, beta, tau, molecular=True) if baro else o.step(5)
            m.step(5 if blk % 2 else 2)
            if blk % 2 == 0:
                m.step(3)
            o.group_temperature(); m.group_temperatures()
            e, vir, rk, _ = m.energies()
            err = max(err, abs(e["total"] - eo["total"]) / abs(eo["total"]), abs(rk - rko) / rko, np.abs(vir - vo).max() / np.abs(vo).max())
            if baro:
                err = max(err, np.abs(m.box() - o.box).max() / o.box.max())
        m.close()
        ok = err < 1e-6
        bad += not ok
        worst = max(worst, err)
        if verbose:
            print("%-9s constraints=%d barostat=%d restraints=%d: worst relative error over 25 steps %.1e%s" % (thermo, cons, baro, rest, err, "" if ok else "   <-- MISMATCH"), flush=True)
    return worst, bad


if __name__ == "__main__":
    w, bad = run()
    print("worst %.2e, %d mismatching combinations" % (w, bad))
