"""What a setter invalidates must really be rebuilt, whatever came before: a long-lived context against a fresh one (tools/fuzz_sequence.py, round 6).
Its first pass found that new charges (ddcmi_set_species) or new reaction-field constants (ddcmi_set_nonbonded) under an uploaded state kept the self
term -1/2 sum q^2 keR crf of the upload, and that a force evaluation right behind such a call ran with the class tables of the old parameters."""
import copy
import os
import subprocess
import sys

import numpy as np
import pytest

import ddcmd_amd.martini as martini
from ddcmd_amd.martini import MartiniHIP, _d, _i
from ddcmd_amd.deck import load_deck

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
f64 = lambda a: np.ascontiguousarray(a, dtype=np.float64)
i32 = lambda a: np.ascontiguousarray(a, dtype=np.int32)


def _deck():
    d = os.path.join(ROOT, "tests", "golden", "lipid_deck")
    return load_deck(os.path.join(d, "object_nvt.data"), restart_file=os.path.join(d, "relaxed", "restart"))


def _same(a, b, tol=1e-10):
    ea, eb = a.eval_forces()[0], b.eval_forces()[0]
    da, db = a.download(), b.download()
    fmax = max(np.abs(db["f"][c]).max() for c in range(3))
    assert max(np.abs(da["f"][c] - db["f"][c]).max() for c in range(3)) < tol * fmax
    for k in eb:
        assert abs(ea[k] - eb[k]) < tol * max(abs(eb["total"]), abs(eb["lj"])), (k, ea[k], eb[k])


def test_new_charges_and_constants_under_an_uploaded_state():
    s = _deck()
    live = MartiniHIP(s)
    e0, _ = live.eval_forces()
    assert abs(e0["ele"]) > 1e-3
    live.step(3)
    d = live.download()
    now = copy.deepcopy(s)
    now.rx, now.ry, now.rz = d["r"]; now.vx, now.vy, now.vz = d["v"]
    for charge_scale, crf_scale in ((0.0, 1.0), (1.0, 1.0), (0.5, 1.0), (1.0, 0.5), (1.0, 1.0)):
        now.charge = np.asarray(s.charge) * charge_scale
        now.crf = s.crf * crf_scale
        live._chk(live.lib.ddcmi_set_species(live.ctx, now.nspecies, _d(f64(now.mass)), _d(f64(now.charge)), _i(i32(now.ljtype)), _i(i32(now.moltype))))
        live._chk(live.lib.ddcmi_set_nonbonded(live.ctx, now.nlj, _d(f64(now.sigma)), _d(f64(now.eps)), _d(f64(now.shift)), now.rmax, now.keR, now.krf, now.crf))
        with pytest.raises(martini.DdcmiError):      # the forces on the device are those of the old parameters: a step asks for a new evaluation first
            live.step(1)
        fresh = MartiniHIP(now)
        _same(live, fresh)
        if charge_scale == 0.0:
            assert live.eval_forces()[0]["ele"] == 0.0
        fresh.close()
    live.step(2)      # ... and goes on
    live.close()


def test_a_pass_of_the_call_sequence_fuzz():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_sequence.py"), "16", "1", "24"], cwd=ROOT, capture_output=True, text=True, timeout=900)
    tail = "\n".join(r.stdout.strip().splitlines()[-6:])
    assert r.returncode == 0 and " 0 differ from a fresh context" in tail, tail + r.stderr[-2000:]


@pytest.mark.parametrize("family", ["bricks", "loopback"])
def test_call_sequences_on_decomposed_contexts(family):
    """the same on 2x2x2 bricks of an in-process group and on a rank of the RCCL loopback (the calls a decomposed run can take in mid-run: neighbour settings, cut-off,
    LJ entries, charges, masses, thermostats, molecule tables).  The group used to evaluate with the class tables of the old parameters (its rebuild does not pass
    through ddcmi_build_list); it refreshes them like ddcmi_eval_forces now, and its step asks for forces that are valid"""
    env = dict(os.environ, FUZZ_ONLY=family)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_sequence.py"), "8", "21", "24"], cwd=ROOT, capture_output=True, text=True, timeout=900, env=env)
    tail = "\n".join(r.stdout.strip().splitlines()[-6:])
    assert r.returncode == 0 and " 0 differ from a fresh context" in tail, tail + r.stderr[-2000:]
