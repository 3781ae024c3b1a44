"""Generate the golden fixtures under tests/golden/ (run in the build container).

Inputs come from the reference's only fixture, the examples/waterbox deck
(/root/reference/examples/waterbox: object.data, martini.data, snapshot.mem/*),
read through the host C deck loader with the deck's own commented alternatives
`nglf INTEGRATOR {type = NGLF;}` and a FREE group (the shipped
NGLFCONSTRAINT+LANGEVIN selection is non-deterministic and outside nglf.c).
Expected outputs come from the CPU oracle (oracle/ddc_oracle.c), which is
cross-checked here against its independent O(N^2) evaluation before anything is
written.  PARITY UNPINNED: the reference ships no expected outputs.

    python tests/golden/make_golden.py
"""
import os
import sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import ddcmd_amd  # noqa: E402
from ddcmd_amd.deck import load_deck, setup_to_dict  # noqa: E402
import pyoracle  # noqa: E402

DECK = "/root/reference/examples/waterbox/object.data"
EXTRA = "nglf INTEGRATOR {type = NGLF;}\n group GROUP { type = FREE; }\n free GROUP { type = FREE; }\n"
OUT = os.path.dirname(os.path.abspath(__file__))


def main():
    s = load_deck(DECK, None, EXTRA)
    o = pyoracle.Oracle(s)
    npairs = o.build_list()
    e, vir = o.forces()
    fx, fy, fz, vlj, vele, bvir, nin = o.brute_force()
    fmax = max(np.abs(o.fx).max(), np.abs(o.fy).max(), np.abs(o.fz).max())
    assert max(np.abs(fx - o.fx).max(), np.abs(fy - o.fy).max(), np.abs(fz - o.fz).max()) < 1e-12 * fmax
    assert abs(vlj - e["lj"]) < 1e-12 * abs(vlj)
    assert np.allclose(bvir, vir, rtol=1e-11, atol=0)
    d = setup_to_dict(s)
    d.update(gold_fx=o.fx.copy(), gold_fy=o.fy.copy(), gold_fz=o.fz.copy(),
             gold_e=np.array([e[k] for k in pyoracle.E_NAMES]), gold_virial=vir,
             gold_npairs_list=np.array(npairs[0]), gold_npairs_cut=np.array(nin))
    # 10-step NVE trace (deltaloop=10 of the deck), dt=20 fs
    rk0, tion0 = o.kinetic()
    trace = [[0, e["total"], rk0] + list(vir) + list(tion0)]
    for step in range(10):
        e2, v2, rk, tion = o.step(1)
        trace.append([step + 1, e2["total"], rk] + list(v2) + list(tion))
    d.update(gold_trace=np.array(trace), gold_rx10=o.rx.copy(), gold_ry10=o.ry.copy(), gold_rz10=o.rz.copy(),
             gold_vx10=o.vx.copy(), gold_vy10=o.vy.copy(), gold_vz10=o.vz.copy())
    np.savez_compressed(os.path.join(OUT, "waterbox.npz"), **d)
    print("waterbox.npz: natoms=%d list pairs=%d in-cut pairs=%d E=%.12g" % (s.natoms, npairs[0], nin, e["total"]))
    for row in trace:
        print("step %2d  Epot %.10f  Ekin %.10f  Etot %.10f" % (row[0], row[1], row[2], row[1] + row[2]))


if __name__ == "__main__":
    main()
