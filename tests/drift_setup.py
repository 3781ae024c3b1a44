"""The drifting systems of the migration soak (tools/soak_migration_r06.py), shared by tests/mp_worker.py (one rank of a run between real processes) and
its parent test, which runs the same system on one domain.  No oracle here: the worker must not import it."""
import os
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def drifting_setup(workload, grid):
    """a cube of water in a box three times its size, or the relaxed bilayer patch tiled 2x2 with vacuum above and
    below, drifting about a third of a brick per rebuild period along every axis: the ranks' bead counts swing between (nearly) nothing and everything"""
    from ddcmd_amd.synth import make_water_setup, replicate_setup
    from ddcmd_amd.deck import load_deck
    if workload == "lipid_drift":
        deck = os.path.join(ROOT, "tests", "golden", "lipid_deck")
        s = replicate_setup(load_deck(os.path.join(deck, "object_nvt.data"), restart_file=os.path.join(deck, "relaxed", "restart")), (2, 2, 1))
        s.h = np.array(s.h, dtype=np.float64)
        bricks = np.array([s.h[0], s.h[4], 3.0 * s.h[8]]) / np.array(grid)
        s.h[8] *= 3.0
    else:
        s = make_water_setup(12, temperature_K=300.0)
        s.h = np.array(s.h, dtype=np.float64) * 3.0
        bricks = np.array([s.h[0], s.h[4], s.h[8]]) / np.array(grid)
    drift = np.array([0.37, 0.23, 0.31]) * bricks / (int(s.updateRate) * s.dt)
    s.vx = np.asarray(s.vx) + drift[0]; s.vy = np.asarray(s.vy) + drift[1]; s.vz = np.asarray(s.vz) + drift[2]
    return s
