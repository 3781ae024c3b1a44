"""Several contexts driven from several host threads at once (a plugin host may well do that): every context is its own world -- streams, buffers, the
host mailbox, the error text; the process-wide tables (kernel attributes asked once per device and kernel, the roctx marker library) are behind mutexes.
ctypes releases the GIL inside every call, so the threads really overlap in the library."""
import threading

import numpy as np
import pytest

from ddcmd_amd.martini import MartiniHIP
from ddcmd_amd.synth import make_water_setup

pytestmark = pytest.mark.gpu


def _run(s, nblocks, out, k, barrier=None):
    try:
        m = MartiniHIP(s)
        e = [m.eval_forces()[0]["lj"]]
        for _ in range(nblocks):
            if barrier is not None:
                barrier.wait()
            m.step(23)
            ee, vir, rk, _ = m.energies()
            e += [ee["lj"], rk]
        d = m.download()
        m.close()
        out[k] = (np.array(e), np.concatenate(d["r"] + d["v"] + d["f"]))
    except Exception as ex:      # (a thread's exception must reach the test)
        out[k] = ex


def test_contexts_on_threads_equal_the_same_runs_one_after_the_other():
    setups = [make_water_setup(n, seed=100 + n) for n in (6, 7, 8, 9, 7, 6)]
    serial, threaded = {}, {}
    for k, s in enumerate(setups):
        _run(s, 4, serial, k)
    barrier = threading.Barrier(len(setups))
    ts = [threading.Thread(target=_run, args=(s, 4, threaded, k, barrier)) for k, s in enumerate(setups)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=600)
    for k in range(len(setups)):
        assert not isinstance(serial[k], Exception) and not isinstance(threaded.get(k), Exception), (k, serial[k], threaded.get(k))
        assert np.array_equal(serial[k][0], threaded[k][0]) and np.array_equal(serial[k][1], threaded[k][1]), k      # bit for bit: a run repeats


def test_contexts_created_and_destroyed_on_threads_while_others_run():
    """eight threads, each creating, running and destroying five contexts in a row (lipid deck and water alternating): creation and destruction of one
    context -- allocations, frees, their implicit synchronisations -- beside the launches of the others"""
    import os
    from ddcmd_amd.deck import load_deck
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lipid = load_deck(os.path.join(root, "tests", "golden", "lipid_deck", "object_nvt.data"), restart_file=os.path.join(root, "tests", "golden", "lipid_deck", "relaxed", "restart"))
    systems = [lipid, make_water_setup(7, seed=5), lipid, make_water_setup(8, seed=6), lipid]
    ref = {}
    for k, s in enumerate(systems):
        _run(s, 2, ref, k)
        assert not isinstance(ref[k], Exception), ref[k]
    results = [dict() for _ in range(8)]

    def worker(t):
        for k, s in enumerate(systems):
            _run(s, 2, results[t], k)
    ts = [threading.Thread(target=worker, args=(t,)) for t in range(8)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=900)
    for t in range(8):
        for k in range(len(systems)):
            got = results[t].get(k)
            assert got is not None and not isinstance(got, Exception), (t, k, got)
            assert np.array_equal(got[0], ref[k][0]) and np.array_equal(got[1], ref[k][1]), (t, k)
