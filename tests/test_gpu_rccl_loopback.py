"""GPU test (-m gpu) of the RCCL transport on ONE GPU: with DDCMI_RCCL_LOOPBACK=1 a single rank
reaches its periodic neighbours through a 1-rank RCCL communicator instead of local image copies,
so the count exchange, the grouped ncclSend/ncclRecv of migration, halo build and per-step halo
refresh, the message-matching order and ncclAllReduce all execute for real.  (A multi-GPU node is
not available to the tests; the in-process group of test_gpu_domains.py covers the multi-domain
logic, this covers the wire.)"""
import ctypes
import os
import numpy as np
import pytest

import pyoracle
from ddcmd_amd.synth import make_water_setup
from conftest import rel_force_err

pytestmark = pytest.mark.gpu
TOL = 1e-6


def _loopback_rank(s, monkeypatch, overlap=False):
    from ddcmd_amd.martini import MartiniRank, _declare_domains
    monkeypatch.setenv("DDCMI_RCCL_LOOPBACK", "1")
    monkeypatch.setenv("DDCMI_HALO_OVERLAP", "1" if overlap else "0")
    m = MartiniRank(s, np.arange(s.natoms))
    _declare_domains(m.lib)
    buf = ctypes.create_string_buffer(128)
    assert m.lib.ddcmi_comm_unique_id(buf) == 0
    m.comm_init(0, 1, buf.raw, (1, 1, 1))
    m.upload_local()
    return m


def test_preflight_through_rccl_loopback(monkeypatch):
    """ddcmi_comm_preflight over RCCL itself -- the grouped ncclSend/ncclRecv along all 26 directions (to the one rank
    there is), ncclAllReduce of 24 doubles, ncclAllGather of the count block, queued with an event behind each stage and polled under
    a deadline -- and its fault injection: a spoiled message is named with its sender and direction"""
    from ddcmd_amd.martini import MartiniRank, DdcmiError
    s = make_water_setup(8)
    monkeypatch.setenv("DDCMI_RCCL_LOOPBACK", "1")
    for corrupt in (None, 22):
        if corrupt is not None:
            monkeypatch.setenv("DDCMI_DEBUG_HOOKS", "1")
            monkeypatch.setenv("DDCMI_DEBUG_PREFLIGHT_CORRUPT", str(corrupt))
        m = MartiniRank(s, np.arange(s.natoms))
        buf = ctypes.create_string_buffer(128)
        assert m.lib.ddcmi_comm_unique_id(buf) == 0
        m.comm_init(0, 1, buf.raw, (1, 1, 1))
        if corrupt is None:
            rep = m.preflight(timeout=30.0)
            assert rep["stages_verified"] == 3 and rep["directions"] == 26 and rep["peers"] == [0] and rep["bytes_per_direction"] == 4096
            assert m.comm_stats()["transport"] == "rccl-loopback"
            # the communicator is as good as new: the run goes on
            m.upload_local()
            m.eval_forces()
            m.step(3)
        else:
            with pytest.raises(DdcmiError) as ei:
                m.preflight(timeout=30.0)
            # code 22 = (0, 0, +1): received in slot 22 from the rank in my direction (0, 0, -1)
            assert "the message from rank 0 (my direction (+0,+0,-1), its direction code 22) is wrong at element 7" in str(ei.value), str(ei.value)
            assert m.preflight_report["failed_peer"] == 0 and m.preflight_report["failed_direction_code"] == 4
        m.close()
    monkeypatch.delenv("DDCMI_DEBUG_PREFLIGHT_CORRUPT", raising=False)
    monkeypatch.delenv("DDCMI_DEBUG_HOOKS", raising=False)


@pytest.mark.parametrize("overlap", [False, True])
def test_water_through_rccl_loopback(monkeypatch, overlap):
    """overlap=True: the exchange runs on a second stream under the tiles with all-owned neighbourhoods"""
    s = make_water_setup(12)
    o = pyoracle.Oracle(s)
    e0, v0 = o.forces()
    m = _loopback_rank(s, monkeypatch, overlap)
    e, vir = m.eval_forces()
    p = m.download_particles()
    order = np.argsort(p["gid"], kind="stable")
    f = [p["f"][c][order] for c in range(3)]
    assert np.array_equal(p["gid"][order], np.sort(s.gid))
    assert rel_force_err(f, (o.fx, o.fy, o.fz)) < 1e-10
    assert abs(e["lj"] - e0["lj"]) < 1e-10 * abs(e0["lj"])
    # 45 steps: beads leave the box and come back in through the migration messages
    for block in range(3):
        eo, vo, rko, _ = o.step(15)
        m.step(15)
        e, vir, rk, _ = m.energies()
        assert abs(e["total"] - eo["total"]) < TOL * abs(eo["total"]), block
        assert abs(rk - rko) < TOL * rko
    tot = m.allreduce([1.0, 2.5])          # ncclAllReduce over the 1-rank communicator
    assert tot[0] == 1.0 and tot[1] == 2.5
    assert int(m.lib.ddcmi_nlocal(m.ctx)) == s.natoms
    m.close()


def test_lipid_deck_through_rccl_loopback(monkeypatch):
    """bonded terms by gid + Berendsen group temperature (all-reduced) over the loopback wire"""
    from ddcmd_amd.deck import load_deck
    deck = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "lipid_deck")
    s = load_deck(os.path.join(deck, "object_nvt.data"), restart_file=os.path.join(deck, "relaxed", "restart"))
    o = pyoracle.Oracle(s)
    e0, v0 = o.forces()
    o.group_temperature()
    m = _loopback_rank(s, monkeypatch, overlap=True)
    e, vir = m.eval_forces()
    for k in ("lj", "ele", "bond", "angle", "tors", "impr", "total"):
        assert abs(e[k] - e0[k]) < 1e-9 * max(abs(e0[k]), 1e-12), k
    Tg = m.group_temperatures()
    assert abs(Tg[0] - o.groups[0].temperature) < 1e-10 * Tg[0]
    for block in range(2):
        eo, vo, rko, _ = o.step(10)
        m.step(10)
        e, vir, rk, _ = m.energies()
        assert abs(e["total"] - eo["total"]) < TOL * abs(eo["total"]), block
        assert abs(rk - rko) < TOL * rko
        o.group_temperature()
        m.group_temperatures()
    m.close()


def test_nglfconstraint_through_rccl_loopback(monkeypatch):
    """constraint groups and molecules named by gid over the RCCL wire: the velocity halo before each constraint solve and the
    all-reduce of the barostat's sums (virial, molecular term, split-molecule {P, F}) are real RCCL calls"""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from test_oracle import CONSTRAINT_X
    from ddcmd_amd.deck import load_deck, units_convert
    from ddcmd_amd.martini import MartiniRank, _declare_domains
    deck = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "lipid_deck")
    s = load_deck(os.path.join(deck, "object_nvt.data"), restart_file=os.path.join(deck, "relaxed", "restart"), extra_objects=CONSTRAINT_X)
    T, P0 = units_convert(310.0, "K"), units_convert(1.0, "bar")
    beta, tau = units_convert(3.0e-4, "1/bar") * 20.0, units_convert(1.0, "ps")
    s.npt_T, s.npt_P0, s.npt_beta, s.npt_tau = T, P0, beta, tau
    o = pyoracle.Oracle(s, constraints=True)
    o.forces()
    o.group_temperature()
    monkeypatch.setenv("DDCMI_RCCL_LOOPBACK", "1")
    monkeypatch.setenv("DDCMI_HALO_OVERLAP", "0")
    m = MartiniRank(s, np.arange(s.natoms), constraints=True)
    _declare_domains(m.lib)
    buf = ctypes.create_string_buffer(128)
    assert m.lib.ddcmi_comm_unique_id(buf) == 0
    m.comm_init(0, 1, buf.raw, (1, 1, 1))
    m.upload_local()
    m.eval_forces()
    m.group_temperatures()
    for block in range(3):
        eo, vo, rko, _ = o.step_npt(5, T, P0, beta, tau, molecular=True)
        m.step(5)
        o.group_temperature()
        m.group_temperatures()
        e, vir, rk, _ = m.energies()
        assert np.abs(m.barostat_pressure() - o.pmol).max() < 1e-8 * np.abs(o.pmol).max(), block
        assert np.abs(m.box() - o.box).max() < 1e-10 * o.box.max(), block
        assert abs(e["total"] - eo["total"]) < TOL * abs(eo["total"]), block
        assert abs(rk - rko) < TOL * rko
    sweeps, bad = m.constraint_stats()
    assert bad == 0 and sweeps > 1
    m.close()


def test_loopback_run_repeats_bit_for_bit(monkeypatch):
    """the RCCL path twice: the order in which atomics fill the send lists does not reach the neighbour lists (in-cell order by
    gid), so energies, positions and forces of two runs are identical in every bit"""
    outs = []
    for _ in range(2):
        m = _loopback_rank(make_water_setup(12), monkeypatch)
        m.eval_forces()
        m.step(45)
        e, vir, rk, tion = m.energies()
        p = m.download_particles()
        order = np.argsort(p["gid"], kind="stable")
        outs.append((e["total"], rk, vir.copy(), np.stack(p["r"])[:, order], np.stack(p["f"])[:, order]))
        m.close()
    assert outs[0][0] == outs[1][0] and outs[0][1] == outs[1][1]
    assert np.array_equal(outs[0][2], outs[1][2]) and np.array_equal(outs[0][3], outs[1][3]) and np.array_equal(outs[0][4], outs[1][4])


def test_a_rebuild_that_fails_mid_run_ends_every_rank_without_a_host_wait(monkeypatch):
    """from the third rebuild on the ranks no longer waited for each other's verdict on a rebuild's local phase,
    and a rank that failed there left its peers inside the next halo kernel.  Every rebuild now carries an agreement that nobody
    waits for (one small all-reduce on the stream; the result is read in front of the next host wait).  Through the loopback: (a) a
    healthy run takes that path at every later rebuild and stays on the oracle; (b) this rank's own failure at its 4th rebuild comes
    back as its own error, after the collective; (c) a peer's failure -- its code injected into the 4th rebuild's all-reduce --
    surfaces as DDCMI_ECOMM at the next host wait (here: the read of the energies), naming the rebuild, and nothing hangs"""
    from ddcmd_amd.martini import DdcmiError
    s = make_water_setup(10)
    o = pyoracle.Oracle(s)
    o.forces()
    m = _loopback_rank(s, monkeypatch)
    m.eval_forces()
    m.step(100)                                     # five more rebuilds, each with an agreement in flight until the next count round
    eo, _, rko, _ = o.step(100)
    e, _, rk, _ = m.energies()
    assert m.list_stats()["rebuilds"] >= 6 and abs(e["total"] - eo["total"]) < TOL * abs(eo["total"]) and abs(rk - rko) < TOL * rko
    m.close()
    monkeypatch.setenv("DDCMI_DEBUG_HOOKS", "1")
    monkeypatch.setenv("DDCMI_DEBUG_FAIL_REBUILD", "0:4")
    m = _loopback_rank(s, monkeypatch)
    m.eval_forces()
    with pytest.raises(DdcmiError, match="injected failure"):
        m.step(100)
    assert m.list_stats()["rebuilds"] <= 4
    m.close()
    monkeypatch.delenv("DDCMI_DEBUG_FAIL_REBUILD")
    monkeypatch.setenv("DDCMI_DEBUG_PEER_FAILS_REBUILD", "4")
    m = _loopback_rank(s, monkeypatch)
    m.eval_forces()
    m.step(70)                                      # rebuilds 2, 3, 4 (at loops 20, 40, 60): nothing waits for the 4th one's agreement
    with pytest.raises(DdcmiError, match="another rank failed during the list rebuild at loop 60"):
        m.energies()
    m.close()


@pytest.mark.parametrize("vscale", [3.0, 0.02])
def test_loopback_rows_end_at_the_last_shell_that_can_matter(monkeypatch, vscale):
    """the same through the RCCL transport (every image bead is a received halo bead here, and the fused step -- pair kernel with the
    integrator's pass, two position buffers, halo messages packed in the reduction launch -- is the path a production rank takes):
    bit for bit the full walk"""
    s = make_water_setup(12)
    s.vx, s.vy, s.vz = (np.asarray(v) * vscale for v in (s.vx, s.vy, s.vz))
    monkeypatch.delenv("DDCMI_NO_SHELL_SKIP", raising=False)
    a = _loopback_rank(s, monkeypatch)
    monkeypatch.setenv("DDCMI_NO_SHELL_SKIP", "1")
    b = _loopback_rank(s, monkeypatch)
    monkeypatch.delenv("DDCMI_NO_SHELL_SKIP", raising=False)
    o = pyoracle.Oracle(s)
    o.forces()
    a.eval_forces(); b.eval_forces()
    for block, n in enumerate((17, 20, 8)):
        a.step(n); b.step(n)
        eo, vo, rko, _ = o.step(n)
        pa, pb = a.download_particles(), b.download_particles()
        assert np.array_equal(pa["gid"], pb["gid"])
        for k in ("r", "v", "f"):
            for c in range(3):
                assert np.array_equal(pa[k][c], pb[k][c]), (block, k, c)
        ea, _, rka, _ = a.energies()
        eb, _, rkb, _ = b.energies()
        assert ea["total"] == eb["total"] and rka == rkb
        assert abs(ea["total"] - eo["total"]) < TOL * abs(eo["total"]) and abs(rka - rko) < TOL * max(rko, 1e-300)
    a.close(); b.close()


def test_halo_staged_from_the_receive_buffer_equals_the_update_launch(monkeypatch):
    """round 5: a rank whose halo holds received beads only stages them in k_nonbond straight from the exchange's receive buffer
    (through halo_src) -- no k_halo_update launch between the exchange and the pair kernel.  Same run with DDCMI_NO_DIRECT_HALO=1
    (the update launch, the measured halo displacement): every bit of positions, velocities, forces and energies agrees, across
    rebuilds, single force evaluations, batches of fused steps and print steps"""
    s = make_water_setup(12)
    monkeypatch.delenv("DDCMI_NO_DIRECT_HALO", raising=False)
    a = _loopback_rank(s, monkeypatch)
    monkeypatch.setenv("DDCMI_NO_DIRECT_HALO", "1")
    b = _loopback_rank(s, monkeypatch)
    monkeypatch.delenv("DDCMI_NO_DIRECT_HALO", raising=False)
    o = pyoracle.Oracle(s)
    o.forces()
    a.eval_forces(); b.eval_forces()
    for block, n in enumerate((1, 1, 17, 23, 5)):
        a.step(n); b.step(n)
        eo, vo, rko, _ = o.step(n)
        if block == 3:
            ea1, _ = a.eval_forces(); eb1, _ = b.eval_forces()      # a force evaluation between batches: no exchange, no move
            assert ea1["total"] == eb1["total"]
        pa, pb = a.download_particles(), b.download_particles()
        assert np.array_equal(pa["gid"], pb["gid"])
        for k in ("r", "v", "f"):
            for c in range(3):
                assert np.array_equal(pa[k][c], pb[k][c]), (block, k, c)
        ea, va, rka, _ = a.energies()
        eb, vb, rkb, _ = b.energies()
        assert ea["total"] == eb["total"] and rka == rkb and np.array_equal(va, vb)
        assert abs(ea["total"] - eo["total"]) < TOL * abs(eo["total"]) and abs(rka - rko) < TOL * rko
    a.close(); b.close()


def test_bonded_partners_out_of_the_receive_buffer_equal_the_update_launch(monkeypatch):
    """round 6: a decomposed rank WITH bonded terms stages its halo from the receive buffer too -- the bonded kernels take a received
    partner's position out of the exchange's receive buffer (k_bonded_gather's bead()) -- and so runs lean (bonded kernels, pair kernel with
    the integrator inside, pack, exchange: nothing else between rebuilds).  The lipid deck tiled 2x2x1 through the RCCL loopback (lipids
    across every face, Berendsen group) against the same run with DDCMI_NO_DIRECT_HALO=1 (halo update launch, reduction launch per step):
    positions, velocities and forces agree to the last bit across rebuilds, single evaluations and batches; energies by kind agree with the
    oracle's run of the deck."""
    from ddcmd_amd.deck import load_deck
    from ddcmd_amd.synth import replicate_setup
    deck = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "lipid_deck")
    s0 = load_deck(os.path.join(deck, "object_nvt.data"), restart_file=os.path.join(deck, "relaxed", "restart"))
    reps = (2, 2, 1)
    ncopy = 4
    s = replicate_setup(s0, reps)
    monkeypatch.delenv("DDCMI_NO_DIRECT_HALO", raising=False)
    a = _loopback_rank(s, monkeypatch)
    monkeypatch.setenv("DDCMI_NO_DIRECT_HALO", "1")
    b = _loopback_rank(s, monkeypatch)
    monkeypatch.delenv("DDCMI_NO_DIRECT_HALO", raising=False)
    o = pyoracle.Oracle(s0)
    o.forces(); o.group_temperature()
    a.eval_forces(); b.eval_forces()
    a.group_temperatures(); b.group_temperatures()
    for block, n in enumerate((1, 9, 10, 10, 7)):          # (the deck rebuilds every 10 steps; its Berendsen group reads the temperature published here)
        a.step(n); b.step(n)
        eo, vo, rko, _ = o.step(n)
        if block == 2:
            ea1, _ = a.eval_forces(); eb1, _ = b.eval_forces()      # a force evaluation between batches: no exchange, no move
            assert ea1["total"] == eb1["total"]
        pa, pb = a.download_particles(), b.download_particles()
        assert np.array_equal(pa["gid"], pb["gid"])
        for k in ("r", "v", "f"):
            for c in range(3):
                assert np.array_equal(pa[k][c], pb[k][c]), (block, k, c)
        ea, va, rka, _ = a.energies()
        eb, vb, rkb, _ = b.energies()
        for k in ("lj", "ele", "bond", "angle", "tors", "impr"):
            assert abs(ea[k] - eb[k]) <= 1e-12 * max(abs(eb[k]), 1.0), (block, k)          # (the lean steps' sums are formed in batches: the same additions)
            assert abs(ea[k] - ncopy * eo[k]) < TOL * ncopy * max(abs(eo[k]), abs(eo["total"]) * 1e-3), (block, k)
        assert abs(rka - rkb) <= 1e-12 * rkb and abs(rka - ncopy * rko) < TOL * ncopy * rko
        o.group_temperature()
        Ta, Tb = a.group_temperatures(), b.group_temperatures()
        assert Ta[0] == Tb[0]
    a.close(); b.close()


def test_explicit_rebuild_then_forces_through_the_loopback(monkeypatch):
    """constructList called by hand (ddcmi_build_list, the neighbour hook of ddcUpdateAll.c:136-139) followed by a force evaluation
    on a decomposed rank: the rebuild itself has placed the halo beads -- the evaluation must not refresh them from the per-step
    receive buffer, which no exchange has filled yet (round 4 dropped the rebuild's 5 -> 3 copy of the received records)"""
    s = make_water_setup(10)
    o = pyoracle.Oracle(s)
    e0, v0 = o.forces()
    m = _loopback_rank(s, monkeypatch)
    e, _ = m.eval_forces()
    for _ in range(2):
        m.build_list()
        e, _ = m.eval_forces()
        assert abs(e["lj"] - e0["lj"]) < 1e-10 * abs(e0["lj"])
    e2, _ = m.eval_forces()                      # twice in a row, no step and no exchange in between: the halo stays where the rebuild put it
    assert e2["lj"] == e["lj"]
    m.step(3)
    m.build_list()
    e, _ = m.eval_forces()
    eo, _, _, _ = o.step(3)
    assert abs(e["total"] - eo["total"]) < TOL * abs(eo["total"])
    m.step(2)
    e, _ = m.eval_forces(); e2, _ = m.eval_forces()      # and after steps (the receive buffer holds the last exchange)
    eo, _ = o.step(2)[0], None
    assert e2["total"] == e["total"] and abs(e["total"] - eo["total"]) < TOL * abs(eo["total"])
    m.close()
