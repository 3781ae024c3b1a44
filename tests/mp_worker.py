"""One rank of a multi-process decomposed run (started by tests/test_gpu_multiproc.py as a fresh child
process): rendezvous from RANK / WORLD_SIZE / MASTER_ADDR, libddcmi context on device 0, decomposition over
the transport DDCMI_TRANSPORT names, a few NGLF steps, then every rank writes its own beads (by gid) and
its partial sums to <outdir>/rank<r>.npz.  Never imports the oracle: the parent process is the checker."""
import ctypes
import os
import sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    workload, gridtxt, outdir, nsteps, block = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4]), int(sys.argv[5])
    grid = tuple(int(x) for x in gridtxt.split("x"))
    import ddcmd_amd
    from ddcmd_amd.martini import MartiniRank, Rendezvous, domain_of, _declare_domains
    rdzv = Rendezvous.from_env(timeout=float(os.environ.get("DDCMI_TEST_RDZV_TIMEOUT", "120")))
    rank, world = rdzv.rank, rdzv.world
    cons = False
    if workload in ("water", "water_fault", "water_preflight"):
        s = ddcmd_amd.make_water_setup(12)
    elif workload in ("water_drift", "lipid_drift"):
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from drift_setup import drifting_setup      # (the parent runs the same system on one domain)
        s = drifting_setup(workload, grid)
    else:
        from ddcmd_amd.deck import load_deck, units_convert
        deck = os.path.join(ROOT, "tests", "golden", "lipid_deck")
        extra = None
        if workload == "lipid_npt":          # the full nglfconstraint step: constraint lists + barostat on the molecular pressure
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            extra = os.environ["DDCMI_TEST_CONSTRAINT_X"]
            cons = True
        s = load_deck(os.path.join(deck, "object_nvt.data"), restart_file=os.path.join(deck, "relaxed", "restart"), extra_objects=extra)
        if cons:
            s.npt_T, s.npt_P0 = units_convert(310.0, "K"), units_convert(1.0, "bar")
            s.npt_beta, s.npt_tau = units_convert(3.0e-4, "1/bar") * 20.0, units_convert(1.0, "ps")
    owner = domain_of(s, grid)
    if os.environ.get("DDCMI_TEST_DETUNE_RANK") == str(rank):
        s.deltaR = s.deltaR * 1.01      # this rank "read another deck": the preflight's parameter check must name it on every rank
    if workload == "water_fault":
        # one bead of the LAST rank is not a number: that rank's list build refuses the state, and every rank must
        # come back from the rebuild with an error instead of waiting for the one that has gone
        s.rx = np.array(s.rx, dtype=float)
        s.rx[np.flatnonzero(owner == world - 1)[0]] = float("nan")
    m = MartiniRank(s, np.flatnonzero(owner == rank), device=0, constraints=cons)
    _declare_domains(m.lib)
    if os.environ.get("DDCMI_TRANSPORT", "host") == "host":
        m.comm_init_host(rdzv, grid)
    else:
        buf = ctypes.create_string_buffer(128)
        if rank == 0:
            assert m.lib.ddcmi_comm_unique_id(buf) == 0
        m.comm_init(rank, world, rdzv.bcast(buf.raw, 0), grid)
    if workload == "water_preflight":
        # the communicator's preflight (ddcmi_comm_preflight) between real processes; fault hooks come in through the environment
        from ddcmd_amd.martini import DdcmiError
        try:
            rep = m.preflight(timeout=float(os.environ.get("DDCMI_TEST_PREFLIGHT_TIMEOUT", "20")))
        except DdcmiError as ex:
            sys.stderr.write("PREFLIGHT rank %d: %s\n" % (rank, ex))
            sys.exit(4)
        sys.stderr.write("PREFLIGHT-OK rank %d: %r\n" % (rank, rep))
        sys.exit(0)
    m.upload_local()
    n0 = m.n
    rec = {}
    if workload == "water_fault":
        from ddcmd_amd.martini import DdcmiError
        try:
            m.eval_forces()
        except DdcmiError as ex:
            sys.stderr.write("FAULT rank %d: %s\n" % (rank, ex))
            sys.exit(3)
        sys.exit(0)
    e, vir = m.eval_forces()
    p = m.download_particles()
    rec.update(gid0=p["gid"], f0=np.stack(p["f"]), e0=np.array([e[k] for k in ("lj", "ele", "bond", "angle", "tors", "impr", "total")]), vir0=vir)
    Tg = []
    if any(int(t) == 1 for t in np.asarray(s.group_type).ravel()):
        Tg.append(m.group_temperatures().copy())
    done = 0
    traj = []
    baro = []
    while done < nsteps:
        k = min(block, nsteps - done)
        m.step(k)
        done += k
        e, vir, rk, tion = m.energies()
        tot = m.allreduce([e["total"], rk] + list(vir))              # energyInfo.c allreduce() over the transport
        traj.append(tot)
        if cons:
            baro.append(np.concatenate((m.barostat_pressure(), m.box())))
        if Tg:
            Tg.append(m.group_temperatures().copy())
    p = m.download_particles()
    rec.update(gid=p["gid"], r=np.stack(p["r"]), v=np.stack(p["v"]), f=np.stack(p["f"]), traj=np.array(traj), Tg=np.array(Tg), baro=np.array(baro),
               nloc=np.array([n0, int(m.lib.ddcmi_nlocal(m.ctx))]), rebuilds=np.array([m.list_stats()["rebuilds"]]))
    np.savez(os.path.join(outdir, "rank%d.npz" % rank), **rec)
    m.close()
    rdzv.barrier()
    rdzv.close()


if __name__ == "__main__":
    main()
