#!/usr/bin/env python3
"""bench.py -- atom-steps/s of the Martini MD inner loop on synthetic water.

    python bench.py --gpus N --steps K --warmup W [--lattice N] [--no-cpu]

One "step" is one NGLF velocity-Verlet step of the whole box (half kick, drift,
image refresh, nonbonded + bonded forces with energy and virial, half kick +
kinetic terms; neighbour-list rebuild every 20 steps inside the timed region).
Workload at N=1: BASELINE.json's headline config, the 4.0M-bead Martini water
box (FCC n=100 lattice, rcut 12 A, skin 4 A, dt 20 fs), state resident in HBM.
Prints ONE JSON line (rank 0).  `roofline` prices the nonbonded kernel with the
ALGORITHMIC bytes of SURVEY 8(d): (36 + 24 + 4*L) B per atom-step, L = stored
full-list entries per atom, over the HIP-event time of that kernel measured on
the library's own stream.  `cpu_baseline` times the CPU oracle (a port of the
reference's serial per-rank path; the reference itself cannot be built) on one
host core on a bounded sample of the same workload.
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
DT_FS = 20.0


def cpu_baseline(n_lattice, seconds_budget=20.0):
    """Oracle (port of bioMartini.c/pairlist.c/nglf.c) on one core, bounded sample."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pyoracle
    import ddcmd_amd
    import tempfile
    native = os.path.join(tempfile.gettempdir(), "libddc_oracle_native_%d.so" % os.getpid())
    try:
        pyoracle.build(native=True, out=native)
        libpath = native
    except Exception:
        libpath = None
    s = ddcmd_amd.make_water_setup(n_lattice)
    o = pyoracle.Oracle(s, libpath)
    o.forces()
    t0 = time.time()
    o.step(1)
    per = max(time.time() - t0, 1e-4)
    steps = int(max(20, min(200, seconds_budget / per)))
    steps = (steps // 20) * 20           # whole rebuild periods
    o.step(20)                           # warm-up incl. one rebuild
    t0 = time.time()
    o.step(steps)
    el = time.time() - t0
    try:
        os.remove(native)
    except OSError:
        pass
    return {"value": s.natoms * steps / el, "unit": "atom-steps/s", "cores": 1, "kind": "port",
            "host_cores": os.cpu_count(),
            "sample": "%d-bead Martini water (n=%d lattice), %d NGLF steps incl. %d list rebuilds, gcc -O3 -march=native, 1 thread"
                      % (s.natoms, n_lattice, steps, steps // 20)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--lattice", dest="n", type=int, default=100, help="FCC lattice edge: 4*n^3 beads (100 -> 4.0M, 64 -> 1.05M, 25 -> 62.5k)")
    ap.add_argument("--cpu-n", type=int, default=25, help="lattice edge of the CPU-baseline sample (25 -> 62.5k beads)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--workload", choices=("water", "lipid"), default="water",
                    help="water: the headline Martini water box; lipid: tests/golden/lipid_deck (DPPC-style bilayer patch in water, "
                         "all bonded term kinds, charges, Berendsen) tiled --reps times")
    ap.add_argument("--rccl-loopback", action="store_true",
                    help="N=1 only: reach the periodic images through a 1-rank RCCL communicator (the multi-GPU transport on one GPU)")
    ap.add_argument("--reps", default="12,12,6", help="lipid workload: copies of the 2363-bead deck along x,y,z (12,12,6 -> 2.04M beads)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if os.environ.get("DDCMI_BENCH_SINGLE_DEVICE"):      # debugging aid: all ranks on device 0 (RCCL normally refuses)
        local_rank = 0
    dist = None
    torch = None
    if world > 1:
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        # The data path (halo exchange, migration, count all-gathers, energy all-reduce) runs over libddcmi's own RCCL
        # communicator.  The control plane of this script (broadcast of the RCCL unique id, barriers, max of the
        # timings) goes over gloo by default, so that the process holds exactly one RCCL communicator -- the
        # configuration validated on one GPU through the loopback mode; DDCMI_BENCH_CONTROL=nccl uses torch's
        # RCCL backend for it instead.
        control = os.environ.get("DDCMI_BENCH_CONTROL", "gloo")
        if control == "gloo":
            dist.init_process_group(backend="gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        ctl_dev = "cpu" if control == "gloo" else "cuda"

    import numpy as np
    import ddcmd_amd
    from ddcmd_amd.martini import MartiniHIP, MartiniRank, domain_of
    if args.workload == "water":
        s = ddcmd_amd.make_water_setup(args.n)
        wname = "martini_water_%dk_beads" % (s.natoms // 1000)
    else:
        from ddcmd_amd.deck import load_deck
        from ddcmd_amd.synth import replicate_setup
        reps = tuple(int(x) for x in args.reps.split(","))
        deck = os.path.join(ROOT, "tests", "golden", "lipid_deck")
        # 310 K restart relaxed by tests/golden/make_lipid_relaxed.py, Berendsen group (Teq 310 K, tau 1 ps)
        s = replicate_setup(load_deck(os.path.join(deck, "object_nvt.data"), restart_file=os.path.join(deck, "relaxed", "restart")), reps)
        wname = "martini_lipid_bilayer_%dk_beads" % (s.natoms // 1000)
    dt_fs = float(ddcmd_amd.units_convert(s.dt, None, "fs"))
    grid = {1: (1, 1, 1), 2: (2, 1, 1), 4: (2, 2, 1), 8: (2, 2, 2)}.get(world)
    if grid is None:
        raise SystemExit("bench.py supports 1, 2, 4 or 8 GPUs (2x1x1, 2x2x1, 2x2x2 bricks)")
    if world == 1 and args.rccl_loopback:
        os.environ["DDCMI_RCCL_LOOPBACK"] = "1"
        m = MartiniRank(s, np.arange(s.natoms), device=local_rank)
        buf = ctypes.create_string_buffer(128)
        assert m.lib.ddcmi_comm_unique_id(buf) == 0
        m.comm_init(0, 1, buf.raw, (1, 1, 1))
        m.upload_local()
    elif world == 1:
        m = MartiniHIP(s, device=local_rank)
    else:
        # spatial decomposition: this rank uploads the beads of its brick; halo exchange
        # and migration run inside libddcmi over RCCL point-to-point (include/ddcmi.h)
        owner = domain_of(s, grid)
        m = MartiniRank(s, np.flatnonzero(owner == rank), device=local_rank)
        uid = torch.zeros(128, dtype=torch.uint8, device=ctl_dev)
        if rank == 0:
            buf = ctypes.create_string_buffer(128)
            assert m.lib.ddcmi_comm_unique_id(buf) == 0
            uid = torch.frombuffer(bytearray(buf.raw), dtype=torch.uint8).to(ctl_dev)
        dist.broadcast(uid, src=0)
        m.comm_init(rank, world, bytes(uid.cpu().numpy().tobytes()), grid)
        m.upload_local()
    nlocal0 = m.n
    m.eval_forces()                       # firstEnergyCall (masters.c:579)
    # RCCL prints a version banner through C stdio at communicator creation; flush it now so that the JSON
    # line below is the last line of stdout
    ctypes.CDLL(None).fflush(None)
    thermostat = any(int(t) == 1 for t in np.asarray(s.group_type).ravel())
    if thermostat:
        m.group_temperatures()            # the temperature Berendsen scales with (published by eval_energyInfo in the reference)
    m.step(args.warmup)
    if thermostat:
        m.group_temperatures()
    m.sync()

    def barrier():
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()
        m.sync()

    m.timing(True)
    barrier()
    reb0 = m.list_stats()["rebuilds"]
    t0 = time.perf_counter()
    m.step(args.steps)
    barrier()
    el = time.perf_counter() - t0
    launches, kernel_ms = m.timing_read()
    m.timing(False)
    e, vir, rk, tion = m.energies()
    st = m.list_stats()
    nlocal = int(m.lib.ddcmi_nlocal(m.ctx))
    epot, ekin = e["total"], rk
    if dist is not None:
        t = torch.tensor([el], dtype=torch.float64, device=ctl_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())
        tot = m.allreduce([epot, ekin, float(nlocal)])      # energyInfo.c allreduce()
        epot, ekin = float(tot[0]), float(tot[1])
        assert int(round(tot[2])) == s.natoms, "beads lost in migration"

    value = s.natoms * args.steps / el     # whole box, whole job
    L = st["entries"] / float(max(nlocal, 1))
    bytes_per_atom = 36.0 + 24.0 + 4.0 * L
    t_kernel = kernel_ms * 1e-3 / max(1, launches)
    achieved = bytes_per_atom * nlocal / t_kernel / 1e9
    # HBM bytes of the nonbonded kernel from the PMC passes committed under profiles/ (same workload only)
    traffic = None
    try:
        pmc_file = {100: "r01_traffic.json", 64: "r01_1M_traffic.json"}.get(args.n)
        if world == 1 and pmc_file and args.workload == "water":
            traffic = json.load(open(os.path.join(ROOT, "profiles", pmc_file)))["traffic_bytes_per_launch"]
    except Exception:
        traffic = None
    out = {
        "metric": "atom_steps_per_sec", "value": value, "unit": "atom-steps/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": el * 1e3 / args.steps, "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "ns_per_day": (args.steps / el) * dt_fs * 1e-6 * 86400.0,
        "config": {"workload": wname, "beads_total": s.natoms, "beads_rank0": nlocal,
                   "lattice": "fcc" if args.workload == "water" else "deck tiled %s" % args.reps, "lattice_n": args.n if args.workload == "water" else None,
                   "rcut_A": float(ddcmd_amd.units_convert(s.rmax, None, "Angstrom")), "skin_A": float(ddcmd_amd.units_convert(s.deltaR, None, "Angstrom")),
                   "dt_fs": dt_fs, "list_rebuild_every": int(s.updateRate),
                   "bonded_terms": {k: int(m.terms[k].size) for k in ("bond_kb", "angle_k", "tors_k")},
                   "energy_virial_every_step": True,
                   "parallelism": ("spatial decomposition %dx%dx%d, RCCL p2p halo (control plane: %s)" % (grid + (control,))) if world > 1 else ("single GPU, images through RCCL loopback" if args.rccl_loopback else "single GPU"),
                   "list_entries_per_atom": L, "image_or_halo_beads_rank0": st["images"], "rebuilds_in_timed_region": st["rebuilds"] - reb0},
        "roofline": {"bound": "hbm", "kernel": "k_nonbond", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                     "algorithmic_bytes_per_atom_step": bytes_per_atom, "kernel_ms_avg": t_kernel * 1e3, "launches": launches,
                     "note": "rank 0's kernel on its own beads" if world > 1 else "whole box"},
        "check": {"epot": epot, "ekin": ekin},
    }
    if rank == 0 and world == 1 and not args.no_cpu:
        out["cpu_baseline"] = cpu_baseline(args.cpu_n)      # N=1 only (the contract): a bounded sample on one host core
    m.close()
    if dist is not None:
        dist.destroy_process_group()
    if rank == 0:
        ctypes.CDLL(None).fflush(None)                      # anything the libraries left in C stdio goes out first
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
