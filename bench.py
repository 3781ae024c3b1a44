#!/usr/bin/env python3
"""bench.py -- atom-steps/s of the Martini MD inner loop on synthetic water.

    python bench.py --gpus N --steps K --warmup W [--lattice N] [--no-cpu]

One "step" is one NGLF velocity-Verlet step of the whole box (half kick, drift,
image refresh, nonbonded + bonded forces with energy and virial, half kick +
kinetic terms; neighbour-list rebuild every 20 steps inside the timed region).
Workload at N=1: BASELINE.json's headline config, the 4M-bead Martini water
box (FCC n=102 lattice = 4 244 832 beads >= SURVEY 8d's 4 096 000, even under a 2x2x2 split; rcut 12 A, skin 4 A, dt 20 fs),
state resident in HBM;
200 untimed steps in front of the warm-up melt the lattice start (50 K) into the
liquid (~307 K), so the timed steps see production list lengths.
The timed region is max(--steps, 60) steps rounded up to windows of 20, each window between a
barrier + stream drain on both sides; `ms_per_step` and `value` come from the MEDIAN window
(`window_ms` lists them all).  At N=1 the line also carries `also`: the other single-GPU configs of
BASELINE.json (1 M-bead water, the 2 M-bead lipid bilayer) and one rank's brick of the 8-GPU runs
through the RCCL loopback, each timed the same way -- in ONE fresh child process started when the
headline has finished its GPU work (a fault in a side workload never costs the headline's line),
followed by the rocprofv3 --pmc child runs when nothing else of the launch uses the GPU any more.
Prints ONE JSON line (rank 0).  `roofline` prices the nonbonded kernel with the
ALGORITHMIC bytes of SURVEY 8(d): (36 + 24 + 4*L) B per atom-step, L = stored
full-list entries per atom, over the HIP-event time of that kernel measured on
the library's own stream; `roofline.traffic` = that kernel's HBM bytes per launch MEASURED IN THIS RUN (N=1 headline: two short
child runs under `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE`, see live_traffic; the committed profiles/*traffic.json of the
same device sources only if the profiler is not there).  `cpu_baseline` times the CPU oracle (a port of the
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

HEADLINE_N = 102               # FCC edge of the headline box: 4 n^3 = 4 244 832 beads (n = 100 was 2.3 % under SURVEY 8d's 4 096 000)
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP64_VALU_PEAK_TFLOPS = 78.6   # vector FP64: half of the guide's 157.3 TFLOP/s FP32 vector peak (AMD's MI355X figure)
WINDOW = 20                    # timed steps come in windows of 20: one list rebuild each for water (two for the lipid deck's 10-step period)
MIN_TIMED_STEPS = 60           # at least three windows whatever --steps says (a 20-step region is one rebuild's luck)


def cpu_baseline(n_lattice, seconds_budget=12.0, one_million=True):
    """Oracle (port of bioMartini.c/pairlist.c/nglf.c) on one core, bounded samples: the synthetic water
    box at n_lattice (26: 70 304 beads, BASELINE configs[1]), -- SURVEY 8(d) -- the reference's own 6173-bead examples/waterbox
    and (one_million) one rebuild period of the 1.05 M-bead box of configs[2]."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pyoracle
    import ddcmd_amd
    import tempfile
    import numpy as np
    native = os.path.join(tempfile.gettempdir(), "libddc_oracle_native_%d.so" % os.getpid())
    try:
        pyoracle.build(native=True, out=native)
        libpath = native
    except Exception:
        libpath = None

    def timed(s, budget, warm=True):
        period = max(int(s.updateRate), 1)
        o = pyoracle.Oracle(s, libpath)
        o.forces()
        t0 = time.time()
        o.step(1)
        per = max(time.time() - t0, 1e-4)
        steps = int(max(period, min(10 * period, budget / per)))
        steps = (steps // period) * period   # whole rebuild periods
        if warm:
            o.step(period)                   # warm-up incl. one rebuild
        else:
            o.step(period - 1)               # (the big sample: up to the rebuild, so that the timed period holds exactly one)
        t0 = time.time()
        o.step(steps)
        return steps, time.time() - t0

    s = ddcmd_amd.make_water_setup(n_lattice)
    steps, el = timed(s, seconds_budget)
    out = {"value": s.natoms * steps / el, "unit": "atom-steps/s", "cores": 1, "kind": "port",
           "host_cores": os.cpu_count(),
           "sample": "%d-bead Martini water (n=%d lattice), %d NGLF steps incl. %d list rebuilds, gcc -O3 -march=native, 1 thread"
                     % (s.natoms, n_lattice, steps, steps // max(int(s.updateRate), 1))}
    try:
        from ddcmd_amd.deck import setup_from_dict
        w = setup_from_dict(dict(np.load(os.path.join(ROOT, "tests", "golden", "waterbox.npz"))))
        steps, el = timed(w, 4.0)
        out["also"] = [{"value": w.natoms * steps / el, "unit": "atom-steps/s", "cores": 1,
                        "sample": "the reference's examples/waterbox (%d beads, rcut 11 A, skin 4 A), %d NGLF steps" % (w.natoms, steps)}]
    except Exception as ex:      # the fixture is optional for the figure above
        out["also"] = [{"error": str(ex)}]
    if one_million:
        try:
            big = ddcmd_amd.make_water_setup(64)
            t0 = time.time()
            steps, el = timed(big, 0.0, warm=False)      # exactly one rebuild period
            out["also"].append({"value": big.natoms * steps / el, "unit": "atom-steps/s", "cores": 1,
                                "sample": "%d-bead Martini water (n=64 lattice: BASELINE configs[2]), %d NGLF steps incl. 1 list rebuild, %.0f s of CPU work in all"
                                          % (big.natoms, steps, time.time() - t0)})
        except Exception as ex:
            out["also"].append({"error": str(ex)})
    try:
        os.remove(native)
    except OSError:
        pass
    return out


def nonbond_is_fused(kernel_name):
    """k_nonbond<HAS_Q, PACKED, SHBIT, threads, waves, CH, ZOFF, FUSE, LVL>: is FUSE true?"""
    k = kernel_name.split("(")[0]
    if "<" not in k or ">" not in k:
        return False
    a = [x.strip() for x in k[k.index("<") + 1:k.rindex(">")].split(",")]
    return len(a) >= 2 and a[-2] == "true"


def live_traffic(extra_args, fused, timeout_s=240.0):
    """HBM bytes per launch of the pair kernel, MEASURED IN THIS RUN and IN THE STATE OF THE TIMED RUN (the passes used to
    sample a lattice start within six steps of a rebuild, where the shell-limited walk reads the fewest list bytes): two child runs of
    this very bench with the SAME equilibration and warm-up as the timed run and MIN_TIMED_STEPS timed steps, under
    `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` -- separate passes, as MI355X_MICROARCH.md prescribes; counters are KiB;
    FETCH_SIZE reports half the bytes of wide coalesced streaming reads on gfx950 and is doubled.  Only the launches of the child's
    timed windows are averaged (the last MIN_TIMED_STEPS steps' launches by dispatch order: whole rebuild periods, every age of the
    list).  Fresh child processes (never an exec of this one); None if the profiler is not there or a pass fails.
    fused: price the launches whose epilogue is the integrator's pass (k_nonbond<..., FUSE, LVL>: the template's last argument but one)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    prof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(prof):
        return None
    vals = {}
    nlaunch = 0
    with tempfile.TemporaryDirectory(prefix="ddcmi_pmc_") as d:
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            out = os.path.join(d, ctr)
            cmd = [prof, "--pmc", ctr, "--output-format", "csv", "-d", out, "-o", "p", "--", sys.executable, os.path.abspath(__file__),
                   "--no-cpu", "--no-also", "--no-pmc", "--steps", str(MIN_TIMED_STEPS), "--warmup", "20"] + list(extra_args)
            env = dict(os.environ)
            env["TMPDIR"] = "/tmp"
            try:
                r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=timeout_s)
            except Exception:
                return None
            if r.returncode != 0:
                return None
            rows = []
            for fn in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(fn)):
                    if "k_nonbond" in row["Kernel_Name"] and row["Counter_Name"] == ctr:
                        rows.append((int(row["Dispatch_Id"]), nonbond_is_fused(row["Kernel_Name"]), float(row["Counter_Value"])))
            rows.sort()
            # the timed windows are the child's last MIN_TIMED_STEPS force evaluations (one pair launch each on a single domain)
            rows = rows[-MIN_TIMED_STEPS:]
            acc = [v for _, f, v in rows if f == bool(fused)]
            if not acc:
                return None
            vals[ctr] = sum(acc) / len(acc)
            nlaunch = len(acc)
    return {"bytes_per_launch": (2.0 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024.0, "FETCH_SIZE_KiB": vals["FETCH_SIZE"], "WRITE_SIZE_KiB": vals["WRITE_SIZE"],
            "source": "live: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over child runs of this bench in the timed run's own state "
                      "(same equilibration and warm-up; the %d %s launches of the child's last %d steps: whole rebuild periods); FETCH_SIZE doubled (gfx950), KiB counters"
                      % (nlaunch, "fused" if fused else "plain", MIN_TIMED_STEPS)}


def runtime_libs():
    """which HIP / RCCL runtime this process has mapped (the library is built and validated against /opt/rocm)"""
    libs = set()
    try:
        for line in open("/proc/self/maps"):
            path = line.split()[-1]
            base = os.path.basename(path)
            if base.startswith(("libamdhip64", "librccl", "libhsa-runtime64", "libtorch", "libc10")):
                libs.add(path)
    except OSError:
        pass
    return sorted(libs)


def kernel_source_id():
    """short hash of the device sources: a PMC file under profiles/ is quoted only for the kernels it measured"""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "ddcmd_amd", "csrc", "hip")
    for f in sorted(os.listdir(d)):      # ddcmi.hip and the .inl parts it includes, the shared header, scan.hip, bonded.hip
        if f.endswith((".hip", ".inl", ".h")):
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def build_setup(workload, n, reps, types=0, cadence="reference"):
    import ddcmd_amd
    if workload == "water":
        s = ddcmd_amd.make_water_setup(n, density_scale=float(os.environ.get("DDCMI_BENCH_DENSITY_SCALE", "1.0")))      # (the variable: kernel-tuning experiments only)
        return s, "martini_water_%dk_beads" % (s.natoms // 1000), "fcc", n
    from ddcmd_amd.deck import load_deck
    from ddcmd_amd.synth import replicate_setup
    r3 = tuple(int(x) for x in reps.split(","))
    deck = os.path.join(ROOT, "tests", "golden", "lipid_deck")
    # 310 K restart relaxed by tests/golden/make_lipid_relaxed.py, Berendsen group (Teq 310 K, tau 1 ps)
    s = replicate_setup(load_deck(os.path.join(deck, "object_nvt.data"), restart_file=os.path.join(deck, "relaxed", "restart")), r3)
    if os.environ.get("DDCMI_BENCH_NOCHARGE"):      # (kernel-tuning experiments only: what the charges cost the pair kernel)
        import numpy as _np
        s.charge = _np.zeros_like(_np.asarray(s.charge, dtype=float))
    if cadence == "reference":
        # the cadence of the reference's shipped decks (examples/waterbox/object.data:10,35; examples/object/object.data:13,41): dt = 20 fs, list
        # rebuild every 20 steps.  The deck itself says 10 fs / 10 (a leftover of its relaxation); the 2.04 M-bead box holds 310 K, its bond
        # lengths and its aged-list forces over 20 000 steps at 20 / 20 (tools/lipid_soak_r05.py, profiles/r05_lipid_soak.txt)
        import ddcmd_amd
        s.dt = ddcmd_amd.units_convert(20.0, "fs")
        s.updateRate = 20
    if types:
        # the same physics under `types` LJ types (every type split into copies of itself, each bead's copy drawn at random): the size of
        # the pair kernel's class table is what changes (bioMartini.c:868-950 builds nspecies^2 entries; a real Martini deck has ~40 types)
        from ddcmd_amd.synth import relabel_types
        s = relabel_types(s, types)
        return s, "martini_lipid_bilayer_%dk_beads_%d_types" % (s.natoms // 1000, types), "deck tiled %s, %d LJ types" % (reps, types), None
    return s, "martini_lipid_bilayer_%dk_beads" % (s.natoms // 1000), "deck tiled %s" % reps, None


def run_config(workload, n, reps, steps, warmup, equil, world, rank, local_rank, rdzv, transport, loopback, types=0, cadence="reference"):
    """one workload on this launch's ranks: equilibration, warm-up, then the timed windows.  Returns the pieces of the JSON line."""
    import numpy as np
    import ddcmd_amd
    from ddcmd_amd.martini import MartiniHIP, MartiniRank, domain_of
    s, wname, lattice, lattice_n = build_setup(workload, n, reps, types, cadence)
    dt_fs = float(ddcmd_amd.units_convert(s.dt, None, "fs"))
    grid = {1: (1, 1, 1), 2: (2, 1, 1), 4: (2, 2, 1), 8: (2, 2, 2)}.get(world)
    preflight = None
    if grid is None:
        raise SystemExit("bench.py supports 1, 2, 4 or 8 GPUs (2x1x1, 2x2x1, 2x2x2 bricks)")
    if world == 1 and loopback:
        os.environ["DDCMI_RCCL_LOOPBACK"] = "1"
        m = MartiniRank(s, np.arange(s.natoms), device=local_rank)
        buf = ctypes.create_string_buffer(128)
        assert m.lib.ddcmi_comm_unique_id(buf) == 0
        m.comm_init(0, 1, buf.raw, (1, 1, 1))
        preflight = m.preflight(timeout=float(os.environ.get("DDCMI_PREFLIGHT_TIMEOUT", "60")))
        m.upload_local()
    elif world == 1:
        os.environ.pop("DDCMI_RCCL_LOOPBACK", None)
        m = MartiniHIP(s, device=local_rank)
    else:
        # spatial decomposition: this rank uploads the beads of its brick; halo exchange
        # and migration run inside libddcmi over RCCL point-to-point (include/ddcmi.h)
        owner = domain_of(s, grid)
        m = MartiniRank(s, np.flatnonzero(owner == rank), device=local_rank)
        if transport == "host":
            m.comm_init_host(rdzv, grid)
        else:
            buf = ctypes.create_string_buffer(128)
            if rank == 0:
                assert m.lib.ddcmi_comm_unique_id(buf) == 0
            uid = rdzv.bcast(buf.raw, 0)                                 # MPI_Bcast of the id in ddcMD
            try:
                m.comm_init(rank, world, uid, grid)
            except Exception as ex:
                # (typically: several ranks of this launch see the SAME device -- RCCL refuses two ranks on one GPU; tests on a one-GPU box use
                #  DDCMI_TRANSPORT=host + DDCMI_BENCH_SINGLE_DEVICE=1)
                sys.stderr.write("bench.py rank %d of %d: the RCCL communicator could not be created on device %d: %s -- do several ranks of this launch see the same device?  "
                                 "RCCL wants one GPU per rank (a one-GPU box: DDCMI_TRANSPORT=host DDCMI_BENCH_SINGLE_DEVICE=1, a new launch)\n" % (rank, world, local_rank, ex))
                sys.stderr.flush()
                os._exit(4)
        # the first real exchange of the launch is a CHECKED one, before any state goes up and before any timing: one grouped exchange
        # of a known pattern with every peer of the brick plan, one 24-double all-reduce, one int all-gather (ddcmi_comm_preflight).
        # A mismatch or the deadline ends the launch on every rank with the stage / peer / direction in the message (exit 4; a rerun
        # over DDCMI_TRANSPORT=host is a NEW launch the caller starts: this process has touched the GPU and never re-execs)
        from ddcmd_amd.martini import DdcmiError
        try:
            preflight = m.preflight(timeout=float(os.environ.get("DDCMI_PREFLIGHT_TIMEOUT", "60")))
        except DdcmiError as ex:
            sys.stderr.write("bench.py rank %d of %d: communicator preflight failed (%s transport): %s\n" % (rank, world, transport, ex))
            sys.stderr.flush()
            os._exit(4)
        m.upload_local()
    m.eval_forces()                       # firstEnergyCall (masters.c:579)
    # RCCL prints a version banner through C stdio at communicator creation; flush it now so that the JSON
    # line below is the last line of stdout
    ctypes.CDLL(None).fflush(None)
    thermostat = any(int(t) == 1 for t in np.asarray(s.group_type).ravel())
    if thermostat:
        m.group_temperatures()            # the temperature Berendsen scales with (published by eval_energyInfo in the reference)
    if equil < 0:
        equil = 200 if workload == "water" else 0
    done = 0
    while done < equil:                   # (in rebuild periods, so that a Berendsen group sees its temperature as in a production run)
        k = min(20, equil - done)
        m.step(k)
        done += k
        if thermostat:
            m.group_temperatures()
    m.step(warmup)
    if thermostat:
        m.group_temperatures()
    m.sync()

    def barrier():
        m.sync()                          # this rank's stream is drained ...
        if rdzv is not None:
            rdzv.barrier()                # ... and so is everybody else's

    nwin = max(1, -(-max(steps, MIN_TIMED_STEPS) // WINDOW))
    m.timing(True)
    reb0 = m.list_stats()["rebuilds"]
    wins = []
    for _ in range(nwin):
        barrier()
        t0 = time.perf_counter()
        m.step(WINDOW)
        barrier()
        el = time.perf_counter() - t0
        if rdzv is not None:
            el = float(rdzv.allreduce([el], "max")[0])       # the slowest rank's time
        wins.append(el)
    launches, kernel_ms = m.timing_read()
    launches_f, kernel_ms_f = m.timing_fused()
    m.timing(False)
    e, vir, rk, tion = m.energies()
    st = m.list_stats()
    nlocal = int(m.lib.ddcmi_nlocal(m.ctx))
    epot, ekin = e["total"], rk
    comm = None
    if rdzv is not None or loopback:
        cs = m.comm_stats()
        comm = {"transport": cs["transport"], "rccl_version": cs["rccl_version"], "halo_beads_sent_per_step_rank0": cs["send_beads"],
                "halo_messages_per_step_rank0": cs["send_msgs"], "halo_bytes_per_step_rank0": cs["send_beads"] * 24,
                "preflight_rank0": preflight,
                "peers_rank0": [{"peer": pr, "send_bytes_per_step": 24 * sb, "recv_bytes_per_step": 24 * rb} for pr, sb, rb in m.comm_peers()]}
    if rdzv is not None:
        tot = m.allreduce([epot, ekin, float(nlocal), float(cs["send_beads"])])      # energyInfo.c allreduce()
        epot, ekin = float(tot[0]), float(tot[1])
        assert int(round(tot[2])) == s.natoms, "beads lost in migration"
        per_rank = rdzv.allgather(np.array([nlocal, cs["send_beads"], cs["send_msgs"]], dtype=np.int64))
        comm.update({"ranks_met": int(rdzv.allreduce([1.0])[0]), "beads_per_rank": [int(x) for x in per_rank[:, 0]],
                     "halo_beads_sent_per_step": [int(x) for x in per_rank[:, 1]], "peers_per_rank": [int(x) for x in per_rank[:, 2]],
                     "halo_bytes_per_step_all_ranks": int(round(tot[3])) * 24})
    med = sorted(wins)[len(wins) // 2]
    ms_per_step = med * 1e3 / WINDOW
    L = st["entries"] / float(max(nlocal, 1))
    # SURVEY 8(d), compulsory bytes per bead and launch.  The plain pair kernel: read r_i 24 + q_i 8 + type_i 4, write f_i 24, list 4 L.
    # Between print steps the pair kernel ends in the integrator's pass (BACK kick, kinetic terms,
    # FRONT kick, drift: SURVEY's rows "kick+KE" and "kick+drift"): the force never goes to memory (-24), v is read and written
    # (+48), the drifted r written (+24) -- each datum once.  The dominant kind of launch in the timed region is the one priced.
    # Systems with bonded terms: their kernels run first and leave their force on every bead (24 B); either kind of launch reads it (+24).
    nbonded = sum(int(m.terms[k].size) for k in ("bond_kb", "angle_k", "tors_k"))
    fb = 24.0 if nbonded > 0 else 0.0
    plain = {"bytes": 36.0 + 24.0 + 4.0 * L + fb, "launches": launches - launches_f, "ms": kernel_ms - kernel_ms_f}
    fusedk = {"bytes": 36.0 + 4.0 * L + 48.0 + 24.0 + fb, "launches": launches_f, "ms": kernel_ms_f}
    dom = fusedk if fusedk["ms"] > plain["ms"] else plain
    bytes_per_atom = dom["bytes"]
    t_kernel = dom["ms"] * 1e-3 / max(1, dom["launches"])
    achieved = bytes_per_atom * nlocal / t_kernel / 1e9
    other = plain if dom is fusedk else fusedk
    other_row = None
    if other["launches"] > 0:
        t_o = other["ms"] * 1e-3 / other["launches"]
        other_row = {"kernel": "k_nonbond<FUSE>" if other is fusedk else "k_nonbond (plain: print and last steps)", "launches": other["launches"],
                     "kernel_ms_avg": t_o * 1e3, "algorithmic_bytes_per_atom_step": other["bytes"], "frac": other["bytes"] * nlocal / t_o / 1e9 / HBM_PEAK_GBS}
    # the second, honest bound: FP64 vector work of the pair kernel, ~45 flop per in-cutoff pair visit + ~10 per list entry
    # that fails the distance test; in-cutoff visits per bead from the density (4/3 pi rcut^3 rho: the full list visits a pair from both sides)
    rho = s.natoms / float(s.volume)
    n_in = 4.0 / 3.0 * 3.141592653589793 * float(s.rmax) ** 3 * rho
    flops = (45.0 * n_in + 10.0 * max(L - n_in, 0.0)) * nlocal
    # HBM bytes of the nonbonded kernel from the PMC passes committed under profiles/: quoted only when the file
    # was collected on this workload AND on these device sources (kernel_src_id), else null
    traffic = None
    try:
        if world == 1 and not loopback:
            for fn in sorted(os.listdir(os.path.join(ROOT, "profiles")), reverse=True):
                if not fn.endswith("traffic.json"):
                    continue
                t = json.load(open(os.path.join(ROOT, "profiles", fn)))
                if t.get("workload") == wname and t.get("kernel_src_id") == kernel_source_id():
                    traffic = float(t["traffic_bytes_per_launch"])
                    break
    except Exception:
        traffic = None
    traffic_source = "committed: profiles/*traffic.json of these device sources (tools/profile_r04.sh)" if traffic else None
    res = {
        "value": s.natoms / (med / WINDOW), "ms_per_step": ms_per_step, "ms_per_step_mean": sum(wins) * 1e3 / (len(wins) * WINDOW),
        "steps_timed": nwin * WINDOW, "window_steps": WINDOW, "window_ms": [round(w * 1e3, 4) for w in wins],
        "ns_per_day": (WINDOW / med) * dt_fs * 1e-6 * 86400.0,
        "config": {"workload": wname, "beads_total": s.natoms, "beads_rank0": nlocal,
                   "lattice": lattice, "lattice_n": lattice_n,
                   "stands_for": ("BASELINE configs[3] '4M-bead Martini water' (SURVEY 8d: 4 096 000 on a simple-cubic start that explodes at 20 fs; "
                                  "FCC 4 n^3 at the same density with n = %d: the smallest even n with 4 n^3 >= 4 096 000)" % HEADLINE_N
                                  if (workload == "water" and n == HEADLINE_N) else None),
                   "rcut_A": float(ddcmd_amd.units_convert(s.rmax, None, "Angstrom")), "skin_A": float(ddcmd_amd.units_convert(s.deltaR, None, "Angstrom")),
                   "dt_fs": dt_fs, "list_rebuild_every": int(s.updateRate),
                   "bonded_terms": {k: int(m.terms[k].size) for k in ("bond_kb", "angle_k", "tors_k")},
                   "energy_virial_every_step": True, "equilibration_steps_untimed": equil,
                   "parallelism": ("spatial decomposition %dx%dx%d, %s (control plane: libddcmi TCP rendezvous, no torch)"
                                   % (grid + ("RCCL p2p halo" if transport != "host" else "host-staged TCP halo",))) if world > 1
                                  else ("single GPU, images through RCCL loopback" if loopback else "single GPU"),
                   "list_entries_per_atom": L, "image_or_halo_beads_rank0": st["images"], "rebuilds_in_timed_region": st["rebuilds"] - reb0},
        "roofline": {"bound": "hbm", "kernel": "k_nonbond<FUSE> (pair kernel + the integrator's pass as its epilogue)" if dom is fusedk else "k_nonbond", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source, "dominant_is_fused": dom is fusedk,
                     "hbm_frac_measured": (traffic / t_kernel / 1e9 / HBM_PEAK_GBS) if traffic else None,
                     "fp64_valu_frac": flops / t_kernel / 1e12 / FP64_VALU_PEAK_TFLOPS,
                     "fp64_valu_note": "(45 flop x %.1f in-cutoff pair visits + 10 x %.1f rejected entries) per bead over %.1f TFLOP/s" % (n_in, max(L - n_in, 0.0), FP64_VALU_PEAK_TFLOPS),
                     "algorithmic_bytes_per_atom_step": bytes_per_atom, "kernel_ms_avg": t_kernel * 1e3, "launches": dom["launches"], "other_launches": other_row,
                     "note": "rank 0's kernel on its own beads" if world > 1 else "whole box"},
        "check": {"epot": epot, "ekin": ekin},
    }
    if comm:
        res["comm"] = comm
    m.close()
    return res


def spawn_ranks(n):
    """`python bench.py --gpus N` with no launcher around it: start the N ranks ourselves -- N fresh child processes of this
    very command line with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set, as `python -m torch.distributed.run
    --nproc-per-node N` would.  The parent has loaded no library and touched no device (and never execs): it relays the
    children's output (rank 0 prints the JSON line), ends the others when one fails, and returns the first non-zero exit code."""
    import subprocess
    if n not in (2, 4, 8):
        sys.stderr.write("bench.py supports 1, 2, 4 or 8 GPUs (2x1x1, 2x2x1, 2x2x2 bricks)\n")
        return 2
    import tempfile
    rc = 0
    # rank 0 listens on a port of its own choosing and publishes it through a file only this launch knows (host/rdzv.c) -- no port
    # picked here that somebody else could take in between
    with tempfile.TemporaryDirectory(prefix="ddcmi_bench_") as d:
        procs = []
        for r in range(n):
            env = dict(os.environ)
            env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(n), "LOCAL_WORLD_SIZE": str(n), "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "0",
                        "DDCMI_RDZV_FILE": os.path.join(d, "port"), "HSA_ENABLE_IPC_MODE_LEGACY": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")})
            env.pop("DDCMI_RDZV_PORT", None)
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))      # stdout / stderr inherited: one JSON line from rank 0
        alive = set(range(n))
        while alive:
            for r in sorted(alive):
                c = procs[r].poll()
                if c is None:
                    continue
                alive.discard(r)
                if c != 0 and rc == 0:
                    rc = c if c > 0 else 1
                    sys.stderr.write("bench.py: rank %d of %d exited with code %d; ending the other ranks\n" % (r, n, c))
                    for q in alive:
                        procs[q].terminate()      # (exactly the PIDs started above)
            time.sleep(0.05)
    return rc


def also_rows(reps, local_rank, transport):
    """the other single-GPU configurations, each timed like the headline (>= 100 steps): BASELINE configs[2] (1 M-bead water), configs[4] (the
    ~2 M-bead lipid bilayer, also at the deck's own cadence and under 20 / 40 LJ types) and one rank's brick of the 8-GPU runs of configs[3] and
    configs[4] through the RCCL loopback.  Returns (rows, the bench arguments of each row's PMC child runs or None)."""
    brick_reps = ",".join(str(max(1, int(x) // 2)) for x in reps.split(","))      # 12,12,6 -> 6,6,3: what one of 2x2x2 ranks owns
    rows, pmc = [], []
    for kw in (dict(workload="water", n=64, loopback=False, tag="BASELINE configs[2]: 1M-bead water, rebuild every 20 steps", pmc=["--lattice", "64"]),
               dict(workload="lipid", n=None, loopback=False, tag="BASELINE configs[4]: lipid bilayer in water, ~2M beads, bonded terms, Berendsen, at the reference decks' cadence (dt 20 fs, rebuild every 20 steps)", pmc=["--workload", "lipid", "--reps", reps]),
               dict(workload="lipid", n=None, loopback=False, cadence="deck", tag="the same bilayer at the lipid deck's own cadence (dt 10 fs, rebuild every 10 steps): rounds 1-4's row", pmc=None),
               dict(workload="lipid", n=None, loopback=False, types=20, tag="the same bilayer under 20 LJ types (every type split into copies of itself): the class table of a mid-size Martini deck", pmc=["--workload", "lipid", "--reps", reps, "--types", "20"]),
               dict(workload="lipid", n=None, loopback=False, types=40, tag="the same bilayer under 40 LJ types / 48 (type, charge) classes: the pair table in two levels (k_nonbond<LVL>)", pmc=["--workload", "lipid", "--reps", reps, "--types", "40"]),
               dict(workload="water", n=HEADLINE_N // 2, loopback=True, tag="one rank's brick of the 8-GPU run of the headline box (n/2 per axis), periodic images through the RCCL loopback", pmc=["--lattice", str(HEADLINE_N // 2), "--rccl-loopback"]),
               dict(workload="lipid", n=None, loopback=True, reps=brick_reps, tag="one rank's brick of the 8-GPU run of BASELINE configs[4] (the bilayer tiled %s: an eighth of %s), bonded terms by gid, Berendsen temperature all-reduced, periodic images through the RCCL loopback" % (brick_reps, reps), pmc=["--workload", "lipid", "--reps", brick_reps, "--rccl-loopback"]),
               dict(workload="water", n=50, loopback=True, tag="the 500k-bead brick of rounds 1-4 (n = 50: one eighth of the 4.0M box), same loopback -- kept for continuity with VERDICT r4's target", pmc=None)):
        try:
            sys.stderr.write("bench.py rows: %s\n" % kw["tag"][:90]); sys.stderr.flush()
            r = run_config(kw["workload"], kw["n"], kw.get("reps", reps), 100, 20, -1, 1, 0, local_rank, None, transport, kw["loopback"], kw.get("types", 0), kw.get("cadence", "reference"))
            rows.append({"what": kw["tag"], "workload": r["config"]["workload"], "value": r["value"], "unit": "atom-steps/s", "ms_per_step": r["ms_per_step"],
                         "steps_timed": r["steps_timed"], "window_ms": r["window_ms"], "rebuilds_in_timed_region": r["config"]["rebuilds_in_timed_region"],
                         "list_entries_per_atom": r["config"]["list_entries_per_atom"], "parallelism": r["config"]["parallelism"],
                         "dt_fs": r["config"]["dt_fs"], "list_rebuild_every": r["config"]["list_rebuild_every"],
                         "roofline": r["roofline"], "comm": r.get("comm")})
            pmc.append(kw["pmc"])
        except Exception as ex:      # (the other rows stand on their own)
            rows.append({"what": kw["tag"], "error": str(ex)})
            pmc.append(None)
    return rows, pmc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--lattice", dest="n", type=int, default=HEADLINE_N, help="FCC lattice edge: 4*n^3 beads (102 -> 4.24M: the headline; 100 -> 4.0M: rounds 1-4; 64 -> 1.05M, 25 -> 62.5k)")
    ap.add_argument("--cpu-n", type=int, default=26, help="lattice edge of the CPU-baseline sample (26 -> 70 304 beads: BASELINE configs[1] '64k-bead water', the smallest FCC box with at least 64 000)")
    ap.add_argument("--cpu-1m", type=int, default=1, help="1: also time the oracle on the 1.05 M-bead box of configs[2] for one rebuild period (SURVEY 8d: '1 M if time allows', ~20 s); 0: skip")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-also", action="store_true", help="N=1: only the headline workload, not the other single-GPU configs")
    ap.add_argument("--no-pmc", action="store_true", help="N=1 headline: do not measure the pair kernel's HBM bytes live (two short child runs under rocprofv3 --pmc)")
    ap.add_argument("--workload", choices=("water", "lipid"), default="water",
                    help="water: the headline Martini water box; lipid: tests/golden/lipid_deck (DPPC-style bilayer patch in water, "
                         "all bonded term kinds, charges, Berendsen) tiled --reps times")
    ap.add_argument("--rccl-loopback", action="store_true",
                    help="N=1 only: reach the periodic images through a 1-rank RCCL communicator (the multi-GPU transport on one GPU)")
    ap.add_argument("--cadence", choices=("reference", "deck"), default="reference",
                    help="lipid workload: reference = dt 20 fs, rebuild every 20 steps (what the reference's shipped decks use); deck = the lipid deck's own 10 fs / 10")
    ap.add_argument("--types", type=int, default=0, help="lipid workload: relabel the beads to this many LJ types (same physics, bigger class table; 0: the deck's own 6)")
    ap.add_argument("--reps", default="12,12,6", help="lipid workload: copies of the 2363-bead deck along x,y,z (12,12,6 -> 2.04M beads)")
    ap.add_argument("--equil", type=int, default=-1,
                    help="untimed steps in front of the warm-up that take the synthetic water box from its lattice start (50 K, FCC + jitter) to its "
                         "liquid state (~307 K after 200 steps): default 200 for water, 0 for the lipid deck (a relaxed restart)")
    ap.add_argument("--check-runtime", action="store_true", help="rendezvous + library load only: print which HIP/RCCL runtime is mapped, touch no device")
    ap.add_argument("--rows-only", action="store_true", help="(internal) run the `also` workloads and print them as one JSON object: what the N=1 headline run starts as a child process")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    if "WORLD_SIZE" in os.environ and int(os.environ["WORLD_SIZE"]) != args.gpus:
        # a launcher that started W ranks of a bench asked for N GPUs: a flat scaling curve in the making -- refuse
        sys.stderr.write("bench.py: --gpus %d but WORLD_SIZE=%s: the launcher's world and --gpus must agree\n" % (args.gpus, os.environ["WORLD_SIZE"]))
        sys.exit(2)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus))      # `python bench.py --gpus N` without a launcher: this process only starts and watches the N ranks

    # stdout carries the ONE JSON line and nothing else: RCCL prints its version banner to C stdout when a communicator comes up
    # (the loopback brick of `also`, every multi-rank run), so file descriptor 1 points at stderr while the libraries work and the
    # line goes out through a duplicate of the real stdout at the end
    sys.stdout.flush()
    real_stdout = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if os.environ.get("DDCMI_BENCH_SINGLE_DEVICE"):      # all ranks on device 0 (tests on a one-GPU box; needs DDCMI_TRANSPORT=host: RCCL refuses two ranks on a device)
        local_rank = 0
    elif not args.check_runtime:      # (--check-runtime touches no device: not even the count)
        # a launcher that gives every rank exactly ONE visible device (HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES per process): ordinal 0
        # there, not LOCAL_RANK.  Fewer devices than ranks in any other way is refused: two ranks folded onto one GPU would print a
        # scaling number that means nothing
        import ddcmd_amd
        lib0 = ddcmd_amd.load_library()
        lib0.ddcmi_device_count.restype = ctypes.c_int
        ndev = int(lib0.ddcmi_device_count())
        if ndev == 1:
            local_rank = 0
        elif 0 < ndev <= local_rank:
            sys.stderr.write("bench.py: LOCAL_RANK %d but only %d devices are visible: one rank per GPU (DDCMI_BENCH_SINGLE_DEVICE=1 + DDCMI_TRANSPORT=host "
                             "shares one device on purpose)\n" % (local_rank, ndev))
            sys.exit(2)
    transport = os.environ.get("DDCMI_TRANSPORT", "rccl")
    # No torch in this process: libddcmi.so is built and validated against /opt/rocm's HIP and RCCL, and
    # `import torch` would map torch's bundled copies of the same sonames first.  The control plane (the 128-byte
    # RCCL id, barriers, the max of the timings) runs over libddcmi's own TCP rendezvous (host/rdzv.c); the data
    # path (halo exchange, migration, count all-gathers, energy all-reduce) over libddcmi's RCCL communicator.
    rdzv = None
    if world > 1:
        # a launch that wedges (a peer that died inside a collective, a fabric that never answers) must end by itself: after
        # DDCMI_BENCH_DEADLINE seconds (default 900) every rank says so and exits 3 -- no hang for the driver to time out on
        import threading

        def _deadline():
            # (a thread, not SIGALRM: a Python signal handler cannot run while the main thread sits inside a library call)
            time.sleep(float(os.environ.get("DDCMI_BENCH_DEADLINE", "900")))
            sys.stderr.write("bench.py rank %d of %d: no result after %s s, giving up\n" % (rank, world, os.environ.get("DDCMI_BENCH_DEADLINE", "900")))
            sys.stderr.flush()
            os._exit(3)
        threading.Thread(target=_deadline, daemon=True).start()
        from ddcmd_amd.martini import Rendezvous, DdcmiError
        try:
            # a rank that never arrives ends the launch with a message and a non-zero exit after two minutes: no hang, no re-exec
            rdzv = Rendezvous.from_env(timeout=float(os.environ.get("DDCMI_RDZV_TIMEOUT", "120")))
        except DdcmiError as ex:
            sys.stderr.write("bench.py rank %d of %d: rendezvous failed: %s\n" % (rank, world, ex))
            sys.exit(2)
    if os.environ.get("DDCMI_BENCH_FAIL_RANK") == str(rank) and world > 1:      # test hook: this rank dies after the rendezvous
        os._exit(7)
    if args.check_runtime:
        # the launcher and the runtime binding, without touching a device
        import ddcmd_amd
        ddcmd_amd.load_library()
        tok = rdzv.bcast(b"ddcmi-runtime-check-%06d" % os.getpid() if rank == 0 else bytes(26), 0) if rdzv else b""
        n = rdzv.allreduce([1.0])[0] if rdzv else 1.0
        print(json.dumps({"rank": rank, "world": world, "ranks_met": int(n), "token": tok.decode(), "runtime_libs": runtime_libs(),
                          "torch_loaded": "torch" in sys.modules}), file=real_stdout, flush=True)
        if rdzv:
            rdzv.barrier()
            rdzv.close()
        return

    if args.rows_only:
        rows, pmc = also_rows(args.reps, local_rank, transport)
        ctypes.CDLL(None).fflush(None)
        print(json.dumps({"also": rows, "pmc": pmc}), file=real_stdout, flush=True)
        return
    res = run_config(args.workload, args.n, args.reps, args.steps, args.warmup, args.equil, world, rank, local_rank, rdzv, transport, args.rccl_loopback, args.types, args.cadence)
    out = {
        "metric": "atom_steps_per_sec", "value": res.pop("value"), "unit": "atom-steps/s",
        "n_gpus": world, "steps": res["steps_timed"], "steps_requested": args.steps, "warmup": args.warmup,
        "ms_per_step": res.pop("ms_per_step"), "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
    }
    out.update(res)
    out["runtime_libs"] = runtime_libs()
    headline = world == 1 and args.workload == "water" and args.n == HEADLINE_N and not args.rccl_loopback
    pmc_jobs = []      # (row or None for the headline, bench arguments of the PMC child runs, dominant launch fused?)
    if headline and not args.no_also:
        # The other single-GPU configurations run in ONE fresh child process (this very file with --rows-only), started when this process
        # has finished its own GPU work: a fault in one of them costs the `also` block, never the headline's line.  (Round 6 saw "Memory access
        # fault by GPU node" end one default run in eight; the cause -- the context object registered with hipHostRegister at recycled heap
        # addresses -- is gone from the library, the isolation stays.)  The PMC passes of every row follow when NOTHING else of this launch
        # uses the GPU any more.
        import subprocess
        cmd = [sys.executable, os.path.abspath(__file__), "--rows-only", "--reps", args.reps]
        try:
            r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=None, text=True, timeout=float(os.environ.get("DDCMI_BENCH_ROWS_TIMEOUT", "900")))
            lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
            if r.returncode != 0 or not lines:
                out["also"] = [{"error": "the child process of the `also` workloads exited with code %d and %s" % (r.returncode, "no result" if not lines else "a partial result")}]
            else:
                rows = json.loads(lines[-1])
                out["also"] = rows["also"]
                for row, job in zip(rows["also"], rows["pmc"]):
                    if job and "error" not in row:
                        pmc_jobs.append((row, job, row["roofline"].get("dominant_is_fused", False)))
        except Exception as ex:      # the headline stands on its own
            out["also"] = [{"error": "the `also` workloads: %s" % ex}]
    under_profiler = "rocprof" in os.environ.get("LD_PRELOAD", "").lower() or any(k.startswith(("ROCPROF", "ROCP_TOOL")) for k in os.environ)
    if headline and not args.no_pmc and not under_profiler:      # (a run that is itself being profiled starts no profiler of its own)
        # roofline.traffic measured in THIS run: this process and its rows child have finished their GPU work; every pass is a fresh child process
        for row, job, fused in [(None, ["--lattice", str(args.n)], out["roofline"].get("dominant_is_fused", False))] + pmc_jobs:
            lt = live_traffic(job, fused)
            if lt:
                rf = out["roofline"] if row is None else row["roofline"]
                t_k = rf["kernel_ms_avg"] * 1e-3
                rf.update({"traffic": lt["bytes_per_launch"], "traffic_source": lt["source"], "traffic_FETCH_SIZE_KiB": lt["FETCH_SIZE_KiB"],
                           "traffic_WRITE_SIZE_KiB": lt["WRITE_SIZE_KiB"], "hbm_frac_measured": lt["bytes_per_launch"] / t_k / 1e9 / HBM_PEAK_GBS})
    if rank == 0 and world == 1 and not args.no_cpu:
        out["cpu_baseline"] = cpu_baseline(args.cpu_n, one_million=bool(args.cpu_1m) and headline)      # N=1 only (the contract): bounded samples on one host core
    if rdzv is not None:
        rdzv.barrier()
        rdzv.close()
    if rank == 0:
        ctypes.CDLL(None).fflush(None)                      # anything the libraries left in C stdio goes out first (to stderr)
        print(json.dumps(out), file=real_stdout, flush=True)


if __name__ == "__main__":
    main()
