"""gpurun_out/prof_r02_<tag> (tools/profile_r02.sh) -> the tracked evidence under profiles/:
   r02_<tag>_kernel_stats.csv   rocprofv3 --kernel-trace --stats summary of the bench command
   r02_<tag>_bench.log          the JSON lines of the profiled and the plain run of the same command
   r02_<tag>_traffic.json       HBM bytes per k_nonbond launch from the FETCH_SIZE / WRITE_SIZE passes (gfx950 correction of
                                MI355X_MICROARCH.md: FETCH_SIZE doubled, both in KiB), SQ counters, and the identity of what was
                                measured: workload name and a hash of the device sources (bench.py quotes `traffic` only on a match)
   python tools/curate_r02.py <tag> [<tag> ...]"""
import glob, json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench

for tag in sys.argv[1:]:
    out = os.path.join(ROOT, "gpurun_out", "prof_r02_" + tag)
    d = json.load(open(os.path.join(out, "summary.json")))
    stats = glob.glob(os.path.join(out, "stats", "**", "*kernel_stats.csv"), recursive=True)
    shutil.copy(stats[0], os.path.join(ROOT, "profiles", "r02_%s_kernel_stats.csv" % tag))
    lines = []
    for name in ("bench_stats.log", "bench_plain.log"):
        for l in open(os.path.join(out, name)):
            if l.startswith("{"):
                lines.append(json.loads(l))
    with open(os.path.join(ROOT, "profiles", "r02_%s_bench.log" % tag), "w") as f:
        f.write("# first line: under rocprofv3 --kernel-trace --stats; second: the same command un-profiled\n")
        for l in lines:
            f.write(json.dumps(l) + "\n")
    p = d["pmc_k_nonbond_mean_per_launch"]
    nb = [k for k in d["kernel_stats"] if "k_nonbond" in k["name"]]
    plain = lines[-1]
    t = {"workload": plain["config"]["workload"], "kernel_src_id": bench.kernel_source_id(),
         "bench_args": " ".join(a for a in open(os.path.join(out, "bench_stats.log")).read().split("\n")[0:0]) or None,
         "method": "tools/profile_r02.sh: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (4 steps + 2 warm-up), mean over the k_nonbond launches of each pass; SQ counters in two further passes; kernel durations from a --kernel-trace --stats pass",
         "units": "FETCH_SIZE / WRITE_SIZE are reported in KiB",
         "gfx950_correction": "MI355X_MICROARCH.md, HBM: FETCH_SIZE reports exactly 1/2 of the bytes of wide coalesced streaming reads on gfx950 -> doubled; WRITE_SIZE as is; other access widths are uncalibrated, so the doubled figure is an upper estimate for the gather part",
         "FETCH_SIZE_KiB_per_launch": p.get("FETCH_SIZE"), "WRITE_SIZE_KiB_per_launch": p.get("WRITE_SIZE"),
         "traffic_bytes_per_launch": (2.0 * p.get("FETCH_SIZE", 0.0) + p.get("WRITE_SIZE", 0.0)) * 1024.0,
         "k_nonbond_avg_us_rocprof": nb[0]["avg_us"] if nb else None, "k_nonbond_calls": nb[0]["calls"] if nb else None,
         "k_nonbond_avg_us_hip_events_plain_run": plain["roofline"]["kernel_ms_avg"] * 1e3,
         "algorithmic_bytes_per_launch": plain["roofline"]["algorithmic_bytes_per_atom_step"] * plain["config"]["beads_rank0"],
         "sq_counters_mean_per_launch": {k: v for k, v in p.items() if k.startswith("SQ_") and not k.endswith("_launches")},
         "kernel_stats_avg_us": {k["name"]: k["avg_us"] for k in d["kernel_stats"][:18]}}
    t.pop("bench_args")
    t["hbm_frac_measured_rocprof_time"] = t["traffic_bytes_per_launch"] / (t["k_nonbond_avg_us_rocprof"] * 1e-6) / 8e12 if nb else None
    json.dump(t, open(os.path.join(ROOT, "profiles", "r02_%s_traffic.json" % tag), "w"), indent=1)
    print(tag, "k_nonbond %.1f us (rocprof) / %.1f us (HIP events, plain)  traffic %.3e B  hbm_frac_measured %.3f  frac(contract) %.3f" % (
        t["k_nonbond_avg_us_rocprof"], t["k_nonbond_avg_us_hip_events_plain_run"], t["traffic_bytes_per_launch"], t["hbm_frac_measured_rocprof_time"], plain["roofline"]["frac"]))
