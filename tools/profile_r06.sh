#!/bin/bash
# The round's rocprofv3 evidence for ONE bench invocation, every capture a run of its own (kernel trace, then each --pmc set):
#   gpurun --timeout 900 -- 'bash tools/profile_r06.sh <tag> <bench args...>'
# -> gpurun_out/prof_<tag>/{kernel_stats.csv, bench.log, traffic.json}; copy what is to be judged into profiles/r06_<tag>_*.
set -u
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
tag=$1; shift
export TMPDIR=/tmp
root=$PWD
out="gpurun_out/prof_$tag"
rm -rf "$out"; mkdir -p "$out"
cd /tmp
# (every bench invocation below carries --no-also itself: with a headline command line that lacks it each pass would also run the seven `also`
#  workloads under the profiler and the PMC passes would hit their timeout with nothing written -- ADVICE r5)
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$root/$out/stats" -o s -- python3 "$root/bench.py" --no-cpu --no-pmc --no-also "$@" > "$root/$out/bench_profiled.log" 2>&1 || echo "WARNING: the kernel-trace pass exited with $?" | tee -a "$root/$out/warnings.txt"
python3 "$root/bench.py" --no-cpu --no-pmc --no-also "$@" > "$root/$out/bench_plain.log" 2>&1
for set in "FETCH_SIZE" "WRITE_SIZE" \
           "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_THREAD_CYCLES_VALU" \
           "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA SQ_INSTS_BRANCH"; do
   name=$(echo $set | tr ' ' '+' | cut -c1-30)
   timeout 300 rocprofv3 --pmc $set --output-format csv -d "$root/$out/pmc_$name" -o p -- python3 "$root/bench.py" --no-cpu --no-also "$@" --steps 60 --warmup 20 --no-pmc > "$root/$out/log_pmc_$name.txt" 2>&1 || echo "WARNING: the PMC pass '$set' exited with $? (timeout = 124): its counters are missing from traffic.json" | tee -a "$root/$out/warnings.txt"
done
cd "$root"
python3 - "$out" <<'PY'
import csv, glob, json, os, sys, collections
out = sys.argv[1]
res = {"method": "tools/profile_r06.sh: rocprofv3 --kernel-trace --stats (one run), a plain run, then --pmc FETCH_SIZE, --pmc WRITE_SIZE and three SQ counter "
                 "sets in separate 4-step runs; means over the launches of each kernel",
       "units": "FETCH_SIZE / WRITE_SIZE are reported in KiB",
       "gfx950_correction": "MI355X_MICROARCH.md, HBM: FETCH_SIZE reports exactly 1/2 of the bytes of wide coalesced streaming reads on gfx950 -> doubled; WRITE_SIZE as is"}
for f in glob.glob(os.path.join(out, "stats", "**", "*kernel_stats.csv"), recursive=True):
    rows = list(csv.DictReader(open(f)))
    open(os.path.join(out, "kernel_stats.csv"), "w").write(open(f).read())
    res["kernel_stats_avg_us"] = {r["Name"].split("(")[0][:72]: [int(r["Calls"]), round(float(r["AverageNs"]) / 1e3, 2)] for r in rows[:18]}
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(out, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
    for r in sorted(csv.DictReader(open(f)), key=lambda r: int(r["Dispatch_Id"])):
        k = r["Kernel_Name"].split("(")[0]
        if k.startswith("void "): k = k[5:]
        if not (k.startswith("k_nonbond") or k.startswith("k_tile")): continue
        base = k.split("<")[0]
        if base == "k_nonbond" and [x.strip() for x in k[k.index("<") + 1:k.rindex(">")].split(",")][-2] == "true": base = "k_nonbond_fused"      # the pair kernel with the integrator's pass as its epilogue
        acc[base][r["Counter_Name"]].append(float(r["Counter_Value"]))
res["counters_mean_per_launch"] = {k: {c: sum(v[-60:]) / len(v[-60:]) for c, v in d.items()} for k, d in acc.items()}
plain = res["counters_mean_per_launch"].get("k_nonbond", {})
if "FETCH_SIZE" in plain and "WRITE_SIZE" in plain:
    res["traffic_bytes_per_launch_plain"] = (2.0 * plain["FETCH_SIZE"] + plain["WRITE_SIZE"]) * 1024.0
# the kind of launch bench.py prices: the fused one wherever the workload takes it
nb = res["counters_mean_per_launch"].get("k_nonbond_fused", plain)
res["traffic_kernel"] = "k_nonbond<FUSE>" if "k_nonbond_fused" in res["counters_mean_per_launch"] else "k_nonbond"
if "FETCH_SIZE" in nb and "WRITE_SIZE" in nb:
    res["FETCH_SIZE_KiB_per_launch"] = nb["FETCH_SIZE"]; res["WRITE_SIZE_KiB_per_launch"] = nb["WRITE_SIZE"]
    res["traffic_bytes_per_launch"] = (2.0 * nb["FETCH_SIZE"] + nb["WRITE_SIZE"]) * 1024.0
for name in ("bench_plain.log", "bench_profiled.log"):
    for line in open(os.path.join(out, name)):
        if line.startswith("{"):
            b = json.loads(line)
            res[name.split(".")[0]] = {"ms_per_step": b["ms_per_step"], "kernel_ms_avg": b["roofline"]["kernel_ms_avg"], "frac": b["roofline"]["frac"]}
            res["workload"] = b["config"]["workload"]
            if name == "bench_plain.log":
                open(os.path.join(out, "bench.log"), "w").write(line)
import hashlib
h = hashlib.sha256()
d = os.path.join("ddcmd_amd", "csrc", "hip")
for f in sorted(os.listdir(d)):
    if f.endswith((".hip", ".inl", ".h")):
        h.update(open(os.path.join(d, f), "rb").read())
res["kernel_src_id"] = h.hexdigest()[:16]
json.dump(res, open(os.path.join(out, "traffic.json"), "w"), indent=1)
print(json.dumps({k: res[k] for k in res if k not in ("counters_mean_per_launch", "method", "units", "gfx950_correction")}, indent=1)[:3000])
PY
