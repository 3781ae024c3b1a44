"""Reduce a tools/profile_round.sh output directory to one JSON summary:
per-kernel average durations (kernel-trace stats) and per-launch PMC means for
k_nonbond."""
import csv, glob, json, os, sys, collections

def main(out):
    res = {"kernel_stats": [], "pmc_k_nonbond_mean_per_launch": {}}
    for f in glob.glob(os.path.join(out, "stats", "**", "*kernel_stats.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            res["kernel_stats"].append({"name": row["Name"][:60], "calls": int(row["Calls"]),
                                        "avg_us": float(row["AverageNs"]) / 1e3, "pct": float(row["Percentage"])})
    for f in glob.glob(os.path.join(out, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
        acc = collections.defaultdict(list)
        for row in csv.DictReader(open(f)):
            if "k_nonbond" in row["Kernel_Name"]:
                acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
        for k, v in acc.items():
            res["pmc_k_nonbond_mean_per_launch"][k] = sum(v) / len(v)
            res["pmc_k_nonbond_mean_per_launch"][k + "_launches"] = len(v)
    print(json.dumps(res, indent=1))

if __name__ == "__main__":
    main(sys.argv[1])
