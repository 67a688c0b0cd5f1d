"""Copies the summaries of one tools/collect_profiles.sh run (gpurun_out/<tag>/) into profiles/<tag>_* and builds
profiles/<tag>_pmc_summary.json (per kernel and counter: average counter value per launch, summed over the
dispatch's counter instances) from the separate --pmc passes.  usage: python tools/publish_profiles.py <tag>"""
import csv
import glob
import json
import os
import shutil
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
src = os.path.join(ROOT, "gpurun_out", tag)
dst = os.path.join(ROOT, "profiles")

for a, b in [("bench_default.json", "bench_default.json"), ("bench_cfg4_1gpu_262144.json", "bench_cfg4_1gpu_262144.json"),
             ("bench_cfg4_shard_32768.json", "bench_cfg4_shard32768.json"), ("configs.txt", "configs.txt"),
             ("kt_cfg2/cfg2_kernel_stats.csv", "kernel_stats_cfg2.csv"), ("kt_cfg4/cfg4_kernel_stats.csv", "kernel_stats_cfg4_shard32768.csv")]:
    p = os.path.join(src, a)
    if os.path.exists(p):
        shutil.copy(p, os.path.join(dst, f"{tag}_{b}"))

summary = defaultdict(dict)
for d in sorted(glob.glob(os.path.join(src, "pmc_*"))):
    f = os.path.join(d, "pmc_counter_collection.csv")
    if not os.path.isfile(f):
        continue
    per_dispatch = defaultdict(float)  # (kernel, counter, dispatch) -> value summed over instances
    with open(f, newline="") as fh:
        for row in csv.DictReader(fh):
            name = row["Kernel_Name"].split("(")[0]
            if not name.startswith("void k_"):
                continue
            per_dispatch[(name, row["Counter_Name"], row["Dispatch_Id"])] += float(row["Counter_Value"])
    acc = defaultdict(list)
    for (name, counter, _), v in per_dispatch.items():
        acc[(name, counter)].append(v)
    suffix = "_cfg4" if d.endswith("_cfg4") else ""
    for (name, counter), vals in acc.items():
        summary[name][counter + suffix] = {"avg_per_launch": sum(vals) / len(vals), "launches": len(vals)}
with open(os.path.join(dst, f"{tag}_pmc_summary.json"), "w") as fh:
    json.dump(summary, fh, indent=1)
print("kernels:", len(summary))
for name, c in summary.items():
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        mb = (2 * c["FETCH_SIZE"]["avg_per_launch"] + c["WRITE_SIZE"]["avg_per_launch"]) * 1024 / 1e6
        busy = ""
        if "SQ_VALU_MFMA_BUSY_CYCLES" in c and c.get("SQ_BUSY_CU_CYCLES", {}).get("avg_per_launch"):
            busy = f"  mfma busy/cu busy {c['SQ_VALU_MFMA_BUSY_CYCLES']['avg_per_launch'] / c['SQ_BUSY_CU_CYCLES']['avg_per_launch'] / 4:.2f}"
        print(f"  {name[:70]:70s} HBM {mb:9.2f} MB/launch{busy}")
