"""Copies the summaries of one tools/collect_profiles.sh run (gpurun_out/<tag>/) into profiles/<tag>_* and builds
profiles/<tag>_pmc_summary.json ({workload: {kernel: {counter: {avg_per_launch, launches}}}} flattened to
"<kernel>" keys for the headline workload, "<kernel> [cfg3]" etc. for the others) from the separate --pmc passes.
usage: python tools/publish_profiles.py <tag>"""
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from pmc_summary import summarize  # noqa: E402

tag = sys.argv[1]
src = os.path.join(ROOT, "gpurun_out", tag)
dst = os.path.join(ROOT, "profiles")
# a collection is published WITH the parity table of the build it measured, or not at all
pm = os.path.join(src, "parity_measured.json")
if not (os.path.exists(pm) and os.path.getsize(pm) > 1000):
    sys.exit(f"{pm} is missing: tools/collect_profiles.sh runs the GPU suite at its end -- a collection without the parity table "
             "of its build is not published")

copies = [("bench_default.json", "bench_default.json"), ("bench_cfg3.json", "bench_cfg3.json"),
          ("bench_cfg4_1gpu_262144.json", "bench_cfg4_1gpu_262144.json"), ("bench_cfg4_shard_32768.json", "bench_cfg4_shard32768.json"),
          ("configs.txt", "configs.txt"), ("simple.txt", "simple_hbm.txt"),
          ("bench_default_split_calls.json", "bench_default_split_calls.json"), ("bench_default_graph.json", "bench_default_graph.json"),
          ("bench_default_one_wave_per_tile.json", "bench_default_one_wave_per_tile.json"), ("simple_no_mfma.txt", "simple_no_mfma.txt"),
          ("bench_driver_cmd.json", "bench_driver_cmd.json"), ("step_ramp_clocks.txt", "step_ramp_clocks.txt"),
          ("bench_default_fp32_mfma_everywhere.json", "bench_default_fp32_mfma_everywhere.json"),
          ("bench_default_b6_stashing_forward.json", "bench_default_b6_stashing_forward.json"), ("configs_fp32_forward.txt", "configs_fp32_forward.txt"),
          ("bench_default_pair_dw_fp32.json", "bench_default_pair_dw_fp32.json"),
          ("bench_cfg4_shard_32768_fp32_mfma.json", "bench_cfg4_shard32768_fp32_mfma.json"),
          ("bench_cfg4_1gpu_262144_fp32_mfma.json", "bench_cfg4_1gpu_262144_fp32_mfma.json"),
          ("bench_cfg3_bwd_fp32_mfma.json", "bench_cfg3_bwd_fp32_mfma.json"), ("bench_cfg3_fwd_fp32_mfma.json", "bench_cfg3_fwd_fp32_mfma.json"),
          ("configs_deep_off.txt", "configs_deep_off.txt"), ("configs_f64_scalar.txt", "configs_f64_scalar.txt"),
          ("kt_gen/gen_kernel_stats.csv", "kernel_stats_deep_f64.csv"),
          ("parity_ab_default.json", "parity_ab_default.json"), ("parity_ab_fp32_mfma.json", "parity_ab_fp32_mfma.json"),
          ("parity_ab_default.txt", "parity_ab_default.txt"), ("parity_ab_fp32_mfma.txt", "parity_ab_fp32_mfma.txt"),
          ("split_bias_probe.txt", "split_bias_probe.txt"), ("split_mfma_probe.txt", "split_mfma_probe.txt"),
          ("parity_measured.json", "parity_measured.json"), ("gpu_suite.txt", "gpu_suite.txt"), ("kernel_resources.txt", "kernel_resources.txt")]
for w in ("cfg1", "cfg2", "cfg3", "cfg4", "cfg5", "simple"):
    copies.append((f"kt_{w}/{w}_kernel_stats.csv", f"kernel_stats_{w}.csv"))
for a, b in copies:
    p = os.path.join(src, a)
    if os.path.exists(p) and os.path.getsize(p) > 0:  # (two 0-byte files of a failed step were committed in round 5)
        shutil.copy(p, os.path.join(dst, f"{tag}_{b}"))

summary = {}
for w in ("cfg2", "cfg3", "cfg4", "simple"):
    dirs = sorted(glob.glob(os.path.join(src, f"pmc_{w}_*/")))
    for name, counters in summarize(dirs).items():
        summary[name if w == "cfg2" else f"{name} [{w}]"] = counters
with open(os.path.join(dst, f"{tag}_pmc_summary.json"), "w") as fh:
    json.dump(summary, fh, indent=1)
print("kernels:", len(summary))
for name, c in summary.items():
    line = f"  {name[:78]:78s}"
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        line += f" HBM {(2 * c['FETCH_SIZE']['avg_per_launch'] + c['WRITE_SIZE']['avg_per_launch']) * 1024 / 1e6:9.2f} MB/launch"
    if "SQ_VALU_MFMA_BUSY_CYCLES" in c and c.get("SQ_BUSY_CU_CYCLES", {}).get("avg_per_launch"):
        line += f"  mfma busy {c['SQ_VALU_MFMA_BUSY_CYCLES']['avg_per_launch'] / c['SQ_BUSY_CU_CYCLES']['avg_per_launch'] / 4:.2f}"
    print(line)
