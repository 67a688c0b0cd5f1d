"""Per-kernel averages of rocprofv3 --pmc passes.  usage: python tools/pmc_summary.py <out.json> <pmc_dir> [<pmc_dir> ...]
Each <pmc_dir> holds pmc_counter_collection.csv of ONE separate --pmc run.  Writes {kernel: {counter: {avg_per_launch,
launches}}} (values summed over the counter instances of a dispatch) and prints the HBM bytes (2 x FETCH_SIZE +
WRITE_SIZE, KiB -> bytes, gfx950 correction of MI355X_MICROARCH.md) and MFMA-busy fraction where available."""
import csv
import json
import os
import sys
from collections import defaultdict


def summarize(dirs):
    summary = defaultdict(dict)
    for d in dirs:
        f = os.path.join(d, "pmc_counter_collection.csv")
        if not os.path.isfile(f):
            continue
        per_dispatch = defaultdict(float)
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                name = row["Kernel_Name"].split("(")[0]
                if "k_" not in name:
                    continue
                per_dispatch[(name, row["Counter_Name"], row["Dispatch_Id"])] += float(row["Counter_Value"])
        acc = defaultdict(list)
        for (name, counter, _), v in per_dispatch.items():
            acc[(name, counter)].append(v)
        for (name, counter), vals in acc.items():
            summary[name][counter] = {"avg_per_launch": sum(vals) / len(vals), "launches": len(vals)}
    return summary


if __name__ == "__main__":
    out, dirs = sys.argv[1], sys.argv[2:]
    summ = summarize(dirs)
    with open(out, "w") as fh:
        json.dump(summ, fh, indent=1)
    for name, c in summ.items():
        line = f"{name[:64]:64s}"
        if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
            line += f" HBM {(2 * c['FETCH_SIZE']['avg_per_launch'] + c['WRITE_SIZE']['avg_per_launch']) * 1024 / 1e6:9.2f} MB"
        if "SQ_VALU_MFMA_BUSY_CYCLES" in c and c.get("SQ_BUSY_CU_CYCLES", {}).get("avg_per_launch"):
            line += f"  mfma busy {c['SQ_VALU_MFMA_BUSY_CYCLES']['avg_per_launch'] / c['SQ_BUSY_CU_CYCLES']['avg_per_launch'] / 4:.3f}"
        for k in ("SQ_INSTS_VALU_MFMA_MOPS_F32", "SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_WAVE_CYCLES", "SQ_ACTIVE_INST_VALU"):
            if k in c:
                line += f"  {k}={c[k]['avg_per_launch']:.4g}"
        print(line)
