#!/bin/bash
# HBM-bound kernels at scale (planar / radial d=64, 10 layers, 1 M samples): timing table, rocprofv3 kernel
# statistics and HBM-traffic counters (separate --pmc passes).  usage: bash tools/collect_simple_profiles.sh <tag>
TAG=${1:-r1}
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
python3 tools/bench_simple.py > "$OUT/simple_hbm.txt" 2>&1
rocprofv3 --kernel-trace --stats -d "$OUT/kt_simple" -o simple --output-format csv -- python3 tools/bench_simple.py > "$OUT/kt_simple.log" 2>&1
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C -d "$OUT/pmcs_$C" -o pmc --output-format csv -- python3 tools/bench_simple.py 262144 > "$OUT/pmcs_$C.log" 2>&1
done
find "$OUT" -name "*kernel_trace.csv" -size +8M -delete
cat "$OUT/simple_hbm.txt"
head -8 "$OUT/kt_simple/simple_kernel_stats.csv"
