#!/bin/bash
# Runs on the GPU box (gpurun): the default bench line, rocprofv3 kernel-trace statistics of the
# same command, HBM-traffic counters (separate --pmc passes) and the cfg-4 workload.  Everything
# lands in gpurun_out/<tag>/; the summaries worth keeping are copied to profiles/ by hand.
# usage: bash tools/collect_profiles.sh <tag>
TAG=${1:-r1}
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
python3 bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"
python3 bench.py --workload cfg4 --steps 5 --warmup 2 > "$OUT/bench_cfg4_1gpu_262144.json" 2>> "$OUT/bench_default.err"
python3 bench.py --workload cfg4 --batch 32768 --steps 10 --warmup 3 > "$OUT/bench_cfg4_shard_32768.json" 2>> "$OUT/bench_default.err"
python3 tools/bench_configs.py --steps 20 > "$OUT/configs.txt" 2>&1
rocprofv3 --kernel-trace --stats -d "$OUT/kt_cfg2" -o cfg2 --output-format csv -- python3 bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-kernel-events > "$OUT/kt_cfg2.log" 2>&1
rocprofv3 --kernel-trace --stats -d "$OUT/kt_cfg4" -o cfg4 --output-format csv -- python3 bench.py --workload cfg4 --batch 32768 --steps 5 --warmup 2 --no-kernel-events > "$OUT/kt_cfg4.log" 2>&1
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C -d "$OUT/pmc_$C" -o pmc --output-format csv -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-events > "$OUT/pmc_$C.log" 2>&1
done
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 -d "$OUT/pmc_mfma" -o pmc --output-format csv -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-events > "$OUT/pmc_mfma.log" 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 -d "$OUT/pmc_mfma_cfg4" -o pmc --output-format csv -- python3 bench.py --workload cfg4 --batch 32768 --steps 3 --warmup 1 --no-kernel-events > "$OUT/pmc_mfma_cfg4.log" 2>&1
find "$OUT" -name "*.csv" | head -50
# keep the merge-back small: per-dispatch traces can be large
find "$OUT" -name "*kernel_trace.csv" -size +8M -delete
du -sh "$OUT"
