#!/bin/bash
# Runs on the GPU box (gpurun): rocprofv3 kernel-trace statistics and the HBM / MFMA counter passes (separate --pmc
# runs, program directly after `--`) for ONE of the non-headline configurations of tools/bench_configs.py.
# usage: bash tools/collect_config_profiles.sh <tag> <cfg1|cfg3|cfg5|...> [steps]
TAG=${1:-r2}
CFG=${2:-cfg3}
STEPS=${3:-20}
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
python3 tools/bench_configs.py --only $CFG --steps $STEPS > "$OUT/configs_$CFG.txt" 2>&1
rocprofv3 --kernel-trace --stats -d "$OUT/kt_$CFG" -o $CFG --output-format csv -- python3 tools/bench_configs.py --only $CFG --steps $STEPS > "$OUT/kt_$CFG.log" 2>&1
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C -d "$OUT/pmc_${CFG}_$C" -o pmc --output-format csv -- python3 tools/bench_configs.py --only $CFG --steps 3 > "$OUT/pmc_${CFG}_$C.log" 2>&1
done
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 -d "$OUT/pmc_${CFG}_mfma" -o pmc --output-format csv -- python3 tools/bench_configs.py --only $CFG --steps 3 > "$OUT/pmc_${CFG}_mfma.log" 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU -d "$OUT/pmc_${CFG}_valu" -o pmc --output-format csv -- python3 tools/bench_configs.py --only $CFG --steps 3 > "$OUT/pmc_${CFG}_valu.log" 2>&1
find "$OUT" -name "*kernel_trace.csv" -size +8M -delete
find "$OUT" -name "*.csv" | head -40
du -sh "$OUT"
