#!/bin/bash
# Runs on the GPU box (gpurun): the default bench line, rocprofv3 kernel-trace statistics of the same command, the
# HBM-traffic / MFMA counter passes (separate --pmc runs, program directly after `--`), the cfg-3 and cfg-4 workloads
# and the non-headline configurations.  Everything lands in gpurun_out/<tag>/; tools/publish_profiles.py copies the
# summaries worth keeping to profiles/<tag>_*.
# usage: bash tools/collect_profiles.sh <tag>
TAG=${1:-r2}
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
# the probes are git-ignored binaries: build them here (a fresh checkout published "No such file" as a result)
for P in split_bias_probe split_mfma_probe; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -I normalizingflows.jl_amd/csrc -I include tools/probe/$P.hip -o tools/probe/$P > "$OUT/build_$P.log" 2>&1
done
python3 bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"
# the driver's own command line (value = the 20 steps after 5 warm-ups; *_sustained_clock beside it)
python3 bench.py --gpus 1 --steps 20 --warmup 5 > "$OUT/bench_driver_cmd.json" 2>> "$OUT/bench_default.err"
# per-step time WITH the shader clock next to every step, three phases (fresh, after 3 s idle, after 50 ms idle)
python3 tools/bench_ramp.py 200 3 > "$OUT/step_ramp_clocks.txt" 2>&1
python3 bench.py --workload cfg3 --steps 50 --no-cpu-baseline > "$OUT/bench_cfg3.json" 2>> "$OUT/bench_default.err"
python3 bench.py --workload cfg4 --steps 5 --warmup 2 > "$OUT/bench_cfg4_1gpu_262144.json" 2>> "$OUT/bench_default.err"
python3 bench.py --workload cfg4 --batch 32768 --steps 10 --warmup 3 > "$OUT/bench_cfg4_shard_32768.json" 2>> "$OUT/bench_default.err"
# (planar / radial BEFORE the configuration table: on one box of round 4 everything measured after the table's Float64 rows --
# 114 ms steps of scalar fp64 -- ran 20-100 % slower for minutes, radial 397 instead of 330 us; a fresh box gave the usual numbers)
python3 tools/bench_simple.py > "$OUT/simple.txt" 2>&1
# (the Float64 rows of the table run LAST, at the end of this script, for the same reason)
python3 tools/bench_configs.py --steps 30 --only cfg1,cfg2_,cfg2b,cfg3,cfg4,cfg2c,cfg5,gen,fwd > "$OUT/configs.txt" 2>&1
rocprofv3 --kernel-trace --stats -d "$OUT/kt_cfg2" -o cfg2 --output-format csv -- python3 bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-kernel-events > "$OUT/kt_cfg2.log" 2>&1
rocprofv3 --kernel-trace --stats -d "$OUT/kt_cfg3" -o cfg3 --output-format csv -- python3 bench.py --workload cfg3 --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-events > "$OUT/kt_cfg3.log" 2>&1
rocprofv3 --kernel-trace --stats -d "$OUT/kt_cfg4" -o cfg4 --output-format csv -- python3 bench.py --workload cfg4 --batch 32768 --steps 5 --warmup 2 --no-kernel-events > "$OUT/kt_cfg4.log" 2>&1
rocprofv3 --kernel-trace --stats -d "$OUT/kt_cfg1" -o cfg1 --output-format csv -- python3 tools/bench_configs.py --only cfg1 --steps 100 > "$OUT/kt_cfg1.log" 2>&1
rocprofv3 --kernel-trace --stats -d "$OUT/kt_cfg5" -o cfg5 --output-format csv -- python3 tools/bench_configs.py --only cfg5 --steps 10 > "$OUT/kt_cfg5.log" 2>&1
rocprofv3 --kernel-trace --stats -d "$OUT/kt_simple" -o simple --output-format csv -- python3 tools/bench_simple.py > "$OUT/kt_simple.log" 2>&1
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C -d "$OUT/pmc_cfg2_$C" -o pmc --output-format csv -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-events > "$OUT/pmc_cfg2_$C.log" 2>&1
  rocprofv3 --pmc $C -d "$OUT/pmc_cfg3_$C" -o pmc --output-format csv -- python3 bench.py --workload cfg3 --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-events > "$OUT/pmc_cfg3_$C.log" 2>&1
  rocprofv3 --pmc $C -d "$OUT/pmc_simple_$C" -o pmc --output-format csv -- python3 tools/bench_simple.py 262144 > "$OUT/pmc_simple_$C.log" 2>&1
done
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 -d "$OUT/pmc_cfg2_mfma" -o pmc --output-format csv -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-events > "$OUT/pmc_cfg2_mfma.log" 2>&1
# (round 4: part of the GEMMs run on the bf16 matrix cores -- their MFMA ops are counted by a counter of their own)
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_MFMA -d "$OUT/pmc_cfg2_mfma_bf16" -o pmc --output-format csv -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-events > "$OUT/pmc_cfg2_mfma_bf16.log" 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 -d "$OUT/pmc_cfg3_mfma" -o pmc --output-format csv -- python3 bench.py --workload cfg3 --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-events > "$OUT/pmc_cfg3_mfma.log" 2>&1
# (round 5: the NSF kernels' output-layer GEMMs run on the bf16 matrix cores)
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_MFMA -d "$OUT/pmc_cfg3_mfma_bf16" -o pmc --output-format csv -- python3 bench.py --workload cfg3 --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-events > "$OUT/pmc_cfg3_mfma_bf16.log" 2>&1
# A/B of the cfg-3 kernels on this box: the fp32-MFMA reverse kernel / forward chain of rounds 2-4
NF_RQS_BWD_FP32=1 python3 bench.py --workload cfg3 --steps 50 --no-cpu-baseline > "$OUT/bench_cfg3_bwd_fp32_mfma.json" 2>> "$OUT/bench_default.err"
NF_RQS_FWD_FP32=1 python3 bench.py --workload cfg3 --steps 50 --no-cpu-baseline > "$OUT/bench_cfg3_fwd_fp32_mfma.json" 2>> "$OUT/bench_default.err"
# the general layer-by-layer path for the shapes nf_deep.hip fuses since round 5, and the scalar Float64 kernels
NF_DEEP_OFF=1 python3 tools/bench_configs.py --steps 30 --only gen > "$OUT/configs_deep_off.txt" 2>&1
# kernel-trace statistics of the deep / Float64 configurations (which kernels run)
rocprofv3 --kernel-trace --stats -d "$OUT/kt_gen" -o gen --output-format csv -- python3 tools/bench_configs.py --only gen,f64 --steps 30 > "$OUT/kt_gen.log" 2>&1
# arithmetic A/B of the named parity arrays (tools/parity_ab.py): default and fp32 MFMA chains
python3 tools/parity_ab.py "$OUT/parity_ab_default.json" > "$OUT/parity_ab_default.txt" 2>&1
NF_FWD_FP32=1 NF_BWD_FP32=1 NF_WIDE_FP32=1 python3 tools/parity_ab.py "$OUT/parity_ab_fp32_mfma.json" > "$OUT/parity_ab_fp32_mfma.txt" 2>&1
[ -x ./tools/probe/split_bias_probe ] && ./tools/probe/split_bias_probe > "$OUT/split_bias_probe.txt" 2>&1
[ -x ./tools/probe/split_mfma_probe ] && ./tools/probe/split_mfma_probe > "$OUT/split_mfma_probe.txt" 2>&1
# issue-side counters of the cfg-2 kernels (vector instructions per launch next to the matrix pipe's busy time)
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU -d "$OUT/pmc_cfg2_valu" -o pmc --output-format csv -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-events > "$OUT/pmc_cfg2_valu.log" 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU -d "$OUT/pmc_cfg3_valu" -o pmc --output-format csv -- python3 bench.py --workload cfg3 --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-events > "$OUT/pmc_cfg3_valu.log" 2>&1
# issue-side counters of the planar / radial step (missing 5): where k_simple_step's time goes
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_LDS SQ_INSTS_SALU -d "$OUT/pmc_simple_issue" -o pmc --output-format csv -- python3 tools/bench_simple.py 262144 > "$OUT/pmc_simple_issue.log" 2>&1
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES SQ_WAIT_INST_LDS -d "$OUT/pmc_simple_issue2" -o pmc --output-format csv -- python3 tools/bench_simple.py 262144 > "$OUT/pmc_simple_issue2.log" 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_F32 -d "$OUT/pmc_simple_mfma" -o pmc --output-format csv -- python3 tools/bench_simple.py 1048576 > "$OUT/pmc_simple_mfma.log" 2>&1
# the three forms of the cfg-2 step: nf_elbo_step (default), split calls, hipGraph replay
python3 bench.py --no-cpu-baseline --split-calls > "$OUT/bench_default_split_calls.json" 2>> "$OUT/bench_default.err"
python3 bench.py --no-cpu-baseline --graph > "$OUT/bench_default_graph.json" 2>> "$OUT/bench_default.err"
# A/B of the cfg-2 reverse kernel on this box: one wavefront per tile (k_affine_bwd_stashed) instead of the pair kernel
NF_BWD_NO_PAIR=1 python3 bench.py --no-cpu-baseline > "$OUT/bench_default_one_wave_per_tile.json" 2>> "$OUT/bench_default.err"
# round 4 A/B switches: fp32 MFMAs everywhere (the round-3 arithmetic), the bf16 six-term products also in the stashing forward
NF_BWD_FP32=1 NF_FWD_FP32=1 python3 bench.py --no-cpu-baseline > "$OUT/bench_default_fp32_mfma_everywhere.json" 2>> "$OUT/bench_default.err"
NF_FWD_B6_STASH=1 python3 bench.py --no-cpu-baseline > "$OUT/bench_default_b6_stashing_forward.json" 2>> "$OUT/bench_default.err"
# the pair kernel with fp32 dW GEMMs (the producer's dX GEMMs alone on the bf16 cores), and the wide path on fp32 MFMAs
NF_BWD_DW_FP32=1 python3 bench.py --no-cpu-baseline > "$OUT/bench_default_pair_dw_fp32.json" 2>> "$OUT/bench_default.err"
NF_WIDE_FP32=1 python3 bench.py --workload cfg4 --batch 32768 --steps 10 --warmup 3 > "$OUT/bench_cfg4_shard_32768_fp32_mfma.json" 2>> "$OUT/bench_default.err"
NF_WIDE_FP32=1 python3 bench.py --workload cfg4 --steps 5 --warmup 2 > "$OUT/bench_cfg4_1gpu_262144_fp32_mfma.json" 2>> "$OUT/bench_default.err"
NF_FWD_FP32=1 python3 tools/bench_configs.py --only cfg5,fwd --steps 10 > "$OUT/configs_fp32_forward.txt" 2>&1
NF_PLANAR_NO_MFMA=1 NF_RADIAL_NO_LANE=1 python3 tools/bench_simple.py > "$OUT/simple_no_mfma.txt" 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 -d "$OUT/pmc_cfg4_mfma" -o pmc --output-format csv -- python3 bench.py --workload cfg4 --batch 32768 --steps 3 --warmup 1 --no-kernel-events > "$OUT/pmc_cfg4_mfma.log" 2>&1
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C -d "$OUT/pmc_cfg4_$C" -o pmc --output-format csv -- python3 bench.py --workload cfg4 --batch 32768 --steps 3 --warmup 1 --no-kernel-events > "$OUT/pmc_cfg4_$C.log" 2>&1
done
python3 tools/bench_configs.py --steps 30 --only f64 >> "$OUT/configs.txt" 2>&1
NF_G64_NO_F64_MFMA=1 python3 tools/bench_configs.py --steps 30 --only f64 > "$OUT/configs_f64_scalar.txt" 2>&1
# the parity table of THIS build on THIS box (a collection without it is not published): the whole GPU suite;
# tests/conftest.py writes gpurun_out/parity_measured.json at the end of the session
timeout 1800 python3 -m pytest tests -m gpu -q > "$OUT/gpu_suite.txt" 2>&1
tail -3 "$OUT/gpu_suite.txt"
cp gpurun_out/parity_measured.json "$OUT/parity_measured.json" 2>/dev/null
python3 tools/kernel_resources.py > "$OUT/kernel_resources.txt" 2>&1
# keep the merge-back small: per-dispatch traces can be large
find "$OUT" -name "*kernel_trace.csv" -size +4M -delete
find "$OUT" -name "*.csv" | head -60
du -sh "$OUT"
