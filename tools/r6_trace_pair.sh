#!/bin/bash
# clock stamps of k_affine_bwd_pair (tiles + phase boundary) from the trace variant built by tools/ab_build.py trace -DNF_KERNEL_TRACE=1
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
cp normalizingflows.jl_amd/libnfhip.so /tmp/libnfhip_shipped.so
cp normalizingflows.jl_amd/ab/${1:-trace}.so normalizingflows.jl_amd/libnfhip.so
python3 tools/trace_bwd_pair.py > gpurun_out/${2:-trace_bwd_pair}.txt 2>&1
N=262144 python3 tools/trace_bwd_pair.py > gpurun_out/${2:-trace_bwd_pair}_n262144.txt 2>&1
cp /tmp/libnfhip_shipped.so normalizingflows.jl_amd/libnfhip.so
cat gpurun_out/${2:-trace_bwd_pair}.txt
