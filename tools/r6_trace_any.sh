#!/bin/bash
# run a tools/trace_*.py script against a library variant built by tools/ab_build.py: r6_trace_any.sh <variant> <script> <out-name> [ENV=.. ...]
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
v=$1; s=$2; o=$3; shift 3
cp normalizingflows.jl_amd/libnfhip.so /tmp/libnfhip_shipped.so
cp normalizingflows.jl_amd/ab/$v.so normalizingflows.jl_amd/libnfhip.so
env "$@" python3 tools/$s > gpurun_out/$o.txt 2>&1
cp /tmp/libnfhip_shipped.so normalizingflows.jl_amd/libnfhip.so
cat gpurun_out/$o.txt
