#!/bin/bash
# Same-box A/B of library variants on rows of tools/bench_configs.py:  tools/ab_run_configs.sh <out-prefix> <repeats> "<--only list>" v1 v2 ...
cd "$(dirname "$0")/.."
out=$1; rep=$2; only=$3; shift 3
mkdir -p "$(dirname "$out")"
cp normalizingflows.jl_amd/libnfhip.so /tmp/libnfhip_shipped.so
for i in $(seq 1 "$rep"); do
  for v in "$@"; do
    cp "normalizingflows.jl_amd/ab/$v.so" normalizingflows.jl_amd/libnfhip.so
    python3 tools/bench_configs.py --only "$only" --steps 10 2>/dev/null | grep -v amdgpu.ids | sed "s/^/$v  /" | cut -c1-260
  done
done | tee "${out}.txt"
cp /tmp/libnfhip_shipped.so normalizingflows.jl_amd/libnfhip.so
