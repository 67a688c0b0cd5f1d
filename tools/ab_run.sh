#!/bin/bash
# Same-box A/B of library variants built by tools/ab_build.py (or copies of earlier builds) under normalizingflows.jl_amd/ab/:
#   tools/ab_run.sh <out-prefix> <repeats> "<bench.py args>" variantA variantB ...
# alternates the variants <repeats> times (each run: the variant copied over the box's libnfhip.so, one bench.py process) and
# prints, per run, the step time (as commanded / sustained) and the per-kernel averages of bench.py's own event brackets.
cd "$(dirname "$0")/.."
out=$1; rep=$2; bargs=$3; shift 3
mkdir -p "$(dirname "$out")"
cp normalizingflows.jl_amd/libnfhip.so /tmp/libnfhip_shipped.so
for i in $(seq 1 "$rep"); do
  for v in "$@"; do
    cp "normalizingflows.jl_amd/ab/$v.so" normalizingflows.jl_amd/libnfhip.so
    python3 bench.py $bargs > "${out}_${v}_$i.json" 2> "${out}_${v}_$i.err"
  done
done
cp /tmp/libnfhip_shipped.so normalizingflows.jl_amd/libnfhip.so
python3 - "$out" "$@" <<'PY'
import glob, json, sys
out, variants = sys.argv[1], sys.argv[2:]
for v in variants:
    for f in sorted(glob.glob(f"{out}_{v}_*.json")):
        for l in open(f):
            if l.startswith("{"):
                d = json.loads(l)
                ks = " ".join(f"{k}={x['avg_ms']*1e3:.1f}x{x['launches_per_step']:g}" for k, x in d.get("kernels", {}).items())
                print(f"{v:12s} step {d['ms_per_step']:.4f} sustained {d.get('ms_per_step_sustained_clock')} | {ks}")
PY
