#!/bin/bash
# round-6 experiment 1: cfg 2 with the stash in chunks small enough for the Infinity Cache
cd /root/repo
mkdir -p gpurun_out/r6
python3 bench.py --steps 50 --warmup 20 > gpurun_out/r6/chunk_base.json 2>gpurun_out/r6/chunk_base.err
for mb in 48 96 144 192 240 400; do
  NF_AFFINE_STASH_MAX_MB=$mb python3 bench.py --steps 50 --warmup 20 > gpurun_out/r6/chunk_$mb.json 2>gpurun_out/r6/chunk_$mb.err
done
python3 bench.py --steps 50 --warmup 20 > gpurun_out/r6/chunk_base2.json 2>gpurun_out/r6/chunk_base2.err
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r6/chunk_*.json')):
    for l in open(f):
        l=l.strip()
        if l.startswith('{'):
            d=json.loads(l)
            print(f, d['ms_per_step'], d.get('ms_per_step_sustained_clock'), {k:(v['avg_ms'],v['launches_per_step']) for k,v in d.get('kernels',{}).items()})
PY
