python -m pytest tests/test_gpu_parity.py tests/test_gpu_tape.py -m gpu -q -x 2>&1 | tail -2
for i in 1 2; do python bench.py --steps 100 --warmup 30 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg2', round(r['ms_per_step'],4), round(r['ms_per_step_sustained_clock'],4), round(r['roofline']['avg_launch_ms'],4), {k:v['avg_ms'] for k,v in r['kernels'].items()} if 'kernels' in r else '')"; done
python tools/bench_configs.py --only cfg5,fwd,cfg4 2>/dev/null | grep "^cfg\|^fwd" | cut -c1-330
