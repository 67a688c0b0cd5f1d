python -m pytest tests -m gpu -q 2>&1 | grep -E "^E  |^FAILED|passed|failed" > gpurun_out/r2d_gputest.txt
python bench.py --steps 50 > gpurun_out/r2d_bench1.json 2> gpurun_out/r2d_bench1.err
