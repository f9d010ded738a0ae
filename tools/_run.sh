python bench.py --no-config4 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/bench_now.json
