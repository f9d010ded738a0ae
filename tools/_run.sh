timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | grep -E "passed|failed|error" | tail -3
python bench.py 2>/dev/null | tail -1 > gpurun_out/bench_now.json
