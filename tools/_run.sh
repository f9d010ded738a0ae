for i in 1 2 3; do timeout 900 python -m pytest tests/test_gpu_round2.py -x -q -m gpu -k "row_buffer" 2>&1 | grep -E "passed|failed|^E" | head -3; done
python bench.py --no-config4 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/bench_now.json
