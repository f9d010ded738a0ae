timeout 900 python -m pytest tests/test_gpu_round3.py -x -q -m gpu -k "full_size_planning" 2>&1 | grep -E "passed|failed|^E" | head -8
