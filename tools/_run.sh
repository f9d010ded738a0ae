timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | grep -E "passed|failed|error" | tail -3
python tools/solve_time.py 2>/dev/null | grep "B=" > gpurun_out/r04_solve_park.txt
