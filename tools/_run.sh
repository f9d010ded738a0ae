timeout 1500 python -m pytest tests/test_gpu_planner.py -x -q -m gpu 2>&1 | grep -E "passed|failed|^E" | head
python tools/solve_time.py 2>/dev/null | grep "B=" > gpurun_out/r04_solve_park.txt; head -4 gpurun_out/r04_solve_park.txt
