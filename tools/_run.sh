timeout 600 python -m pytest tests/test_gpu_planner.py -x -q -m gpu -k "parked" 2>&1 | grep -E "passed|failed|Error" | tail -3
timeout 600 python tools/solve_time.py 2>/dev/null | grep "B=" > gpurun_out/r4_solve_park.txt; cat gpurun_out/r4_solve_park.txt
