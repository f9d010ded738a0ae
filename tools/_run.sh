timeout 900 python -m pytest tests/test_gpu_control.py -x -q -m gpu 2>&1 | grep -E "passed|failed|Error|assert" | tail -3
timeout 600 python tools/plan_vs_rows.py 2>/dev/null | grep "B=" | head -9
