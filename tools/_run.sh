timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | grep -E "passed|failed|Error|assert" | tail -4
timeout 600 python tools/plan_vs_rows.py 2>/dev/null | grep "B=" | head -8
