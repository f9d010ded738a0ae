timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | grep -E "passed|failed|error" | tail -3
