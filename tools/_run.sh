timeout 900 python -m pytest tests/test_gpu_control.py -x -q -m gpu -k "second_wave or corner" 2>&1 | grep -E "passed|failed|Error|assert" | tail -5
