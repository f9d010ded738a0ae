timeout 900 python -m pytest tests/test_gpu_round3.py -x -q -m gpu -k "streaming" 2>&1 | grep -E "passed|failed|Error|assert" | tail -3
timeout 600 python tools/sampler_stream_ab.py 6 65536 12 1x1,4x1x1,4x1x2,4x1x3,4x1x6,2x1x2,2x1x4,2x1x11,8x1x2,8x1x3 2>/dev/null | grep waves_x | grep -v planning
