set -x
timeout 600 python -m pytest tests/test_gpu_round3.py -x -q -m gpu -k "streaming" 2>&1 | tail -15
timeout 600 python -m pytest tests/test_gpu_planner.py -x -q -m gpu 2>&1 | tail -5
timeout 900 python tools/sampler_stream_ab.py 8 65536 12 > gpurun_out/r4_stream_ab1.jsonl 2>&1
tail -20 gpurun_out/r4_stream_ab1.jsonl
