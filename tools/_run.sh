timeout 300 python tools/sampler_stream_ab.py 8 65536 12 1x1,4x1,3x1,2x1,3x2 2>/dev/null | grep waves_x
