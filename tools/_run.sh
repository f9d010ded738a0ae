timeout 600 python tools/sampler_stream_ab.py 6 65536 12 1x1,4x1,2x1,8x1,4x2,8x2 2>/dev/null | grep waves_x
