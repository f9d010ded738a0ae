timeout 300 python tools/sampler_stream_ab.py 6 65536 12 4x1,4x4,8x2,8x4,8x8,16x4,16x8 2>/dev/null | grep waves_x | grep -v planning
