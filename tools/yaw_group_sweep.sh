rocm-smi --showuniqueid 2>/dev/null | grep -i "unique id:" | head -1
for yg in 4 1 8 16 4; do UAVAC_YAW_GROUP=$yg python3 tools/sampler_time.py 2>/dev/null | tail -2 | sed "s/^/yg=$yg /"; done
