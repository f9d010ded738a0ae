for B in 65536 32768 4096; do timeout 600 python tools/rollout_ab.py tools/ab/libuavac_wpe2.so $B 2>&1 | tail -3; done
