"""Where an item's cycles go in the chunk-streaming sampler: a DIAGNOSTIC build (`-DUAVAC_DIAG_STAMPS=1`,
tools/ab/libuavac_sstamps.so) stamps s_memtime after the evaluation, after the wait for the carry and after the write-out.
    UAVAC_LIB=$PWD/tools/ab/libuavac_sstamps.so python3 tools/sampler_stamps_probe.py [WxG ...]"""
import ctypes as C, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "uav-autonomous-control_amd")]
import numpy as np
import torch
from bench import missions
from uav_ac import _native as nat
from uav_ac.fleet import Engine
lib = nat.lib()
lib.uavac_sampler_diag_read.argtypes = [C.c_void_p, C.c_int]
eng = Engine("cuda:0")
plan = eng.plan(missions(65536, 12, 0, 65536), 3.0, 0.01)
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for shape in (sys.argv[1:] or ["4x1", "2x1", "8x1", "4x2"]):
    W, G = (int(v) for v in shape.split("x"))
    eng.ctx.set_option("sampler_waves", W)
    eng.ctx.set_option("sampler_group", G)
    for _ in range(3):
        eng.sample(plan)
    a.record()
    for _ in range(3):
        eng.sample(plan)
    b.record()
    torch.cuda.synchronize()
    nwg = (65536 + G - 1) // G
    d = np.zeros(nwg * 4, dtype=np.int64)
    lib.uavac_sampler_diag_read(d.ctypes.data, nwg * 4)
    d = d.reshape(nwg, 4).sum(axis=0)
    n = float(d[3])
    print(json.dumps({"waves_x_group": shape, "ms": round(a.elapsed_time(b) / 3, 4), "items_per_launch": int(n),
                      "cycles_per_item": {"evaluate_stage_headings": round(d[0] / n, 1), "wait_for_carry": round(d[1] / n, 1),
                                          "handover_yaw_writeout": round(d[2] / n, 1)}}), flush=True)
