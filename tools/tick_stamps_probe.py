"""Where a logged tick's cycles go, by batch size (round-3 VERDICT 3): a DIAGNOSTIC build of the rollout
(`-DUAVAC_DIAG_STAMPS=1`, tools/ab/libuavac_stamps.so) stamps s_memtime around the per-tick barrier of the compute wave and of the
store wave and around the store wave's 13 stores, and adds the differences up per workgroup.
    UAVAC_LIB=$PWD/tools/ab/libuavac_stamps.so python3 tools/tick_stamps_probe.py"""
import ctypes as C, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "uav-autonomous-control_amd")]
import numpy as np
import torch
from bench import missions
from uav_ac import _native as nat
from uav_ac.fleet import Engine
K = 2000
lib = nat.lib()
lib.uavac_diag_read.argtypes = [C.c_void_p, C.c_int]
lib.uavac_diag_read2.argtypes = [C.c_void_p, C.c_int]
FEEDS = [f == "plan" for f in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["plan", "rows"])]
eng = Engine("cuda:0")
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
def med(x):
    return round(float(np.median(x)), 1)


for B, idle in ((8192, 0), (16384, 0), (32768, 0), (32768, 1), (49152, 0), (65536, 0)):
    eng.ctx.set_option("idle_waves", idle)
    plan = eng.plan(missions(B, 8, 0, B), 3.0, 0.01)
    log = torch.empty((K, 13, B), dtype=torch.float64, device="cuda:0")
    for feed in FEEDS:
        fleet = eng.fleet(plan, from_plan=feed)
        for _ in range(3):
            fleet.rollout(K, state_log=log)
        a.record()
        fleet.rollout(K, state_log=log)
        b.record()
        torch.cuda.synchronize()
        n = min(B // 64, 8192)
        d = np.zeros(n * 8, dtype=np.int64)
        assert lib.uavac_diag_read(d.ctypes.data, n * 8) == 0
        d = d.reshape(n, 8).astype(np.float64) / K
        d2 = np.zeros(n * 4, dtype=np.int64)
        assert lib.uavac_diag_read2(d2.ctypes.data, n * 4) == 0
        d2 = d2.reshape(n, 4).astype(np.float64) / K
        print(json.dumps({"B": B, "feed": "plan" if feed else "rows", "placeholder_wave": idle,
                          "us_per_tick": round(a.elapsed_time(b) / K * 1e3, 4),
                          "cycles_per_tick": {"compute_wave_total": med(d[:, 0]), "compute_wave_at_barrier": med(d[:, 1]),
                                              "store_wave_total": med(d[:, 4]), "store_wave_reads_and_13_stores": med(d[:, 3]),
                                              "between_ticks_and_outer_loop": med(d[:, 5]), "inner_tick_arithmetic": med(d[:, 6] + d[:, 7]),
                                              "slab_writes": med(d[:, 2]),
                                              "outer_target_row": med(d2[:, 0]), "outer_scalar_constants": med(d2[:, 1]),
                                              "outer_controllers": med(d2[:, 2]), "outer_advance_and_loads": med(d2[:, 3])}}), flush=True)
        del fleet
    del log, plan
eng.ctx.set_option("idle_waves", -1)
