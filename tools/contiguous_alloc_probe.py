"""Does a physically contiguous row buffer (hipExtMallocWithFlags, hipDeviceMallocContiguous) always put the sampler in
its fast mode?  Plain hipMalloc buffers and contiguous ones, alternating, the sampler timed on each through the C ABI."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "uav-autonomous-control_amd")]
import torch
from bench import missions
from uav_ac.fleet import Engine
hip = C.CDLL("libamdhip64.so")
hip.hipExtMallocWithFlags.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_uint]
hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
hip.hipFree.argtypes = [C.c_void_p]
eng = Engine("cuda:0")
plan = eng.plan(missions(65536, 12, 0, 65536), 3.0, 0.01)
nbytes = plan.traj.numel() * 8
P = lambda t: C.c_void_p(t.data_ptr())
def timed(ptr, n=20):
    def go():
        eng._bind_stream()
        eng.ctx.call("uavac_minsnap_sample_dev", P(plan.coeffs), P(plan.times), P(plan.seg_rows), P(plan.row_offsets), plan.B,
                     plan.m, plan.dt, ptr)
    for _ in range(4): go()
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): go()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
for _ in range(10): eng.sample(plan)
bufs = []
for i in range(10):
    p = C.c_void_p()
    contiguous = i % 2 == 1
    rc = hip.hipExtMallocWithFlags(C.byref(p), nbytes, 0x4) if contiguous else hip.hipMalloc(C.byref(p), nbytes)
    if rc != 0:
        print(f"{'contiguous' if contiguous else 'plain'} allocation {i}: error {rc}"); continue
    bufs.append(p)
    print(f"{'contiguous' if contiguous else 'plain     '} {p.value:#x}: sampler {timed(p):.3f} ms")
for p in bufs: hip.hipFree(p)

# ---- the rollout's log stream into plain / contiguous buffers (its pattern is contiguous across the chip per tick)
fleet = eng.fleet(plan)
K = 1000
logbytes = K * 13 * plan.B * 8
def roll(ptr, n=8):
    def go():
        eng._bind_stream()
        eng.ctx.call("uavac_control_rollout_plan_dev", C.byref(fleet.vehicle), P(plan.coeffs), P(plan.seg_rows), P(plan.row_offsets),
                     C.c_void_p(0), P(plan.first_yaw), plan.m, float(plan.dt), P(fleet.state), P(fleet.istate), plan.B, K, ptr,
                     C.c_void_p(0), C.c_void_p(0), 0)
    fleet.reset(); fleet.rollout(2000)
    for _ in range(2): go()
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): go()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
bufs = []
for i in range(6):
    p = C.c_void_p()
    contiguous = i % 2 == 1
    rc = hip.hipExtMallocWithFlags(C.byref(p), logbytes, 0x4) if contiguous else hip.hipMalloc(C.byref(p), logbytes)
    if rc != 0:
        print(f"log {'contiguous' if contiguous else 'plain'} allocation {i}: error {rc}"); continue
    bufs.append(p)
    print(f"log {'contiguous' if contiguous else 'plain     '} {p.value:#x}: rollout {roll(p):.3f} ms per 1000 ticks")
tl = torch.empty((K, 13, plan.B), dtype=torch.float64, device="cuda:0")
print(f"log torch      {tl.data_ptr():#x}: rollout {roll(C.c_void_p(tl.data_ptr())):.3f} ms per 1000 ticks")
for p in bufs: hip.hipFree(p)
