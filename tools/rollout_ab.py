"""The bench's rollout launch (B = 65 536, m = 12, 1 000 logged ticks, plan-fed) through two builds of the library in ONE process,
alternating: the in-tree libuavac.so against another build of the same C ABI (default tools/ab/libuavac_r02.so = round 2's last
commit, `git archive 54af699 | make`).  Both see the same device buffers.   python3 tools/rollout_ab.py [other.so] [B] [plan|rows]
(`rows`: the row-fed kernel -- uavac_control_rollout_dev on the sampled trajectory -- whatever the batch size.)"""
import ctypes as C, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "uav-autonomous-control_amd")]
import torch
from bench import missions
from uav_ac import _native as nat                     # the Vehicle structure only: both libraries are loaded RTLD_LOCAL below
other = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "tools", "ab", "libuavac_r02.so")
B = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
FEED = sys.argv[3] if len(sys.argv) > 3 else "plan"
m, K = 12, 1000
torch.cuda.set_device(0)
_P = C.c_void_p
libs = {"tree": C.CDLL(os.path.join(ROOT, "uav-autonomous-control_amd", "lib", "libuavac.so"), mode=C.RTLD_LOCAL),
        "other": C.CDLL(other, mode=C.RTLD_LOCAL)}
ctxs = {}
for name, lib in libs.items():
    lib.uavac_create.argtypes = [C.POINTER(_P), C.c_int]
    lib.uavac_set_stream.argtypes = [_P, _P]
    lib.uavac_vehicle_default.argtypes = [C.POINTER(nat.Vehicle)]
    lib.uavac_minsnap_row_counts_dev.argtypes = [_P, _P, C.c_int, C.c_int, C.c_double, C.c_double, _P, _P, _P]
    lib.uavac_minsnap_solve_dev.argtypes = [_P, _P, _P, C.c_int, C.c_int, _P, _P]
    lib.uavac_minsnap_sample_derivs_dev.argtypes = [_P, _P, _P, _P, C.c_int, C.c_int, C.c_double, _P, _P, _P, _P, _P]
    lib.uavac_state_init_dev.argtypes = [_P, C.POINTER(nat.Vehicle), _P, C.c_int, C.c_int, _P, _P]
    lib.uavac_control_rollout_dev.argtypes = [_P, C.POINTER(nat.Vehicle), _P, _P, _P, _P, C.c_int, C.c_int, _P, _P, _P, C.c_int]
    lib.uavac_control_rollout_plan_dev.argtypes = [_P, C.POINTER(nat.Vehicle), _P, _P, _P, _P, _P, C.c_int, C.c_double, _P, _P,
                                                   C.c_int, C.c_int, _P, _P, _P, C.c_int]
    h = _P()
    assert lib.uavac_create(C.byref(h), 0) == 0
    assert lib.uavac_set_stream(h, _P(torch.cuda.current_stream().cuda_stream or None)) == 0
    ctxs[name] = h
    lib.uavac_set_option.argtypes = [_P, C.c_char_p, C.c_int]
    for opt in ("late_handover", "idle_waves", "coeff_dma"):          # AB_LATE_HANDOVER=1 ...: the same launch option in both libraries
        if os.environ.get("AB_" + opt.upper()) is not None:
            assert lib.uavac_set_option(h, opt.encode(), int(os.environ["AB_" + opt.upper()])) == 0
p = lambda t: _P(t.data_ptr())   # noqa: E731
kw = dict(device="cuda:0")
wp = torch.as_tensor(missions(B, m, 0, B), dtype=torch.float64).to("cuda:0").contiguous()


class plan:                                            # planned once, with the in-tree library
    times = torch.empty((B, m), dtype=torch.float64, **kw)
    seg_rows = torch.empty((B, m), dtype=torch.int32, **kw)
    row_offsets = torch.empty((B + 1,), dtype=torch.int64, **kw)
    coeffs = torch.empty((B, 8 * m, 3), dtype=torch.float64, **kw)
    first_yaw = torch.empty((B,), dtype=torch.float64, **kw)
    waypoints = wp


L, H = libs["tree"], ctxs["tree"]
assert L.uavac_minsnap_row_counts_dev(H, p(wp), B, m, 3.0, 0.01, p(plan.times), p(plan.seg_rows), p(plan.row_offsets)) == 0
assert L.uavac_minsnap_solve_dev(H, p(wp), p(plan.times), B, m, p(plan.coeffs), None) == 0
rows = torch.empty((int(plan.row_offsets[-1].item()), 11), dtype=torch.float64, **kw)
assert L.uavac_minsnap_sample_derivs_dev(H, p(plan.coeffs), p(plan.seg_rows), p(plan.row_offsets), B, m, 0.01, p(rows), None,
                                         p(plan.first_yaw), None, None) == 0
torch.cuda.synchronize()
if FEED != "rows":
    del rows
V = nat.Vehicle()
libs["tree"].uavac_vehicle_default(C.byref(V))
state = torch.empty((30, B), dtype=torch.float64, device="cuda:0")
istate = torch.empty((4, B), dtype=torch.int32, device="cuda:0")
log = torch.empty((K, 13, B), dtype=torch.float64, device="cuda:0")
pos = plan.waypoints[:, 0, :].contiguous()


def fly(name, launches):
    lib, h = libs[name], ctxs[name]
    assert lib.uavac_state_init_dev(h, C.byref(V), p(pos), B, 1, p(state), p(istate)) == 0
    for _ in range(launches):
        if FEED == "rows":
            assert lib.uavac_control_rollout_dev(h, C.byref(V), p(rows), p(plan.row_offsets), p(state), p(istate), B, K, p(log), None, None, 0) == 0
            continue
        assert lib.uavac_control_rollout_plan_dev(h, C.byref(V), p(plan.coeffs), p(plan.seg_rows), p(plan.row_offsets), None,
                                                  p(plan.first_yaw), m, 0.01, p(state), p(istate), B, K, p(log), None, None, 0) == 0


a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
res = {"tree": [], "other": []}
final = {}
for rnd in range(8):
    for name in ("tree", "other") if rnd % 2 == 0 else ("other", "tree"):
        fly(name, 2)
        torch.cuda.synchronize()
        a.record()
        fly(name, 10)
        b.record()
        torch.cuda.synchronize()
        res[name].append(a.elapsed_time(b) / 10)
        final[name] = state.clone()
print(json.dumps({"B": B, "feed": FEED, "other": os.path.basename(other), "same_bits": bool(torch.equal(final["tree"][:26], final["other"][:26])),
                  "ms_per_launch_tree": [round(x, 4) for x in res["tree"]], "ms_per_launch_other": [round(x, 4) for x in res["other"]],
                  "median_tree": round(sorted(res["tree"])[4], 4), "median_other": round(sorted(res["other"])[4], 4)}))
