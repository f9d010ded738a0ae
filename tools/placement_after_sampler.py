"""Does the slow-down of the first rollout launches after the planning kernels (tools/first_launch_bisect.py) come from
where the hardware places the compute wave and the store wave of each 128-thread workgroup?  A census kernel with the
rollout's launch shape (tools/wave_census.hip) is put where launch 1, 2 or 3 of a bench step would run and reports,
per SIMD, how many first waves ("compute") and second waves ("store") landed there.  Balanced = every SIMD holds one
of each.  Also: which empty "aligner" launches restore the balanced placement, how 4- and 8-wave workgroups are placed
in the same situations, and the raw (HW_ID, XCC_ID) words of every arm (gpurun_out/placement_raw.npz) for offline study
of role-assignment rules.

    hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/wave_census.hip -o tools/libwave_census.so
    python3 tools/placement_after_sampler.py
"""
import ctypes as C
import json
import os
import sys
from collections import Counter

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "uav-autonomous-control_amd")]
import numpy as np
import torch
from bench import missions
from uav_ac.fleet import Engine

B, M, CH = 65536, 12, 1000
LDS1 = 8 * (2 * 13 * 64 + 40 * 64)           # per 64 UAVs
dev = "cuda:0"
lib = C.CDLL(os.path.join(ROOT, "tools", "libwave_census.so"))
lib.wave_census.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]
lib.noop.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
eng = Engine(dev)
plan = eng.plan(missions(B, M, 0, B), 3.0, 0.01, dense_yaw=True)
fleet = eng.fleet(plan)
log = torch.empty((CH, 13, B), dtype=torch.float64, device=dev)
out = torch.zeros((2048, 2), dtype=torch.int32, device=dev)
arrived = torch.zeros((1,), dtype=torch.int32, device=dev)
raw = {}


def stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def census(tag, waves_per_wg=2):
    wgs = 2048 // waves_per_wg
    rc = lib.wave_census(stream(), C.c_void_p(out.data_ptr()), C.c_void_p(arrived.data_ptr()), wgs, 64 * waves_per_wg,
                         LDS1 * waves_per_wg // 2)
    assert rc == 0
    torch.cuda.synchronize()
    h = out.cpu().numpy().astype(np.uint32)
    raw[tag] = h.copy()
    hw, xcc = h[:, 0], h[:, 1] & 0xF
    simd, cu, sh, se = (hw >> 4) & 3, (hw >> 8) & 0xF, (hw >> 12) & 1, (hw >> 13) & 7
    key = (xcc.astype(np.int64) << 16) | (se << 12) | (sh << 11) | (cu << 4) | simd
    role = (np.arange(len(key)) % waves_per_wg) >= waves_per_wg // 2          # second half of a workgroup's waves = store
    per = {}
    for k, r in zip(key.tolist(), role.tolist()):
        per.setdefault(k, [0, 0])[int(r)] += 1
    hist = Counter((c, s) for c, s in per.values())
    kk = key.reshape(-1, waves_per_wg)
    distinct = int(np.sum([len(set(row.tolist())) == waves_per_wg for row in kk]))
    slots = Counter((hw & 0xF).tolist())
    return {"waves_per_wg": waves_per_wg, "simds_used": len(per), "arrived": int(arrived.item()),
            "simds_by_(compute,store)": {f"{c}+{s}": n for (c, s), n in sorted(hist.items())},
            "workgroups_with_all_waves_on_distinct_simds": distinct, "wave_slots": dict(sorted(slots.items()))}


def plan_kernels():
    eng.solve(plan)
    eng.sample(plan)


def history():
    plan_kernels(); fleet.reset()
    for _ in range(4):
        fleet.rollout(CH, state_log=log)
    torch.cuda.synchronize()


def noop(wgs, threads, lds=0):
    assert lib.noop(stream(), wgs, threads, lds) == 0


arms = {
    "cold: census only": lambda: None,
    "B-like: reset": lambda: fleet.reset(),
    "A-like: solve+sample, reset": lambda: (plan_kernels(), fleet.reset()),
    "A-like: solve+sample, reset, 1 rollout": lambda: (plan_kernels(), fleet.reset(), fleet.rollout(CH, state_log=log)),
    "A-like: solve+sample, reset, 2 rollouts": lambda: (plan_kernels(), fleet.reset(), fleet.rollout(CH, state_log=log), fleet.rollout(CH, state_log=log)),
    "H-like: solve+sample, torch fill 0.68 GB, reset": lambda: (plan_kernels(), torch.zeros_like(plan.yaw), fleet.reset()),
    "E-like: solve+sample, reset, 1-tick rollout": lambda: (plan_kernels(), fleet.reset(), fleet.rollout(1)),
    "sampler only": lambda: eng.sample(plan),
    "solve only": lambda: eng.solve(plan),
}
for g, t in ((256, 256), (1024, 256), (4096, 256), (256, 1024), (1024, 1024), (2048, 128), (1024, 128), (4096, 64), (1024, 512)):
    arms[f"solve+sample, reset, noop<<<{g},{t}>>>"] = (lambda g=g, t=t: (plan_kernels(), fleet.reset(), noop(g, t)))
arms["solve+sample, reset, noop<<<1024,128,lds>>>"] = lambda: (plan_kernels(), fleet.reset(), noop(1024, 128, LDS1))
arms["solve+sample, reset, 2 x noop<<<1024,128,lds>>>"] = lambda: (plan_kernels(), fleet.reset(), noop(1024, 128, LDS1), noop(1024, 128, LDS1))
arms["solve+sample, reset, 3 x noop<<<1024,128,lds>>>"] = lambda: (plan_kernels(), fleet.reset(), [noop(1024, 128, LDS1) for _ in range(3)])
arms["E-like + noop<<<1024,256>>>"] = lambda: (plan_kernels(), fleet.reset(), fleet.rollout(1), noop(1024, 256))
arms["solve only + noop<<<1024,256>>>"] = lambda: (eng.solve(plan), noop(1024, 256))

for rep in range(2):
    for name, pre in arms.items():
        for wpw in ((2, 4, 8) if rep == 0 and not name.startswith("solve+sample, reset, noop") else (2,)):
            history()
            pre()
            print(json.dumps({"arm": name, "rep": rep, **census(f"{name}|{wpw}|{rep}", wpw)}), flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
np.savez_compressed(os.path.join(ROOT, "gpurun_out", "placement_raw.npz"), **{k.replace("/", "_"): v for k, v in raw.items()})
