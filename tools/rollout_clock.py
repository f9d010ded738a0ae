"""Effective shader clock of the logged rollout at several batch sizes: GRBM_GUI_ACTIVE / 8 XCDs / kernel duration
(MI355X_MICROARCH.md, DVFS give-back), on launches of 4 000 ticks (3-6 ms).  Is the half-full chip (B = 32 768: 0.93 us per
tick against 0.77 us at B <= 16 384) running the same instruction stream at a lower clock?
    rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pmc_rclock -o pmc -- python3 tools/rollout_clock.py
    python3 tools/rollout_clock.py --report gpurun_out/pmc_rclock"""
import csv, glob, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
K = 4000
if len(sys.argv) > 2 and sys.argv[1] == "--report":
    f = glob.glob(os.path.join(sys.argv[2], "**", "*counter_collection.csv"), recursive=True)[0]
    acc = {}
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != "GRBM_GUI_ACTIVE" or "control_rollout_kernel" not in r["Kernel_Name"]:
            continue
        dur = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        key = (int(r["Grid_Size"]), int(r["Workgroup_Size"]), "plan" if "true, false, true>" in r["Kernel_Name"] else "rows")
        acc.setdefault(key, []).append((float(r["Counter_Value"]) / 8.0 / dur, dur / 1e3))
    for k, v in sorted(acc.items()):
        v = v[len(v) // 2:]
        ghz = sum(x for x, _ in v) / len(v)
        us = sum(d for _, d in v) / len(v)
        print(json.dumps({"threads": k[0], "workgroup": k[1], "feed": k[2], "launches": len(v), "effective_GHz": round(ghz, 3),
                          "us_per_tick": round(us / K, 4), "cycles_per_tick": round(us / K * ghz * 1e3, 1)}))
    sys.exit(0)
sys.path[:0] = [ROOT, os.path.join(ROOT, "uav-autonomous-control_amd")]
import torch
from bench import missions
from uav_ac.fleet import Engine
eng = Engine("cuda:0")
eng.ctx.set_option("idle_waves", 0)
for B in (4096, 16384, 24576, 32768, 49152, 65536):
    plan = eng.plan(missions(B, 8, 0, B), 3.0, 0.01)
    log = torch.empty((K, 13, B), dtype=torch.float64, device="cuda:0")
    for feed in (False, True):
        fleet = eng.fleet(plan, from_plan=feed)
        for _ in range(8):
            fleet.rollout(K, state_log=log)
        torch.cuda.synchronize()
    del log, plan
