"""Effective shader clock of the planning and rollout kernels on this box: GRBM_GUI_ACTIVE / 8 XCDs / kernel duration
(MI355X_MICROARCH.md, DVFS give-back).  Run as
    rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pmc_clock -o pmc -- python3 tools/sampler_clock.py
then `python3 tools/sampler_clock.py --report gpurun_out/pmc_clock`."""
import csv, glob, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 2 and sys.argv[1] == "--report":
    f = glob.glob(os.path.join(sys.argv[2], "**", "*counter_collection.csv"), recursive=True)[0]
    acc = {}
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != "GRBM_GUI_ACTIVE" or int(r["Grid_Size"]) < 65536:
            continue
        name = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
        dur = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        acc.setdefault(name, []).append((float(r["Counter_Value"]) / 8.0 / dur, dur / 1e3))
    for k, v in acc.items():
        v = v[len(v) // 2:]
        print(json.dumps({"kernel": k[:60], "launches": len(v), "effective_GHz": round(sum(x for x, _ in v) / len(v), 3),
                          "mean_us": round(sum(d for _, d in v) / len(v), 1)}))
    sys.exit(0)
sys.path[:0] = [ROOT, os.path.join(ROOT, "uav-autonomous-control_amd")]
import torch
from bench import missions
from uav_ac.fleet import Engine
eng = Engine("cuda:0")
plan = eng.plan(missions(65536, 12, 0, 65536), 3.0, 0.01)
fleet = eng.fleet(plan)
log = torch.empty((1000, 13, 65536), dtype=torch.float64, device="cuda:0")
for _ in range(6):
    eng.replan(plan); fleet.reset()
    for _ in range(10): fleet.rollout(1000, state_log=log)
torch.cuda.synchronize()
