"""Which counters tell a slow row buffer from a fast one?  Eight row buffers, the one-wave sampler timed on each (that tells the kinds),
then ONE more launch per buffer in order: under `rocprofv3 --pmc <counters> --kernel-trace` the last eight dispatches of
minsnap_sample_kernel are those launches.   python3 tools/scratch/tlb_by_buffer.py  [--report DIR]"""
import csv, glob, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "uav-autonomous-control_amd")]
if len(sys.argv) > 2 and sys.argv[1] == "--report":
    times = json.load(open(os.path.join(sys.argv[2], "times.json")))
    out = {}
    for f in glob.glob(os.path.join(sys.argv[2], "pass*", "**", "*counter_collection.csv"), recursive=True):
        rows = [r for r in csv.DictReader(open(f)) if "minsnap_sample_kernel" in r["Kernel_Name"] and int(r["Grid_Size"]) >= 65536 * 64]
        by = {}
        for r in rows:
            by.setdefault(r["Counter_Name"], []).append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
        for c, v in by.items():
            out[c] = [x for _, x in sorted(v)][-len(times):]
    print(json.dumps({"ms_per_buffer": times}))
    for c, v in sorted(out.items()):
        print(json.dumps({c: [round(x / 1e6, 3) for x in v], "unit": "1e6"}))
    sys.exit(0)
import torch
from bench import missions
from uav_ac.fleet import Engine
eng = Engine("cuda:0")
eng.ctx.set_option("sampler_waves", 1)
plan = eng.plan(missions(65536, 12, 0, 65536), 3.0, 0.01, placement_trials=1)
bufs = [plan.traj] + [torch.empty_like(plan.traj) for _ in range(7)]
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
times = []
for t in bufs:
    plan.traj = t
    eng.sample(plan); eng.sample(plan)
    a.record()
    for _ in range(3): eng.sample(plan)
    b.record(); torch.cuda.synchronize()
    times.append(round(a.elapsed_time(b) / 3, 4))
for t in bufs:
    plan.traj = t
    eng.sample(plan)
torch.cuda.synchronize()
os.makedirs(os.environ.get("TLB_OUT", "gpurun_out/tlb"), exist_ok=True)
json.dump(times, open(os.path.join(os.environ.get("TLB_OUT", "gpurun_out/tlb"), "times.json"), "w"))
print(times)
