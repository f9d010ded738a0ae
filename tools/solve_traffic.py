"""HBM traffic of the coefficient solve alone from the PMC counters (WRITE_SIZE, FETCH_SIZE: separate passes, --kernel-trace only, units
and the gfx950 x2 on FETCH_SIZE as in tools/pmc_traffic.py), against its algorithmic bytes 24(m+1) + 8m + 192m per mission.
    python3 tools/solve_traffic.py            -> one JSON line per (B, m, solve_order)
    python3 tools/solve_traffic.py --run B m order   (what the profiled child runs)"""
import csv, glob, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "uav-autonomous-control_amd")]
if len(sys.argv) > 1 and sys.argv[1] == "--run":
    import torch
    from bench import missions
    from uav_ac.fleet import Engine
    B, m, order = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    eng = Engine("cuda:0")
    eng.ctx.set_option("solve_order", order)
    plan = eng.plan(missions(B, m, 0, B), 3.0, 0.01)
    for _ in range(6):
        eng.solve(plan)
    torch.cuda.synchronize()
    sys.exit(0)
OUT = os.path.join(ROOT, "gpurun_out")
for B, m in ((65536, 12), (65536, 20), (65536, 8)):
    for order in (1, 0):
        rec = {"B": B, "m": m, "solve_order": order, "algorithmic_bytes": B * (24 * (m + 1) + 8 * m + 192 * m)}
        for counter in ("WRITE_SIZE", "FETCH_SIZE"):
            d = os.path.join(OUT, f"pmc_solve_{counter}")
            subprocess.run(["rm", "-rf", d], check=True)
            subprocess.run(["rocprofv3", "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", d, "-o", "pmc", "--", "python3",
                            os.path.abspath(__file__), "--run", str(B), str(m), str(order)], check=True, env=dict(os.environ, TMPDIR="/tmp"),
                           cwd=ROOT, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=280)
            vals, durs, name = [], [], None
            for r in csv.DictReader(open(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)[0])):
                if r["Counter_Name"] == counter and "minsnap_solve_" in r["Kernel_Name"] and "_kernel" in r["Kernel_Name"] and "row_counts" not in r["Kernel_Name"]:
                    vals.append(float(r["Counter_Value"]) * 1024.0 * (2.0 if counter == "FETCH_SIZE" else 1.0))
                    durs.append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
                    name = r["Kernel_Name"].replace("void ", "", 1).replace("(anonymous namespace)::", "").split("(")[0]
            vals, durs = vals[len(vals) // 2:], durs[len(durs) // 2:]
            rec["kernel"] = name
            rec["write_bytes" if counter == "WRITE_SIZE" else "fetch_bytes_x2"] = sum(vals) / len(vals)
            rec["kernel_us_profiled"] = round(sum(durs) / len(durs) / 1e3, 1)
            subprocess.run(["rm", "-rf", d], check=True)
        rec["counter_bytes"] = rec["write_bytes"] + rec["fetch_bytes_x2"]
        rec["counter_over_algorithmic"] = round(rec["counter_bytes"] / rec["algorithmic_bytes"], 3)
        print(json.dumps(rec), flush=True)
