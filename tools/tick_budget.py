"""The rollout's tick below a full chip, from an UN-INSTRUMENTED build (round-4 VERDICT 2): per batch size, microseconds per
logged tick (HIP events), the shader clock the chip held meanwhile (one extra wavefront on a side stream stamping s_memtime /
s_memrealtime: uavac_clock_probe_dev), hence cycles per tick -- and which kernel flew.

    python3 tools/tick_budget.py [m] [sizes ...]                      -> one JSON line per batch size
    rocprofv3 --pmc <counters> --kernel-trace --output-format csv -d D -o pmc -- python3 tools/tick_budget.py --counters-run
    python3 tools/tick_budget.py --report D1 D2 ...                   -> SQ counters per compute-wave tick for the same sizes

The counter passes (tools/collect_tick_budget.sh) split a tick's cycles into vector-ALU issue, scalar, LDS, waits and the rest."""
import csv, glob, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
K = 1000
SIZES = (4096, 8192, 16384, 24576, 32768, 35237, 49152, 65536)
PASSES = {"1": "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS",
          "2": "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_MISC",
          "3": "GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_SMEM SQ_IFETCH SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"}
if len(sys.argv) > 2 and sys.argv[1] == "--counters":
    print(PASSES[sys.argv[2]])
    sys.exit(0)
if len(sys.argv) > 2 and sys.argv[1] == "--report":
    acc = {}
    for d in sys.argv[2:]:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                if "control_rollout_kernel" not in r["Kernel_Name"]:
                    continue
                wg, grid = int(r["Workgroup_Size"]), int(r["Grid_Size"])
                key = (grid // wg, r["Kernel_Name"].split("control_rollout_kernel")[1].split("(")[0])
                acc.setdefault(key, {}).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
                acc[key].setdefault("_dur_ns", []).append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    for (tiles, kern), c in sorted(acc.items()):
        tail = lambda v: v[len(v) // 2:]                                      # noqa: E731
        dur = tail(c.pop("_dur_ns"))
        out = {"tiles": tiles, "kernel": kern, "launches": len(dur), "us_per_tick_profiled": round(sum(dur) / len(dur) / 1e3 / K, 4)}
        per = tiles * K                                                         # compute-wave ticks per launch
        for name, v in sorted(c.items()):
            v = tail(v)
            out[name] = round(sum(v) / len(v) / per, 3)
        print(json.dumps(out))
    sys.exit(0)
sys.path[:0] = [ROOT, os.path.join(ROOT, "uav-autonomous-control_amd")]
import numpy as np
import torch
from bench import missions
from uav_ac.fleet import Engine
counters_run = len(sys.argv) > 1 and sys.argv[1] == "--counters-run"
args = [a for a in sys.argv[1:] if not a.startswith("--")]
m = int(args[0]) if args else 8
sizes = [int(a) for a in args[1:]] or (SIZES if not counters_run else (16384, 32768, 35237, 65536))
eng = Engine("cuda:0")
side = torch.cuda.Stream(device=eng.device)
for B in sizes:
    plan = eng.plan(missions(B, m, 0, B), 3.0, 0.01)
    pitch = -(-B // 16) * 16
    log = torch.empty((K, 13, pitch), dtype=torch.float64, device="cuda:0")
    fleet = eng.fleet(plan, from_plan=True)
    for _ in range(4):
        fleet.rollout(K, state_log=log, log_pitch=pitch)
    torch.cuda.synchronize()
    if counters_run:
        del log, plan, fleet
        continue
    res = []
    for rep in range(5):
        fleet.reset()
        fleet.rollout(K, state_log=log, log_pitch=pitch)          # (the launch behind a reset is not what is timed)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        side.wait_event(a)
        # a window that ends inside the eight launches whatever the size: 8 x 0.75 ms at least
        stamps = eng.clock_probe_begin(5000, stream=side)
        for _ in range(8):
            fleet.rollout(K, state_log=log, log_pitch=pitch)
        b.record()
        torch.cuda.synchronize()
        res.append((a.elapsed_time(b) * 1e3 / (8 * K), eng.clock_probe_ghz(stamps)))
    us = float(np.median([r[0] for r in res]))
    ghz = float(np.median([r[1] for r in res]))
    print(json.dumps({"B": B, "m": m, "tiles": -(-B // 64), "kernel": eng.ctx.last_rollout_kernel().split("kernel")[1], "us_per_tick": round(us, 4),
                      "clock_ghz": round(ghz, 3), "cycles_per_tick": round(us * ghz * 1e3, 1), "G_steps_per_s": round(B / us / 1e3, 2),
                      "us_all": [round(r[0], 4) for r in res], "ghz_all": [round(r[1], 3) for r in res]}), flush=True)
    del log, plan, fleet
