"""What two workgroups of a CU share (round-3 VERDICT 3): the logged rollout at B = 16 384 (one workgroup per CU), 32 768 (two)
and 65 536 (four), with and without the placeholder wave, under `rocprofv3 --pmc` -- LDS, texture-address / vector-L1 and
vector-memory-write counters per launch, normalised per UAV tick.

    for pass in 1 2 3 4 5 6 7; do timeout 300 rocprofv3 --pmc $(python3 tools/half_chip_counters.py --counters $pass) --kernel-trace --output-format csv \\
        -d gpurun_out/pmc_half_$pass -o pmc -- python3 tools/half_chip_counters.py; done
    python3 tools/half_chip_counters.py --report gpurun_out/pmc_half_1 ... gpurun_out/pmc_half_5"""
import csv, glob, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
K, REPS = 2000, 4
PASSES = {"1": "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM_WR",
          "2": "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT",
          "3": "TA_BUSY_avr TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_WRITE_REQ_sum TCP_TA_TCP_STATE_READ_sum GRBM_GUI_ACTIVE",
          "4": "TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum",       # (TCP / TA: a few counters per pass, or the profile is refused)
          "6": "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum",
          "7": "TA_FLAT_WRITE_WAVEFRONTS_sum TA_BUSY_avr",
          "5": "SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_UNALIGNED_STALL SQ_INST_CYCLES_SALU SQ_INSTS_SALU SQ_WAVES SQ_WAIT_INST_ANY SQ_WAIT_ANY"}
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
                # grid in threads: tiles x (64 compute [+ 64 placeholder] + 64 store)
                wg, grid = int(r["Workgroup_Size"]), int(r["Grid_Size"])
                key = (grid // wg * 64, "placeholder" if wg == 192 else "plain")
                acc.setdefault(key, {}).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
                acc[key].setdefault("_dur_ns", []).append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    for (B, shape), c in sorted(acc.items()):
        out = {"B": B, "workgroup": shape, "launches": len(c["_dur_ns"])}
        tail = lambda v: v[len(v) // 2:]                                      # noqa: E731  (the later launches of each case)
        dur = tail(c.pop("_dur_ns"))
        out["us_per_tick"] = round(sum(dur) / len(dur) / 1e3 / K, 4)
        per = B / 64.0 * K                                                      # compute-wave ticks per launch
        for name, v in sorted(c.items()):
            v = tail(v)
            out[name + "_per_wave_tick"] = round(sum(v) / len(v) / per, 3)
        print(json.dumps(out))
    sys.exit(0)
sys.path[:0] = [ROOT, os.path.join(ROOT, "uav-autonomous-control_amd")]
import torch
from bench import missions
from uav_ac.fleet import Engine
eng = Engine("cuda:0")
for B in (16384, 32768, 65536):
    plan = eng.plan(missions(B, 8, 0, B), 3.0, 0.01)
    log = torch.empty((K, 13, B), dtype=torch.float64, device="cuda:0")
    for idle in ((0, 1) if B < 40000 else (0,)):             # (the placeholder wave is never used above 512 tiles)
        eng.ctx.set_option("idle_waves", idle)
        fleet = eng.fleet(plan, from_plan=True)
        for _ in range(REPS):
            fleet.rollout(K, state_log=log)
        torch.cuda.synchronize()
    eng.ctx.set_option("idle_waves", -1)
    del log, plan
