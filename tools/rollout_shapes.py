"""Logged rollout (1 000 ticks, state log) over batch sizes x feed (plan / rows) x log pitch x LDS padding (= cap on the
workgroups a CU takes) x hand-over point.  One JSON line per case: ms per 1 000 ticks, G control-steps/s.
    python3 tools/rollout_shapes.py [--quick] > gpurun_out/rollout_shapes.jsonl
"""
import argparse, itertools, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "uav-autonomous-control_amd")]
import torch
from bench import missions
from uav_ac.fleet import Engine

ap = argparse.ArgumentParser()
ap.add_argument("--sizes", default="4096,8192,16384,32768,49152,65534,65536,131072,262144")
ap.add_argument("--pitch", default="0,16,32,64,272", help="doubles added to roundup16(B)")
ap.add_argument("--pads", default="0,8192,31744,60416")
ap.add_argument("--late", default="-1")
ap.add_argument("--store-waves", default="-1", help="store waves per workgroup (-1: the launcher's choice)")
ap.add_argument("--cw", default="-1", help="compute waves per workgroup (-1: the launcher's choice)")
ap.add_argument("--idle", default="-1", help="placeholder wave between compute and store wave (-1: the launcher's choice)")
ap.add_argument("--alternate", type=int, default=1, help="rounds over all cases of one batch size (interleaved A/B)")
ap.add_argument("--m", type=int, default=8)
ap.add_argument("--feeds", default="plan,rows")
args = ap.parse_args()
eng = Engine("cuda:0")
K = 1000
for B in [int(x) for x in args.sizes.split(",")]:
    plan = eng.plan(missions(B, args.m, 0, B), 3.0, 0.01)
    base_pitch = -(-B // 16) * 16
    cases = list(itertools.product(args.feeds.split(","), [int(x) for x in args.pitch.split(",")],
                                   [int(x) for x in args.pads.split(",")], [int(x) for x in args.late.split(",")],
                                   [int(x) for x in args.store_waves.split(",")], [int(x) for x in args.cw.split(",")],
                                   [int(x) for x in args.idle.split(",")]))
    for rnd, (feed, dp, pad, late, nsw, cw, idle) in itertools.product(range(args.alternate), cases):
        pitch = base_pitch + dp
        if dp < 0:
            pitch = B                                  # the dense layout of round 2
        log = torch.empty((K, 13, pitch), dtype=torch.float64, device="cuda:0")
        eng.ctx.set_option("lds_pad", pad)
        eng.ctx.set_option("late_handover", late)
        if nsw != -1 or cw != -1:     # options of a throw-away build of round 3 (row-splitting store waves, 2 + 2 wave workgroups:
            eng.ctx.set_option("store_waves", nsw)         # measured useless, profiles/r03_rollout_shapes_*.jsonl, and removed)
            eng.ctx.set_option("compute_waves", cw)
        eng.ctx.set_option("idle_waves", idle)
        fleet = eng.fleet(plan, from_plan=(feed == "plan"))
        for _ in range(3):
            fleet.rollout(K, state_log=log, log_pitch=pitch)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        best = 1e9
        for rep in range(3):
            fleet.reset()
            a.record()
            for _ in range(5):
                fleet.rollout(K, state_log=log, log_pitch=pitch)
            b.record()
            torch.cuda.synchronize()
            best = min(best, a.elapsed_time(b) / 5)
        print(json.dumps({"B": B, "m": args.m, "feed": feed, "pitch": pitch, "pitch_extra": dp, "lds_pad": pad, "late": late, "store_waves": nsw, "compute_waves": cw, "idle_waves": idle, "round": rnd, "kernel": eng.ctx.last_rollout_kernel()[22:30],
                          "ms_per_1000_ticks": round(best, 4), "G_steps_per_s": round(B * K / best / 1e6, 2)}), flush=True)
        del log, fleet
    del plan
eng.ctx.set_option("lds_pad", 0)
eng.ctx.set_option("late_handover", -1)
eng.ctx.set_option("idle_waves", -1)
