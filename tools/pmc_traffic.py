#!/usr/bin/env python3
"""HBM traffic of the bench's kernels from the PMC counters, per launch, written to profiles/hbm_traffic.json together
with the fingerprint of the kernel sources it was measured on (bench.py only quotes `roofline.traffic` while that
fingerprint still matches what is in the tree).

Run ON THE GPU BOX from the repository root:

    python3 tools/pmc_traffic.py                  # three rocprofv3 passes (WRITE_SIZE, FETCH_SIZE, SQ_INSTS_VALU), ~1 min

Collected and corrected as MI355X_MICROARCH.md (HBM / rocprofv3 PMC slots) prescribes: one counter per pass (FETCH_SIZE
and WRITE_SIZE do not fit one TCC pass), counter passes carry no trace domain besides --kernel-trace, the profiled
program follows `--` directly (python3, no shell hop); rocprofv3 reports both counters in KiB; on gfx950 FETCH_SIZE
counts a 128-byte request as 64 bytes for wide coalesced reads => doubled (an upper estimate for narrow reads, which
is all the plan-fed rollout has); WRITE_SIZE is exact for 16-byte-per-lane stores and taken as is.
A third pass counts SQ_INSTS_VALU (vector-ALU wave-instructions executed, all waves of the launch): the measured
replacement of round 1's hand count of fp64 instructions per tick.
"""
import csv
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import B_PER_GPU, CHUNK, F, planning_source_sha, rollout_source_sha  # noqa: E402

OUT = os.path.join(ROOT, "gpurun_out")
TAG = os.environ.get("UAVAC_PROFILE_TAG", "r05")          # round tag of the files written
KERNELS = {"control_rollout": "control_rollout_kernel", "minsnap_sample": "minsnap_sample_stream_kernel",
           "minsnap_solve": "minsnap_solve_tw_kernel"}      # (the two-ended solve is the default since round 5)


def collect(counter):
    d = os.path.join(OUT, f"pmc_{TAG}_{counter}")
    subprocess.run(["rm", "-rf", d], check=True)
    env = dict(os.environ, TMPDIR="/tmp")
    cmd = ["rocprofv3", "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", d, "-o", "pmc", "--",
           "python3", os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-config4",
           "--no-extras"]
    subprocess.run(cmd, check=True, env=env, cwd=ROOT, stdout=subprocess.DEVNULL, timeout=280)
    path = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)[0]
    per = {}
    with open(path) as fh:
        for r in csv.DictReader(fh):
            if r["Counter_Name"] != counter:
                continue
            for key, frag in KERNELS.items():
                if frag in r["Kernel_Name"]:
                    # the bench's own launches only (the 8-mission oracle check uses tiny grids)
                    big = int(r["Grid_Size"]) >= B_PER_GPU
                    if big:
                        name = r["Kernel_Name"].replace("void ", "", 1).replace("(anonymous namespace)::", "").split("(")[0]
                        # (the library reports the launched variant under the very name rocprofv3 prints, all nine arguments)
                        per.setdefault((key, name), []).append(float(r["Counter_Value"]))
    return per


def main():
    os.makedirs(OUT, exist_ok=True)
    w, f, v = collect("WRITE_SIZE"), collect("FETCH_SIZE"), collect("SQ_INSTS_VALU")
    rec = {"source": "tools/pmc_traffic.py: rocprofv3 --pmc WRITE_SIZE / --pmc FETCH_SIZE (separate passes, --kernel-trace only) "
                     "on `bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-config4 --no-extras`",
           "units": "WRITE_SIZE / FETCH_SIZE are KiB; bytes = value * 1024; FETCH_SIZE doubled (gfx950 tallies 128-byte requests at 64 B); SQ_INSTS_VALU is a plain count (the CSV column name says KiB for all rows)",
           "rollout_source_sha": rollout_source_sha(),
           "planning_source_sha": planning_source_sha(),
           # (no .git on the GPU box: the caller passes the commit the snapshot was taken from, `UAVAC_GIT_HEAD=$(git rev-parse --short HEAD)`)
           "git": os.environ.get("UAVAC_GIT_HEAD") or subprocess.run(["git", "rev-parse", "--short", "HEAD"], cwd=ROOT, capture_output=True, text=True).stdout.strip() or None}
    rows = [["kernel", "counter", "dispatches", "mean_value_KiB", "min_KiB", "max_KiB"]]
    for (key, name), vals in sorted(w.items()):
        fv = f.get((key, name), [])
        wb = 1024.0 * sum(vals) / len(vals)
        fb = 2.0 * 1024.0 * sum(fv) / len(fv) if fv else None
        rows.append([name, "WRITE_SIZE", len(vals), sum(vals) / len(vals), min(vals), max(vals)])
        if fv:
            rows.append([name, "FETCH_SIZE", len(fv), sum(fv) / len(fv), min(fv), max(fv)])
        rec[f"{key}_write_bytes_per_launch"] = wb
        rec[f"{key}_fetch_bytes_per_launch_x2"] = fb
        rec[f"{key}_bytes_per_launch"] = wb + (fb or 0.0)
        vv = v.get((key, name), [])
        if vv:
            rows.append([name, "SQ_INSTS_VALU", len(vv), sum(vv) / len(vv), min(vv), max(vv)])
            rec[f"{key}_valu_wave_insts_per_launch"] = sum(vv) / len(vv)
        if key == "control_rollout":
            rec["kernel"] = name
            if vv:       # per UAV tick: one compute wave per 64 UAVs carries the arithmetic (the store wave's share is included)
                rec["control_rollout_valu_insts_per_uav_tick"] = sum(vv) / len(vv) / (B_PER_GPU / 64.0) / CHUNK
            rec["control_rollout_algorithmic_bytes_per_launch"] = float(B_PER_GPU) * CHUNK * (104.0 + 88.0 / F)
    os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
    with open(os.path.join(ROOT, "profiles", "hbm_traffic.json"), "w") as fh:
        json.dump(rec, fh, indent=1)
    with open(os.path.join(ROOT, "profiles", f"{TAG}_pmc_summary.csv"), "w", newline="") as fh:
        csv.writer(fh, quoting=csv.QUOTE_NONNUMERIC).writerows(rows)
    # copies for the round trip through gpurun_out/ (profiles/ itself is not merged back from the GPU box)
    subprocess.run(["cp", os.path.join(ROOT, "profiles", "hbm_traffic.json"), os.path.join(ROOT, "profiles", f"{TAG}_pmc_summary.csv"), OUT])
    print(json.dumps(rec, indent=1))


if __name__ == "__main__":
    main()
