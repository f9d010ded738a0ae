#!/usr/bin/env python3
"""Turn a rocprofv3 kernel trace (`*_kernel_trace.csv`) into the two tables kept under profiles/:
  <out>_kernel_stats.csv   per kernel and grid size: calls, total / average / min / max duration
  <out>_per_dispatch.csv   the rollout launches of every bench step in launch order (launch 1..10 per step), so that a
                           launch-position effect (round 1: the first two launches after planning ran 30 % slower) shows
    python3 tools/summarize_trace.py <kernel_trace.csv> <out-prefix>
"""
import csv
import sys
from collections import OrderedDict


def short(name):
    name = name.replace("void ", "", 1).replace("(anonymous namespace)::", "")
    depth, out = 0, []
    for ch in name:                      # cut the argument list, keep template arguments
        if ch == "(" and depth == 0:
            break
        depth += ch == "<"
        depth -= ch == ">"
        out.append(ch)
    return "".join(out)


def main(path, prefix):
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    groups = OrderedDict()
    for r in rows:
        d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        groups.setdefault((short(r["Kernel_Name"]), int(r["Grid_Size_X"]), int(r["Workgroup_Size_X"])), []).append(d)
    total = sum(sum(v) for v in groups.values())
    with open(prefix + "_kernel_stats.csv", "w", newline="") as fh:
        w = csv.writer(fh, quoting=csv.QUOTE_NONNUMERIC)
        w.writerow(["kernel", "grid_size_x", "workgroup_size_x", "calls", "total_ns", "average_ns", "percent", "min_ns", "max_ns"])
        for (k, g, wg), v in sorted(groups.items(), key=lambda kv: -sum(kv[1])):
            w.writerow([k, g, wg, len(v), sum(v), round(sum(v) / len(v), 1), round(100.0 * sum(v) / total, 4), min(v), max(v)])
    # per-dispatch table: steps are delimited by the row-count kernel of the planning chain
    # (a step that starts after the GPU has idled -- process start, a host synchronisation -- runs its first launches at
    # ramping clocks: it is listed with the idle time in front of it and left out of the means)
    WINDOW = 20e6                                  # ns: how busy was the GPU in the 20 ms before the step?
    spans = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows]
    steps, cur = [], None
    for i, r in enumerate(rows):
        k = short(r["Kernel_Name"])
        if k.startswith("row_counts_kernel"):
            t0 = spans[i][0]
            busy = sum(min(e, t0) - max(s, t0 - WINDOW) for s, e in spans[:i] if e > t0 - WINDOW)
            cur = {"idle_ms": (WINDOW - busy) / 1e6, "us": []}
            steps.append(cur)
        elif k.startswith("control_rollout_kernel") and cur is not None and int(r["Grid_Size_X"]) >= 65536:
            cur["us"].append((spans[i][1] - spans[i][0]) / 1e3)
    full = [s for s in steps if len(s["us"]) == 10]
    steady = [s for s in full if s["idle_ms"] < 2.0]
    with open(prefix + "_per_dispatch.csv", "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(["step", "gpu_idle_ms_in_the_20_ms_before"] + [f"launch_{i + 1}_us" for i in range(10)])
        for i, s in enumerate(full):
            w.writerow([i, round(s["idle_ms"], 2)] + [round(x, 1) for x in s["us"]])
        if steady:
            w.writerow(["mean of steps that follow a busy GPU (< 2 ms idle)", ""] +
                       [round(sum(s["us"][j] for s in steady) / len(steady), 1) for j in range(10)])
    if steady:
        means = [sum(s["us"][j] for s in steady) / len(steady) for j in range(10)]
        print(f"rollout launch position means over {len(steady)} back-to-back steps (us):", [round(m, 1) for m in means],
              "spread %.1f %%" % (100 * (max(means) / min(means) - 1)))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
