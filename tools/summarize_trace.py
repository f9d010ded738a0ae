#!/usr/bin/env python3
"""Turn a rocprofv3 kernel trace (`*_kernel_trace.csv`) into the two tables kept under profiles/:
  <out>_kernel_stats.csv   per kernel and grid size: calls, total / average / min / max duration
  <out>_per_dispatch.csv   the rollout launches of every bench step in launch order (launch 1..10 per step), so that a
                           launch-position effect (round 1: the first two launches after planning ran 30 % slower) shows
    python3 tools/summarize_trace.py <kernel_trace.csv> <out-prefix>
"""
import csv
import sys
from collections import OrderedDict, defaultdict


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
    steps, cur = [], None
    for r in rows:
        k = short(r["Kernel_Name"])
        if k.startswith("row_counts_kernel"):
            cur = []
            steps.append(cur)
        elif k.startswith("control_rollout_kernel") and cur is not None and int(r["Grid_Size_X"]) >= 65536:
            cur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    full = [s for s in steps if len(s) == 10]
    with open(prefix + "_per_dispatch.csv", "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(["step"] + [f"launch_{i + 1}_us" for i in range(10)])
        for i, s in enumerate(full):
            w.writerow([i] + [round(x, 1) for x in s])
        if full:
            w.writerow(["mean"] + [round(sum(s[j] for s in full) / len(full), 1) for j in range(10)])
    by_pos = defaultdict(list)
    for s in full:
        for j, x in enumerate(s):
            by_pos[j].append(x)
    if full:
        means = [sum(by_pos[j]) / len(by_pos[j]) for j in range(10)]
        print("rollout launch position means (us):", [round(m, 1) for m in means], "spread %.1f %%" % (100 * (max(means) / min(means) - 1)))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
