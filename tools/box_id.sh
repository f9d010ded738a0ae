#!/bin/bash
# One line per box: identity, partition modes, firmware, and how fast this GPU takes the sampler's write pattern.
rocm-smi --showuniqueid --showmemorypartition --showcomputepartition --showvbios --showmemvendor --showmaxpower 2>/dev/null | grep -E "Unique ID|artition|VBIOS|vendor|Max Graphics" | sed 's/GPU\[0\]\s*: //' | tr '\n' ';'
echo
python3 - <<'PY'
import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", ".")
sys.path[:0] = [ROOT, os.path.join(ROOT, "uav-autonomous-control_amd")]
import torch
from bench import missions
from uav_ac.fleet import Engine
eng = Engine("cuda:0")
plan = eng.plan(missions(65536, 12, 0, 65536), 3.0, 0.01)
for _ in range(20): eng.sample(plan)
torch.cuda.synchronize()
a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(100): eng.sample(plan)
b.record(); torch.cuda.synchronize()
print(f"sampler (rows only) {a.elapsed_time(b)/100:.3f} ms; free/total HBM GiB {torch.cuda.mem_get_info()[0]/2**30:.0f}/{torch.cuda.mem_get_info()[1]/2**30:.0f}")
PY
