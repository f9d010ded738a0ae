"""CPU baseline leg of bench.py: the C oracle (a scalar port of the reference's per-UAV algorithm,
oracle/uavac_oracle.c) timed on the host, one thread, on a bounded sample of the bench workload.
TEST INFRASTRUCTURE -- measured next to the GPU number, never part of it."""
from __future__ import annotations

import time

from . import c_oracle as co
from .minsnap_oracle import synthetic_missions


def effective_cpus():
    """CPUs this process may really use: scheduler affinity, capped by the cgroup CPU quota (a container can see
    256 CPUs and own 8 of them).  -> (n, description)."""
    import math
    import os
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    note = f"os.cpu_count()={os.cpu_count()}, affinity={n}"
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as fh:                      # cgroup v2
            q, period = fh.read().split()[:2]
            if q != "max":
                quota = float(q) / float(period)
    except OSError:
        try:                                                             # cgroup v1
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as fq, open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as fp:
                q, period = float(fq.read()), float(fp.read())
                if q > 0:
                    quota = q / period
        except OSError:
            pass
    if quota is not None:
        note += f", cgroup quota={quota:g} CPUs"
        n = max(1, min(n, int(math.ceil(quota))))
    return n, note


def cpu_model():
    """Model string of the host CPU (stated next to the core count)."""
    try:
        with open("/proc/cpuinfo") as fh:
            for line in fh:
                if line.lower().startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    import platform
    return platform.processor() or "unknown"


def blas_threads():
    """The thread settings a NumPy / BLAS baseline would run under.  The C oracle itself links no BLAS (its
    solve is its own scalar LU), so they do not change its rate; they are stated because the reference's NumPy path does."""
    import os
    return {k: os.environ.get(k) for k in ("OPENBLAS_NUM_THREADS", "OMP_NUM_THREADS", "MKL_NUM_THREADS")}


def _all_cores(wps, V, segments, ticks, velocity, dt, budget_s):
    """The same scalar work on every CPU this process owns at once: POSIX threads inside the C library, each
    planning and flying whole missions with its own buffers (no Python in the loop, no fork from a process that
    holds a GPU context)."""
    n_thr, note = effective_cpus()
    done, elapsed = co.bench_threads(wps, velocity, dt, ticks, n_thr, budget_s, V)
    return {"value": done * ticks / elapsed, "unit": "UAV control-steps/s", "cores": n_thr,
            "sample": f"{done} missions (plan + {ticks} ticks each) on {n_thr} threads, {elapsed:.1f} s wall ({note})"}


def run(segments: int, ticks: int, velocity: float, dt: float, budget_s: float = 12.0, max_missions: int = 100000):
    wps = synthetic_missions(min(max_missions, 4096), segments)     # recycled if the budget outlasts them
    V = co.Vehicle.default()
    co.plan(wps[0], velocity, dt)                       # warm-up (page in, build if needed)
    n = 0
    t_plan = t_roll = 0.0
    rows = 0
    t_start = time.perf_counter()
    while n < max_missions and (time.perf_counter() - t_start) < budget_s:
        t0 = time.perf_counter()
        traj, _, _ = co.plan(wps[n % len(wps)], velocity, dt)
        t1 = time.perf_counter()
        state, istate = co.initial_state(traj[0, 0:3], V)
        co.rollout(traj, state, istate, ticks, V, log_state=True, log_cmd=False)
        t2 = time.perf_counter()
        t_plan += t1 - t0
        t_roll += t2 - t1
        rows += len(traj)
        n += 1
    total = t_plan + t_roll
    all_cores = _all_cores(wps, V, segments, ticks, velocity, dt, budget_s * 0.5)
    return {
        "all_cores": all_cores,
        "value": n * ticks / total,
        "unit": "UAV control-steps/s",
        "cores": 1,
        "kind": "port",
        "cpu_model": cpu_model(),
        "blas_threads": blas_threads(),
        "blas_note": "the C oracle links no BLAS (own scalar LU): these settings do not change its rate",
        "sample": f"{n} missions of the same generator: plan ({segments} segments, {rows} rows) + {ticks} ticks each, "
                  f"scalar C oracle (gcc -O2), {total:.1f} s of CPU",
        "control_only_steps_per_s": n * ticks / t_roll,
        "minsnap_segments_per_s": n * segments / t_plan,
    }
