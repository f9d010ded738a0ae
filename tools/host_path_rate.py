"""PCIe-inclusive rate of the host-pointer twins (pageable NumPy buffers): never the bench `value`."""
import os, sys, time, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "uav-autonomous-control_amd")]
import numpy as np
from bench import missions
from uav_ac import _native as nat
ctx = nat.Context(0)
B, m, K = 4096, 8, 1000
wps = nat.as_f64(missions(B, m, 0, B))
times = np.empty((B, m)); seg = np.empty((B, m), np.int32); offs = np.empty(B + 1, np.int64)
coeffs = np.empty((B, 8 * m, 3))
_traj = [None]
def plan():
    ctx.call("uavac_minsnap_row_counts", nat.np_ptr(wps), B, m, 3.0, 0.01, nat.np_ptr(times), nat.np_ptr(seg), nat.np_ptr(offs))
    ctx.call("uavac_minsnap_solve", nat.np_ptr(wps), B, m, 3.0, nat.np_ptr(coeffs), None)
    if _traj[0] is None or len(_traj[0]) != int(offs[-1]):
        _traj[0] = np.empty((int(offs[-1]), 11))       # first touch of 325 MB of fresh pages costs ~25 ms on its own: reuse
    traj = _traj[0]
    ctx.call("uavac_minsnap_sample", nat.np_ptr(coeffs), nat.np_ptr(times), B, m, 0.01, nat.np_ptr(offs), nat.np_ptr(traj))
    return traj
traj = plan()
t0 = time.perf_counter(); traj = plan(); t1 = time.perf_counter()
V = nat.Vehicle.default()
state = np.zeros((nat.STATE_ROWS, B)); istate = np.zeros((nat.ISTATE_ROWS, B), np.int32)
ctx.call("uavac_state_init", C.byref(V), nat.np_ptr(np.ascontiguousarray(wps[:, 0, :])), B, 1, nat.np_ptr(state), nat.np_ptr(istate))
log = np.empty((K, 13, B))
def roll():
    ctx.call("uavac_control_rollout", C.byref(V), nat.np_ptr(traj), nat.np_ptr(offs), nat.np_ptr(state), nat.np_ptr(istate), B, K,
             nat.np_ptr(log), None, None, 0)
roll()
t2 = time.perf_counter(); roll(); t3 = time.perf_counter()
print(f"host-pointer plan   : B={B} m={m}: {(t1-t0)*1e3:.1f} ms -> {B*m/(t1-t0)/1e6:.1f} M segments/s ({traj.nbytes/(t1-t0)/1e9:.1f} GB/s of rows to host)")
print(f"host-pointer rollout: B={B} K={K}: {(t3-t2)*1e3:.1f} ms -> {B*K/(t3-t2)/1e9:.2f} G steps/s ({log.nbytes/(t3-t2)/1e9:.1f} GB/s of log to host)")
