"""uav_ac -- MI355X-native batched drop-in for the minimum-snap + cascaded-control hot path of
Mdhvince/UAV-Autonomous-control.  Same module layout as the reference package:

    uav_ac.planning.minimum_snap.MinimumSnap       single-mission facade  (reference: same path)
    uav_ac.control.controller.CascadedController   single-UAV facade
    uav_ac.quadrotor.quad.Quad                     single-UAV facade
    uav_ac.main.TrajectoryController               single-UAV facade
    uav_ac.fleet.Engine / Fleet / Plan             batched API (new)

Everything numeric runs in hand-written HIP kernels behind the C ABI of include/uavac.h
(libuavac.so); there is no CPU fallback.
"""
__version__ = "0.1.0"
