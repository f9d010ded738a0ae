"""GPU tests of round 5: the self-check uavac_create runs (sampler heading == device library atan2), the device identity and the
clock probe behind the C ABI."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

LAB_AABBS = np.array([[3.7, 4.3, 4.0, 10.0, -3.4, -2.8], [10.7, 11.3, 4.0, 10.0, -2.2, 0.0],
                      [13.3, 14.7, 6.3, 7.7, -6.0, 0.0], [20.2, 20.8, 4.0, 10.0, -3.3, -2.7]])


@pytest.fixture(scope="module")
def nat():
    from uav_ac import _native
    return _native


@pytest.fixture(scope="module")
def eng():
    from uav_ac.fleet import Engine
    return Engine("cuda:0")


def _missions(B, m):
    from oracle import minsnap_oracle as mo
    return mo.synthetic_missions(B, m)


def test_create_checks_the_heading_against_the_device_library(nat):
    """uavac_create runs the sampler's heading() and the device library's atan2 -- which the rollout's yaw scan calls -- over 2^16
    operand pairs and every special value and refuses a context when a bit differs (UAVAC_ETOOLCHAIN).  Here: it passes on this
    toolchain, and the check is really there (creation costs a kernel more than with UAVAC_SKIP_SELFCHECK)."""
    import os
    import time
    assert nat.ETOOLCHAIN == -7
    ctx = nat.Context(0)
    ident = ctx.device_identity()
    assert ident.startswith("uuid=") and ";pci=" in ident and "gfx950" in ident
    ctx.close()

    def cost(skip):
        if skip:
            os.environ["UAVAC_SKIP_SELFCHECK"] = "1"
        try:
            t0 = time.perf_counter()
            for _ in range(20):
                nat.Context(0).close()
            return (time.perf_counter() - t0) / 20
        finally:
            os.environ.pop("UAVAC_SKIP_SELFCHECK", None)
    cost(False)
    with_check, without = cost(False), cost(True)
    assert with_check > without, (with_check, without)
    assert with_check < 0.05                                    # ... and cheap


def test_clock_probe_reads_a_plausible_shader_clock(eng):
    """uavac_clock_probe_dev: one wave stamps s_memtime / s_memrealtime a window apart; beside a busy GPU it reads between 1 and 2.6 GHz
    and its window is what was asked for."""
    import torch
    plan = eng.plan(_missions(16384, 8), 3.0, 0.01)
    fleet = eng.fleet(plan)
    side = torch.cuda.Stream(device=eng.device)
    probe = eng.clock_probe_begin(2000, stream=side)
    for _ in range(4):
        fleet.rollout(1000)
    torch.cuda.synchronize()
    c0, r0, c1, r1 = (int(v) for v in probe.cpu().tolist())
    assert 1.95e5 <= r1 - r0 <= 3.0e5                           # 2 ms at 100 MHz (plus the last sleep)
    assert 1.0 < eng.clock_probe_ghz(probe) < 2.6
    with pytest.raises(Exception):
        eng.clock_probe_begin(0)
