"""The N > 1 path on CPU: world_size-2 gloo run of the ragged gather-to-root (uav_ac.fleet.gather_rows)
and of the mission sharding bench.py uses."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import PKG, REPO


def _worker(rank, world, port, out_dir):
    for p in (REPO, PKG):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from uav_ac.comm_host import gather_rows
    from uav_ac.sharding import shard_bounds
    from oracle.minsnap_oracle import plan, synthetic_missions
    B = 5
    lo, hi = shard_bounds(B, rank, world)
    wps = synthetic_missions(B, 2)[lo:hi]                       # each rank plans only its own missions
    rows = np.vstack([plan(w, 3.0, 0.01, method="solve") for w in wps])
    allrows, counts = gather_rows(torch.from_numpy(rows), dst=0)
    assert counts[rank] == len(rows)
    if rank == 0:
        np.save(os.path.join(out_dir, "gathered.npy"), allrows.numpy())
        np.save(os.path.join(out_dir, "counts.npy"), np.array(counts))
    else:
        assert allrows is None
    # the same block split into many small messages (what a > 1 GiB block does under RCCL)
    split, c_split = gather_rows(torch.from_numpy(rows), dst=0, max_message_bytes=7 * 88)
    assert c_split == counts
    if rank == 0:
        assert torch.equal(split, allrows)
    # an empty shard must not dead-lock the gather
    empty = torch.zeros((0 if rank == 1 else 3, 11), dtype=torch.float64)
    g2, c2 = gather_rows(empty, dst=0)
    assert c2 == [0 if r == 1 else 3 for r in range(world)]
    if rank == 0:
        assert g2.shape[0] == 3 * (world - 1)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_ragged_gather_to_root(tmp_path, world):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    from oracle.minsnap_oracle import plan, synthetic_missions
    ref = np.vstack([plan(w, 3.0, 0.01, method="solve") for w in synthetic_missions(5, 2)])
    got = np.load(tmp_path / "gathered.npy")
    counts = np.load(tmp_path / "counts.npy")
    assert counts.sum() == len(ref) and len(counts) == world
    assert np.array_equal(got, ref)                      # mission order preserved: rank blocks are contiguous


def _rows_of(mo, coeffs, times, dt):
    """The oracle's sampler + yaw scan on given coefficients (what oracle.minsnap_oracle.plan does after its solve)."""
    pos, vel, acc, sid = mo.sample(coeffs, times, dt)
    return np.hstack((pos, vel, acc, mo.yaws_from_velocity(vel)[:, None], sid[:, None]))


def _plan_worker(rank, world, port, out_dir, root_share=None):
    """Every rank solves its own missions (oracle), the PLAN is gathered (coefficients, durations, rows per spline) and the
    root re-samples the peers' rows from it: the N > 1 form of RcclComm.gather_plan, host path."""
    from types import SimpleNamespace
    for p in (REPO, PKG):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from uav_ac.comm_host import gather_plan, gather_rows
    from uav_ac.sharding import shard_bounds
    from oracle import minsnap_oracle as mo
    B, m, v, dt = 7, 3, 3.0, 0.01
    lo, hi = shard_bounds(B, rank, world, root_share, 0)            # uneven blocks: a small one for the gather's root
    if root_share is not None:
        assert hi - lo == (1 if rank == 0 else (B - 1 + (world - 1) - rank) // (world - 1))
    wps = mo.synthetic_missions(B, m)[lo:hi]
    co, tm, rows = [], [], []
    for w in wps:
        c, t, _, _ = mo.solve_coefficients(w, v, method="solve")
        co.append(c)
        tm.append(t)
        rows.append(_rows_of(mo, c, t, dt))
    n_rows = sum(len(r) for r in rows)
    mine = SimpleNamespace(m=m, velocity=v, dt=dt, total_rows=n_rows,
                           coeffs=np.stack(co) if co else np.zeros((0, 8 * m, 3)),
                           times=np.stack(tm) if tm else np.zeros((0, m)),
                           seg_rows=np.stack([mo.row_counts(t, dt) for t in tm]).astype(np.int32) if tm else np.zeros((0, m), np.int32))
    parts, counts = gather_plan(mine, dst=0)
    assert counts[rank] == n_rows
    allrows, c2 = gather_rows(torch.from_numpy(np.vstack(rows) if rows else np.zeros((0, 11))), dst=0)
    assert c2 == counts
    if rank == 0:
        assert parts["coeffs"].shape == (B, 8 * m, 3) and parts["seg_rows"].dtype == torch.int32
        np.save(os.path.join(out_dir, "coeffs.npy"), parts["coeffs"].numpy())
        np.save(os.path.join(out_dir, "times.npy"), parts["times"].numpy())
        np.save(os.path.join(out_dir, "seg_rows.npy"), parts["seg_rows"].numpy())
        np.save(os.path.join(out_dir, "rows.npy"), allrows.numpy())
    else:
        assert parts is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,root_share", [(2, None), (3, None), (2, 0.1), (3, 0.1)])
def test_plan_gather_reproduces_the_row_gather(tmp_path, world, root_share):
    """... with equal blocks and with a smaller block for the root (round-3 VERDICT 2: `shard_bounds(..., root_share)`): the
    gathered rows are the same bits in the same mission order either way."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_plan_worker, args=(world, port, str(tmp_path), root_share), nprocs=world, join=True)
    from oracle import minsnap_oracle as mo
    co, tm, sr = (np.load(tmp_path / f) for f in ("coeffs.npy", "times.npy", "seg_rows.npy"))
    rows = np.load(tmp_path / "rows.npy")
    # the gathered plan alone determines the gathered rows: re-sampled on the "root" they are the peers' rows, bit for bit
    again = np.vstack([_rows_of(mo, co[b], tm[b], 0.01) for b in range(len(co))])
    assert np.array_equal(again, rows)
    assert np.array_equal(sr, np.stack([mo.row_counts(t, 0.01) for t in tm]))
    wps = mo.synthetic_missions(7, 3)
    ref = np.vstack([mo.plan(w, 3.0, 0.01, method="solve") for w in wps])
    assert np.array_equal(rows, ref)                       # mission order preserved across rank blocks
