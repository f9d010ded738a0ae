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
    from uav_ac.fleet import gather_rows, shard_bounds
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
