"""bench.py --gpus N started WITHOUT a launcher must start its N ranks itself (round-2 VERDICT: it silently ran one GPU and
printed n_gpus: 1).  CPU-side: the launcher with `--launch-check` ranks (rendezvous over gloo, no GPU work)."""
import json
import os
import subprocess
import sys

import pytest

from conftest import REPO

BENCH = os.path.join(REPO, "bench.py")


def _run(args, **env):
    e = dict(os.environ, **env)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        e.pop(k, None)
    return subprocess.run([sys.executable, BENCH] + args, env=e, capture_output=True, text=True, timeout=600)


@pytest.mark.parametrize("n", [2, 3])
def test_self_launch_starts_n_ranks_and_forwards_one_json_line(n):
    r = _run(["--gpus", str(n), "--launch-check"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                                   # rank 0's line only
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == n and rec["rank_sum"] == n * (n + 1) // 2 and rec["master"] == "127.0.0.1"


def test_self_launch_reports_the_worst_exit_code():
    r = _run(["--gpus", "2", "--launch-check"], UAVAC_BENCH_FAIL_RANK="1")
    assert r.returncode == 3


def test_fewer_gpus_than_asked_for_is_an_error_not_a_line():
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("two GPUs visible")
    r = _run(["--gpus", "2"])
    assert r.returncode != 0 and "GPU(s) visible" in r.stderr
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]


def test_world_size_must_match_gpus():
    """Under a launcher with another world size the bench refuses instead of printing a line for the wrong N."""
    e = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--launch-check"], env=e, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE" in (r.stderr + r.stdout)


@pytest.mark.gpu
@pytest.mark.parametrize("n", [2, 3])
def test_rehearsal_ranks_through_the_self_launcher_plan_gather_equals_row_gather(n):
    """The whole N > 1 control flow on one GPU, started by bench.py itself: n ranks (gloo, sharing the GPU) plan and fly
    their shares of BASELINE config 4 -- a smaller block for rank 0, the root of the gather (round-3 VERDICT 2: uneven shards),
    equal blocks with --equal-shards; the rows AND the plan are gathered; rank 0 re-samples the peers' rows from the peers'
    coefficients and finds them bit-identical to the rows the peers sent."""
    from uav_ac.fleet import balanced_root_share, shard_sizes
    # n = 2: the shards are cut from what the ranks MEASURE on their GPU at start-up (round-4 VERDICT 7); n = 3: from a table the
    # test injects.  Either way the line says which table was used and the cut follows from it.
    table = [[30000, 0.9, 0.02], [90000, 1.7, 0.018], [140000, 2.7, 0.018]]
    extra = {} if n == 2 else {"UAVAC_BENCH_TICK_TABLE": json.dumps(table)}
    r = _run(["--gpus", str(n), "--steps", "1", "--warmup", "1", "--no-extras"], UAVAC_BENCH_REHEARSAL="1", **extra)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == n and "REHEARSAL" in line and "gather_error" not in line
    c4 = line["config4"]
    if n == 3:
        assert c4["tick_table"] == table
    else:
        # (round 6: four candidate sizes, and the rows-free planning chain as a fourth column)
        assert [row[0] for row in c4["tick_table"]] == [16384, 65536, 131072, 262144]
        assert all(len(row) == 4 and min(row[1:]) > 0 for row in c4["tick_table"])
    sizes = shard_sizes(262144, n, balanced_root_share(262144, n, 5000, 8, tick_table=c4["tick_table"]), 0)
    assert all(abs(a - b) <= 8 for a, b in zip(c4["shard_sizes"], sizes))       # (the line's table is rounded to 7 digits)
    sizes = c4["shard_sizes"]
    # (measured by two ranks that SHARE the GPU, a shard's cost can come out so high that equal blocks already balance)
    assert c4["batch_per_gpu"] == sizes[0] and sum(sizes) == 262144 and (sizes[0] < min(sizes[1:]) if n == 3 else sizes[0] <= min(sizes[1:]))
    assert len(c4["devices"]) == n and c4["distinct_devices"] == 1 and "rccl_ranks" not in c4       # ranks share the GPU, gloo
    assert c4["gather_verified"] is True and c4["plan_gather_verified"] is True
    # round 6: the rows-free plans of every rank, gathered and sampled part by part in the pipelined gather's order: same rows
    assert c4["pipelined_plan_gather_verified"] is True and c4["compute_rows_free_ms"] > 0
    assert c4["plan_gather_ms"] > 0 and c4["steps_per_s_with_plan_gather"] > 0
    if n == 2:                                  # the same with equal blocks
        r = _run(["--gpus", "2", "--steps", "1", "--warmup", "1", "--no-extras", "--equal-shards"], UAVAC_BENCH_REHEARSAL="1")
        assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
        c4 = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])["config4"]
        assert c4["shard_sizes"] == [131072, 131072] and c4["root_share"] is None
        assert c4["gather_verified"] is True and c4["plan_gather_verified"] is True


@pytest.mark.gpu
def test_rehearsal_with_a_root_that_flies_nothing():
    """At eight ranks the measured balance gives rank 0 NO missions (sampling 262 144 missions' rows is a peer's worth of work, and a
    flight beside its own sampler runs a quarter slower): `shard_sizes[0] == 0`.  Two ranks on one GPU with `--root-share 0`: rank 0
    plans and flies nothing, takes part in every gather with an empty block, and holds all rows at the end -- row gather, one-shot
    plan gather and the pipelined order all verified."""
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "1", "--no-extras", "--root-share", "0"], UAVAC_BENCH_REHEARSAL="1")
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    c4 = line["config4"]
    assert "gather_error" not in line and c4["shard_sizes"] == [0, 262144] and c4["batch_per_gpu"] == 0 and c4["root_share"] == 0.0
    assert c4["gather_verified"] is True and c4["plan_gather_verified"] is True and c4["pipelined_plan_gather_verified"] is True
    assert c4["gather_rows_total"] == 200000 and c4["rows_rank0"] == 0          # (rehearsal: a 200 000-row slice of the peer crosses gloo)


@pytest.mark.gpu
def test_rehearsal_with_one_wrong_gathered_value_exits_3():
    """First contact with eight GPUs is the driver's run: rc 0 must mean the gathered rows were right.  Two ranks on one GPU, one
    value of a PEER's gathered block changed on rank 0 before the verification (UAVAC_BENCH_CORRUPT_GATHER, honoured in rehearsal /
    forced-world-1 runs only): the re-sampled plan no longer equals the gathered rows, the line is still printed with the
    headline and says `mismatch`, every rank exits 3 and so does the launcher -- without UAVAC_BENCH_STRICT."""
    table = [[30000, 0.9, 0.02], [140000, 2.7, 0.018]]
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "1", "--no-extras", "--equal-shards"], UAVAC_BENCH_REHEARSAL="1",
             UAVAC_BENCH_CORRUPT_GATHER="1", UAVAC_BENCH_TICK_TABLE=json.dumps(table))
    assert r.returncode == 3, (r.returncode, (r.stdout + r.stderr)[-3000:])
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["gather_error_kind"] == "mismatch" and "differ" in line["gather_error"] and line["value"] > 0
    assert line["config4"]["plan_gather_verified"] is False and "end_to_end" not in line["config4"]
    assert "FAILED" in r.stderr


@pytest.mark.gpu
def test_two_real_gpus_over_rccl_when_the_box_has_them():
    """First contact with a real peer (round-2 ADVICE): on a box with >= 2 GPUs the self-launcher starts two ranks, one per GPU,
    over RCCL; both gathers of the config-4 leg are verified on rank 0 (the re-sampled plan equals the gathered rows bit for
    bit) and the overlapped variants run.  Skipped on the 1-GPU boxes of this pool."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("one GPU visible")
    r = _run(["--gpus", "2", "--steps", "2", "--warmup", "1", "--no-extras"])
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and "REHEARSAL" not in line and "gather_error" not in line
    c4 = line["config4"]
    assert c4["gather_verified"] is True and c4["plan_gather_verified"] is True and "overlap_error" not in c4
    assert c4["plan_overlapped_verified"] is True and c4["overlapped_verified"] is True
    assert c4["rccl_ranks"] == 2 and c4["distinct_devices"] == 2 and len(c4["tick_table"]) == 4
    assert c4["pipelined_plan_gather_verified"] is True and c4["rows_free_overlapped_verified"] is True
    assert c4["end_to_end"]["form"].startswith("every rank plans ROWS-FREE") and c4["rccl_versions"]["runtime"] > 0


def test_ranks_started_by_torch_distributed_run_are_not_launched_again():
    """The driver's own form: `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`.  WORLD_SIZE is set,
    so bench.py is a rank, not a launcher: exactly one line, from rank 0."""
    e = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        e.pop(k, None)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29581", BENCH, "--gpus", "2", "--launch-check"], env=e, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and json.loads(lines[0])["n_gpus"] == 2
