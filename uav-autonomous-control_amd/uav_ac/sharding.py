"""How a job of B independent missions is cut over the ranks of one node: contiguous mission-index
blocks, no data-path collective; and how large the block of the final gather's root should be so that it finishes with
its peers -- from what a shard of n missions costs on the GPU at hand (`measure_tick_table`).  No reference counterpart
(upstream plans and flies one mission per process, uav_ac/main.py:87-120)."""
from __future__ import annotations

import numpy as np


def shard_sizes(B: int, world: int, root_share: float = None, root: int = 0):
    """Missions per rank: contiguous blocks in rank order.  Equal blocks (sizes differ by at most one) unless
    `root_share` is given: then rank `root` -- the rank the trajectories are gathered to -- takes round(root_share * B)
    missions (at least 1 when B >= world; NONE when root_share is exactly 0: the root then only assembles the trajectories, which
    at eight ranks is as much work as a peer's flight) and the other ranks share the rest equally.  The root of the final gather has
    extra work (it re-samples or receives everybody's rows while it flies), so its block is made smaller:
    `balanced_root_share` says by how much."""
    B, world = int(B), int(world)
    if world < 1 or B < 0 or not (0 <= root < world):
        raise ValueError("need world >= 1, B >= 0, 0 <= root < world")
    if root_share is None or world == 1:
        base, rem = divmod(B, world)
        return [base + (1 if r < rem else 0) for r in range(world)]
    if not (0.0 <= root_share <= 1.0):
        raise ValueError("root_share is a fraction of the batch")
    n_root = int(round(root_share * B))
    n_root = max(min(n_root, B), 1 if (B >= world and root_share > 0.0) else 0)
    n_root = min(n_root, B - (world - 1) if B >= world else n_root)     # every peer keeps at least one mission
    base, rem = divmod(B - n_root, world - 1)
    peers = [base + (1 if i < rem else 0) for i in range(world - 1)]
    return peers[:root] + [n_root] + peers[root:]


def shard_bounds(B: int, rank: int, world: int, root_share: float = None, root: int = 0):
    """Contiguous mission-index block [lo, hi) of this rank: no data-path collective needed.  Sizes: `shard_sizes`."""
    sizes = shard_sizes(B, world, root_share, root)
    lo = sum(sizes[:rank])
    return lo, lo + sizes[rank]


# One MI355X, measured (round 6: profiles/r06_config4_rows_free_peer.jsonl, r06_config_sweep.jsonl; m = 8 .. 12): UAVs in flight on
# the GPU, us per logged tick, ms of the planning chain per 1 000 missions WITH rows, the same ROWS-FREE (times, row counts,
# solve, first headings: Engine.plan(rows=False)).  A FALLBACK: `measure_tick_table` measures the same columns on the GPU at hand in
# a few tens of milliseconds, and `bench.py --gpus N` does so before it cuts the shards.  (A three-column table -- rounds 4 / 5 --
# is still taken: rows-free planning is then priced like planning with rows.)
DEFAULT_TICK_TABLE = ((4096, 0.787, 0.027, 0.0090), (16384, 0.792, 0.0180, 0.0035), (24576, 0.843, 0.0170, 0.0026), (32768, 0.857, 0.0171, 0.0022),
                      (36864, 0.880, 0.0172, 0.0021), (49152, 1.019, 0.0163, 0.0021), (65536, 1.262, 0.0163, 0.0021))


def measure_tick_table(engine: "Engine", segments: int, sizes, velocity: float = 3.0, dt: float = 0.01, ticks: int = 2500,
                       launches: int = 2, seed: int = 7, log_bytes_cap: float = 6e9):
    """What a shard of n missions costs on THIS GPU, for every n in `sizes`: [(n, us per logged tick, ms of the planning chain
    per 1 000 missions with rows, the same rows-free)].  Synthetic missions of the SURVEY 8(d) shape, planned once more after a
    warm-up, then `launches` (>= 2) logged launches of `ticks` ticks (the first is thrown away; long launches, as the job itself
    flies them: a launch boundary costs 50-80 us below a full chip).  The log of a launch is capped at `log_bytes_cap` bytes --
    fewer ticks per launch for a large shard (n = 262 144 would otherwise ask for 68 GB) -- never below 200 ticks.  A few tens of
    milliseconds per size."""
    torch = engine._torch
    if int(launches) < 2:
        raise ValueError("launches >= 2: the first launch is a warm-up and is thrown away")
    rng = np.random.default_rng(seed)
    table = []
    for n in sorted({int(x) for x in sizes if int(x) > 0}):
        d = rng.standard_normal((n, segments, 3)) * np.array([1, 1, 0.25])
        d /= np.linalg.norm(d, axis=2, keepdims=True)
        w0 = np.concatenate([rng.uniform(0, 24, (n, 1, 1)), rng.uniform(0, 14, (n, 1, 1)), np.full((n, 1, 1), -3.0)], axis=2)
        wps = np.concatenate([w0, w0 + np.cumsum(rng.uniform(2.5, 3.5, (n, segments, 1)) * d, axis=1)], axis=1)
        plan = engine.plan(wps, velocity, dt)
        free = engine.plan(wps, velocity, dt, rows=False)
        fleet = engine.fleet(free)                            # plan-fed, as the job flies
        pitch = -(-n // 16) * 16
        k = int(max(200, min(int(ticks), log_bytes_cap // (13 * 8 * pitch))))
        log = torch.empty((k, 13, pitch), dtype=torch.float64, device=engine.device)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(6)]
        engine.replan(plan)
        engine.replan(free)
        ev[0].record()
        engine.replan(plan)
        ev[1].record()
        engine.replan(free)
        ev[2].record()
        fleet.reset()
        fleet.rollout(k, state_log=log, log_pitch=pitch)
        ev[3].record()
        for _ in range(int(launches) - 1):
            fleet.rollout(k, state_log=log, log_pitch=pitch)
        ev[4].record()
        torch.cuda.synchronize(engine.device)
        table.append((n, ev[3].elapsed_time(ev[4]) * 1e3 / ((int(launches) - 1) * k), ev[0].elapsed_time(ev[1]) / (n / 1000.0),
                      ev[1].elapsed_time(ev[2]) / (n / 1000.0)))
        del plan, free, fleet, log
    return table


def candidate_shard_sizes(B: int, world: int):
    """The shard sizes worth measuring before `balanced_root_share` cuts a B-mission job over `world` ranks: a small block (what
    the root ends up with when it samples everybody's rows), half an equal block, an equal block, and a peer's block when the root
    takes next to nothing."""
    eq = max(1, B // world)
    return sorted({max(1, eq // 8), max(1, eq // 2), eq, min(B, -(-B // max(1, world - 1)))})


def balanced_root_share(B: int, world: int, ticks: int, segments: int, rows_per_segment: float = 112.9,
                        plan_gather: bool = True, hbm_write_bytes_per_s: float = 4.8e12, tick_table=None, rows_free: bool = True,
                        first_part_s: float = 0.2e-3, flight_beside_sampler: float = 1.25) -> float:
    """The share of a B-mission job the gather's root should take so that it finishes with its peers (BASELINE configs[3]).

    A PROJECTION from one-GPU measurements, not a measurement of N GPUs.  A peer with n missions plans them -- rows-free when
    `rows_free` (round 6: the rows are sampled once, on the root) -- and flies `ticks` logged ticks: both read off `tick_table` =
    [(n, us per logged tick, ms of planning per 1 000 missions with rows [, rows-free])], as `measure_tick_table` returns it for
    the GPU at hand (default: `DEFAULT_TICK_TABLE`), linear between its points, flat below the first, proportional to n above
    the last.  The root does the same for its own block and, beside it, receives the peers' plans and samples EVERYBODY's rows
    (plan gather; pipelined: it starts `first_part_s` after the job -- the peers' planning + the first part on the links -- and
    then streams at the rate the sampler writes rows in the pipelined gather's 4 x world launches beside a flight,
    `hbm_write_bytes_per_s`: 20.85 GB in 4.2 ms alone, 4.4-4.7 ms beside a small flight on one MI355X,
    profiles/r06_config4_root_pipelined.jsonl) or receives the rows themselves over its links.  Its time is the larger of its own
    plan + flight -- the flight `flight_beside_sampler` times as long as alone: its log stores queue behind the sampler's write
    stream, 3.9 -> 4.9-5.4 ms for 2 048 .. 8 192 UAVs -- and of first_part_s + (its log, 104 B per UAV tick, + all rows) / that
    rate.  Bisection; and when even a small block leaves the root behind its peers the answer is 0.0: the root flies NOTHING
    and only assembles the trajectories (`shard_sizes` honours an exact 0) -- at eight ranks sampling 262 144 missions' rows is
    as much work as a peer's 37 450-UAV flight."""
    if world <= 1:
        return 1.0
    table = sorted(tuple(float(v) for v in row) for row in (tick_table or DEFAULT_TICK_TABLE))
    if not table or any(len(row) not in (3, 4) or any(v <= 0 for v in row) for row in table):
        raise ValueError("tick_table: [(missions, us per tick, ms of planning per 1000 missions [, the same rows-free])], all positive")
    col_plan = 3 if (rows_free and all(len(row) == 4 for row in table)) else 2
    row_bytes = 88.0 * rows_per_segment * segments                       # per mission

    def lookup(n, col):
        if n <= table[0][0]:
            return table[0][col]
        for lo_, hi_ in zip(table, table[1:]):
            if n <= hi_[0]:
                return lo_[col] + (hi_[col] - lo_[col]) * (n - lo_[0]) / (hi_[0] - lo_[0])
        return table[-1][col] * (n / table[-1][0] if col == 1 else 1.0)   # a full chip walks its tiles pass after pass

    def own(n):                                                          # plan + flight of n missions, seconds
        return lookup(n, col_plan) * 1e-3 * n / 1000.0 + ticks * lookup(n, 1) * 1e-6

    def root_time(s):
        n = s * B
        stream = (first_part_s if rows_free else 0.0) + (n * ticks * 104.0 + B * row_bytes) / hbm_write_bytes_per_s
        mine = 0.0 if n < 0.5 else own(n) * (flight_beside_sampler if rows_free else 1.0)
        return max(mine, stream) if plan_gather else own(n) + (B - n) * row_bytes / (7 * 153e9 * min(1.0, (world - 1) / 7.0))

    def peer_time(s):
        return own((1.0 - s) * B / (world - 1))

    lo, hi = 0.0, 1.0 / world
    if root_time(hi) <= peer_time(hi):
        return hi                                                        # equal blocks already balance
    for _ in range(50):
        mid = 0.5 * (lo + hi)
        if root_time(mid) > peer_time(mid):
            hi = mid
        else:
            lo = mid
    best = 0.5 * (lo + hi)
    if plan_gather and rows_free:
        # a root that flies nothing: no flight to be slowed by its own sampler -- taken when it finishes the job earlier
        if max(root_time(0.0), peer_time(0.0)) <= max(root_time(best), peer_time(best)) or best * B < 1.0:
            return 0.0
    return best


# How a pipelined plan gather cuts every rank's block (`RcclComm.gather_plan_begin(parts=...)`): cumulative shares of the
# missions.  The first part is small so that the root starts sampling early; every part takes the sampler longer than the next
# one takes to arrive (the sampler consumes a plan at ~2 % of the rate it writes rows: ~15 GB/s per link at eight ranks), and the
# late parts are large launches.
PIPELINE_SHARES = (1.0 / 16, 1.0 / 4, 1.0 / 2, 1.0)


def part_bounds(n_missions: int, shares=PIPELINE_SHARES):
    """Mission-index boundaries [0, b1, ..., n] of the parts of a block of n missions, from cumulative `shares` (increasing, the
    last one 1).  Pure arithmetic: every rank computes every rank's boundaries from the mission counts alone."""
    shares = tuple(float(s) for s in shares)
    if not shares or shares[-1] != 1.0 or any(b <= a for a, b in zip((0.0,) + shares, shares)):
        raise ValueError("shares: increasing cumulative fractions ending in 1")
    n = int(n_missions)
    return [0] + [min(n, int(n * s)) for s in shares[:-1]] + [n]


def gather_layout(counts, dst: int):
    """Where every rank's block lands in the root's buffer: row offsets (world + 1,) and the peers that send.
    Shared by the RCCL path (whose C side derives the same offsets from the same counts) and the host rehearsal."""
    offs = np.concatenate([[0], np.cumsum(np.asarray(counts, dtype=np.int64))])
    return offs, [r for r in range(len(counts)) if r != dst and counts[r] > 0]
