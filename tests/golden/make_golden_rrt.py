#!/usr/bin/env python3
"""Generate tests/golden/rrt_*.npz by IMPORTING THE REFERENCE's RRT* planner (uav_ac/planning/rrt.py).

Runs only in the build container (the upstream tree is mounted read-only at /root/reference; it does
not exist on the GPU box and nothing under tests/ reads it at test time).  Only the *.npz outputs are
committed: inputs (seed, scene, the nodes _generate_random_node returned) and the reference's outputs.

    python tests/golden/make_golden_rrt.py

While generating, the C oracle (oracle/rrt_oracle.c) is run on the same node sequence and must agree
exactly with the reference -- that is what pins it.
"""
import contextlib
import io
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("UAVAC_REFERENCE", "/root/reference")
sys.path.insert(0, REF)
sys.path.insert(1, REPO)

from uav_ac.planning.rrt import RRTStar                 # noqa: E402  (reference)
from oracle import c_oracle as co                       # noqa: E402  (ours, checked here)

assert os.path.realpath(sys.modules["uav_ac.planning.rrt"].__file__).startswith(os.path.realpath(REF))

LAB_AABBS = np.array([                # lab_course.xml:37,53,54,67 through mujoco_sim.py:282-300
    [3.7, 4.3, 4.0, 10.0, -3.4, -2.8], [10.7, 11.3, 4.0, 10.0, -2.2, 0.0],
    [13.3, 14.7, 6.3, 7.7, -6.0, 0.0], [20.2, 20.8, 4.0, 10.0, -3.3, -2.7]])

SCENES = {
    # tests/conftest.py:8-13 geometry with a useful iteration count
    "cube": dict(limits=[[0, 0, 0], [10, 10, 10]], start=[0, 0, 0], goal=[8, 8, 8], step=2, max_iter=400,
                 obstacles=None, seeds=[0, 1, 2, 3]),
    "cube_wall": dict(limits=[[0, 0, 0], [10, 10, 10]], start=[0.5, 0.5, 0.5], goal=[9, 9, 5], step=1.5, max_iter=1500,
                      obstacles=[[4.0, 5.0, -1.0, 7.0, -1.0, 11.0], [6.5, 7.0, 3.0, 11.0, -1.0, 11.0]],
                      seeds=[5, 6, 7]),
    "lab": dict(limits=[[0, 0, -6], [24, 14, 0]], start=[1.0, 7.0, -1.3], goal=[23.0, 7.0, -2.0], step=1.5,
                max_iter=1200, obstacles=LAB_AABBS.tolist(), seeds=[11, 12]),
    "fine": dict(limits=[[0, 0, 0], [3, 3, 1]], start=[0.1, 0.1, 0.5], goal=[2.9, 2.9, 0.5], step=0.3, max_iter=800,
                 obstacles=[[1.0, 2.0, 1.0, 2.0, 0.0, 1.0]], seeds=[21, 22]),
    # too few iterations: the reference raises
    "short": dict(limits=[[0, 0, 0], [10, 10, 10]], start=[0, 0, 0], goal=[8, 8, 8], step=1, max_iter=5,
                  obstacles=None, seeds=[0]),
}


def run_reference(scene, seed):
    obstacles = None if scene["obstacles"] is None else np.array(scene["obstacles"], dtype=float)
    rrt = RRTStar(space_limits=np.array(scene["limits"], dtype=float), start=np.array(scene["start"], dtype=float),
                  goal=np.array(scene["goal"], dtype=float), max_distance=scene["step"],
                  max_iterations=scene["max_iter"], obstacles=obstacles)
    drawn = []
    inner = rrt._generate_random_node

    def recording():
        node = inner()
        drawn.append(np.array(node, dtype=float))
        return node
    rrt._generate_random_node = recording
    np.random.seed(seed)
    error = ""
    with contextlib.redirect_stdout(io.StringIO()):
        try:
            rrt.run()
        except Exception as exc:                                   # noqa: BLE001
            error = f"{type(exc).__name__}: {exc}"
    state_after = np.random.get_state()
    return rrt, np.array(drawn).reshape(-1, 3), error, state_after


def tree_as_parents(rrt, tree):
    """For every all_nodes entry: the coordinates tree[key(entry)] holds (nan when the dict has no such key)."""
    out = np.full((len(rrt.all_nodes), 3), np.nan)
    for e, node in enumerate(rrt.all_nodes):
        k = RRTStar._node_key(node)
        if tree is not None and k in tree:
            out[e] = tree[k]
    return out


def check_oracle(name, scene, rrt, samples, error):
    n_iter = len(samples)
    padded = np.zeros((scene["max_iter"], 3))
    padded[:n_iter] = samples                                  # iterations after the early stop are never read
    res = co.rrt_star(scene["start"], scene["goal"], scene["step"], padded, scene["obstacles"])
    assert res["iters"] == n_iter, (name, res["iters"], n_iter)
    assert res["dynamic_it_counter"] == rrt.dynamic_it_counter, name
    nodes = np.array(rrt.all_nodes, dtype=float).reshape(-1, 3)
    assert np.array_equal(res["nodes"], nodes), name
    par = tree_as_parents(rrt, rrt.tree)
    got = np.where(res["parent"][res["canon"]][:, None] >= 0, res["nodes"][np.maximum(res["parent"][res["canon"]], 0)], np.nan)
    assert np.array_equal(got, par, equal_nan=True), name
    if error:
        assert res["status"] != 0, (name, error)
    else:
        assert res["status"] == 0, (name, res["status"])
        assert np.array_equal(res["best_path"], rrt.best_path), name
        assert res["best_cost"] == RRTStar.path_cost(list(rrt.best_path[::-1])), name
    return res


def main():
    rng = np.random.default_rng(7)
    # np.linalg.norm of a 3-vector on the BLAS the goldens were made with == the oracle's fma sequence
    pts = np.round(rng.uniform(-20, 20, (20000, 3)), 2)
    q = np.round(rng.uniform(-20, 20, 3), 2)
    assert np.array_equal(co.rrt_distances(pts, q), np.array([np.linalg.norm(q - p) for p in pts]))

    for name, scene in SCENES.items():
        for seed in scene["seeds"]:
            rrt, samples, error, _ = run_reference(scene, seed)
            res = check_oracle(f"{name}/{seed}", scene, rrt, samples, error)
            best_tree_par = tree_as_parents(rrt, rrt.best_tree)
            np.savez_compressed(
                os.path.join(HERE, f"rrt_{name}_{seed}.npz"),
                seed=seed, limits=np.array(scene["limits"], dtype=float), start=np.array(scene["start"], dtype=float),
                goal=np.array(scene["goal"], dtype=float), step=float(scene["step"]), max_iter=scene["max_iter"],
                obstacles=np.zeros((0, 6)) if scene["obstacles"] is None else np.array(scene["obstacles"], dtype=float),
                samples=samples, error=error, dynamic_it_counter=rrt.dynamic_it_counter,
                all_nodes=np.array(rrt.all_nodes, dtype=float).reshape(-1, 3),
                tree_parent=tree_as_parents(rrt, rrt.tree), best_tree_parent=best_tree_par,
                best_path=np.zeros((0, 3)) if rrt.best_path is None else rrt.best_path,
                best_cost=np.nan if rrt.best_path is None else RRTStar.path_cost(list(rrt.best_path[::-1])),
                simplified_path=np.zeros((0, 3)) if rrt.best_path is None else rrt.simplify_path(rrt.best_path))
            tail = f"ERROR {error}" if error else f"path of {len(rrt.best_path)} nodes, cost {res['best_cost']:.4f}"
            print(f"{name}/{seed}: {len(samples)} iterations, {len(rrt.all_nodes)} nodes, {tail}")

    # slab test known answers: random segments (some axis-parallel, some touching faces) x random cuboids
    n = 4000
    a = np.round(rng.uniform(-2, 12, (n, 3)), 2)
    b = np.round(rng.uniform(-2, 12, (n, 3)), 2)
    par = rng.integers(0, 4, n)
    for ax in range(3):
        b[par == ax + 1, ax] = a[par == ax + 1, ax]            # a quarter each: parallel to one axis
    lo = np.round(rng.uniform(0, 6, (n, 3)), 2)
    hi = lo + np.round(rng.uniform(0.5, 7, (n, 3)), 2)
    cub = np.stack([lo[:, 0], hi[:, 0], lo[:, 1], hi[:, 1], lo[:, 2], hi[:, 2]], axis=1)
    touch = rng.integers(0, 8, n) == 0
    a[touch, 0] = cub[touch, 0]                                 # start exactly on a face
    hit = np.array([RRTStar._segment_intersects_cuboid(a[i], b[i], cub[i]) for i in range(n)])
    assert np.array_equal(hit, np.array([co.segment_intersects_cuboid(a[i], b[i], cub[i]) for i in range(n)]))
    np.savez_compressed(os.path.join(HERE, "rrt_slab.npz"), a=a, b=b, cuboid=cub, hit=hit)
    print(f"slab: {hit.sum()} of {n} segments hit")

    # the draw order of _generate_random_node: legacy global RandomState, 1 or 4 uniform() calls
    scene = SCENES["lab"]
    rrt = RRTStar(space_limits=np.array(scene["limits"], dtype=float), start=np.array(scene["start"], dtype=float),
                  goal=np.array(scene["goal"], dtype=float), max_distance=scene["step"], max_iterations=10)
    np.random.seed(1234)
    draws = np.array([rrt._generate_random_node() for _ in range(2000)], dtype=float)
    after = np.random.uniform(0, 1)
    np.savez_compressed(os.path.join(HERE, "rrt_draws.npz"), seed=1234, limits=np.array(scene["limits"], dtype=float),
                        goal=np.array(scene["goal"], dtype=float), nodes=draws, next_uniform=after)


if __name__ == "__main__":
    main()
