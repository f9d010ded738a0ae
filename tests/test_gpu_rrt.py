"""RRT* on the GPU (uavac_rrt_*): bit-exact against the reference's recorded runs (tests/golden/rrt_*.npz), against
the C oracle on seeded random problems, and the reference's own unit tests (upstream
tests/unit/planning/test_rrt.py) replayed on the drop-in `uav_ac.planning.rrt.RRTStar`."""
import glob
import os

import numpy as np
import pytest

from conftest import GOLDEN

pytestmark = pytest.mark.gpu

RUNS = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, "rrt_*_[0-9]*.npz")))


def load(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


def parents_as_coordinates(nodes, canon, parent):
    p = parent[canon]
    return np.where(p[:, None] >= 0, nodes[np.maximum(p, 0)], np.nan)


def padded(g):
    s = np.zeros((int(g["max_iter"]), 3))
    s[:len(g["samples"])] = g["samples"]
    return s


def assert_same_as_oracle(res, b, ref):
    n = ref["nodes"].shape[0]
    assert res.n_nodes[b] == n
    assert res.iterations[b] == ref["iters"]
    assert res.status[b] == ref["status"]
    assert res.dynamic_it_counter[b] == ref["dynamic_it_counter"]
    assert np.array_equal(res.nodes[b, :n], ref["nodes"])
    assert np.array_equal(res.canon[b, :n], ref["canon"])
    assert np.array_equal(res.parent[b, :n], ref["parent"])
    if ref["status"] == 0:
        assert res.best_n[b] == ref["best_n"]
        assert np.array_equal(res.best_parent[b, :n], ref["best_parent"])
        assert np.array_equal(res.path(b), ref["best_path"])
        assert res.best_cost[b] == ref["best_cost"]


@pytest.mark.parametrize("name", RUNS)
def test_run_matches_reference_recording(name):
    """The kernel on the node sequence the reference drew == the reference's all_nodes, tree, best_tree, best_path."""
    from uav_ac.planning.rrt import rrt_star_batch
    g = load(name)
    obstacles = g["obstacles"] if len(g["obstacles"]) else None
    res = rrt_star_batch(g["start"][None], g["goal"][None], float(g["step"]), padded(g)[None], obstacles)
    n = len(g["all_nodes"])
    assert res.iterations[0] == len(g["samples"])
    assert res.n_nodes[0] == n
    assert res.dynamic_it_counter[0] == int(g["dynamic_it_counter"])
    assert np.array_equal(res.nodes[0, :n], g["all_nodes"])
    assert np.array_equal(parents_as_coordinates(res.nodes[0, :n], res.canon[0, :n], res.parent[0, :n]), g["tree_parent"],
                          equal_nan=True)
    if str(g["error"]):
        assert res.status[0] == 1
        assert res.best_len[0] == 0 and np.isinf(res.best_cost[0])
        return
    assert res.status[0] == 0
    bn = int(res.best_n[0])
    best = parents_as_coordinates(res.nodes[0, :n], res.canon[0, :n], res.best_parent[0, :n])
    assert np.array_equal(best[:bn], g["best_tree_parent"][:bn], equal_nan=True)
    assert np.array_equal(res.path(0), g["best_path"])
    assert res.best_cost[0] == float(g["best_cost"])


def test_facade_run_reproduces_seeded_reference_run(capsys):
    """RRTStar(...).run() after np.random.seed(s): same all_nodes / tree / best_path as the reference with the same
    seed, and the global generator is left where the reference leaves it."""
    from uav_ac.planning.rrt import RRTStar
    for name in ("rrt_lab_11", "rrt_cube_1", "rrt_fine_22"):
        g = load(name)
        obstacles = g["obstacles"] if len(g["obstacles"]) else None
        rrt = RRTStar(space_limits=g["limits"], start=g["start"], goal=g["goal"], max_distance=float(g["step"]),
                      max_iterations=int(g["max_iter"]), obstacles=obstacles)
        np.random.seed(int(g["seed"]))
        rrt.run()
        assert "Best path found with cost" in capsys.readouterr().out
        assert np.array_equal(np.array(rrt.all_nodes), g["all_nodes"])
        assert np.array_equal(rrt.best_path, g["best_path"])
        for e, node in enumerate(rrt.all_nodes):
            k = RRTStar._node_key(node)
            if np.isnan(g["tree_parent"][e, 0]):
                assert k not in rrt.tree
            else:
                assert np.array_equal(rrt.tree[k], g["tree_parent"][e])
        path, cost = rrt.get_path(rrt.best_tree)
        assert np.array_equal(path, g["best_path"]) and cost == float(g["best_cost"])
        assert np.array_equal(rrt.simplify_path(rrt.best_path), g["simplified_path"])
        assert len(g["simplified_path"]) <= len(g["best_path"])
        # generator state: the next draws equal what follows the recorded sequence
        after = np.random.uniform(0, 1)
        np.random.seed(int(g["seed"]))
        for _ in range(len(g["samples"])):
            rrt._generate_random_node()
        assert np.random.uniform(0, 1) == after
    g = load("rrt_short_0")
    rrt = RRTStar(space_limits=g["limits"], start=g["start"], goal=g["goal"], max_distance=float(g["step"]),
                  max_iterations=int(g["max_iter"]))
    np.random.seed(0)
    with pytest.raises(Exception, match="No path found"):
        rrt.run()


@pytest.mark.parametrize("max_iter,n_obs", [(300, 0), (700, 3), (1500, 2)])
def test_batch_matches_oracle_on_random_problems(max_iter, n_obs):
    """B problems with their own start / goal / seeds in one launch == the C oracle, problem by problem."""
    from oracle import c_oracle as co
    from uav_ac.planning.rrt import draw_random_nodes, rrt_star_batch
    rng = np.random.default_rng(100 + max_iter)
    B = 48
    lw, up = np.array([0.0, 0.0, -4.0]), np.array([12.0, 9.0, 0.0])
    obstacles = None
    if n_obs:
        lo = rng.uniform([2, 1, -4], [9, 6, -2], (n_obs, 3))
        size = rng.uniform([0.4, 1.5, 1.0], [1.5, 4.0, 4.0], (n_obs, 3))
        obstacles = np.round(np.stack([lo[:, 0], lo[:, 0] + size[:, 0], lo[:, 1], lo[:, 1] + size[:, 1], lo[:, 2],
                                       lo[:, 2] + size[:, 2]], axis=1), 2)
    starts = rng.uniform(lw, lw + [1.5, 9, 4], (B, 3))
    goals = rng.uniform(up - [1.5, 9, 4], up, (B, 3))
    samples = np.stack([draw_random_nodes(np.random.RandomState(1000 + b).random_sample, lw, up, np.round(goals[b], 2),
                                          max_iter)[0] for b in range(B)])
    step = 0.9
    res = rrt_star_batch(starts, goals, step, samples, obstacles)
    found = 0
    for b in range(B):
        ref = co.rrt_star(starts[b], goals[b], step, samples[b], obstacles)
        assert_same_as_oracle(res, b, ref)
        found += ref["status"] == 0
    assert found >= B // 2                              # the comparison is not vacuous
    assert np.all(res.nodes[np.arange(B), 0] == np.round(starts, 2))
    for b in range(B):                                  # rows past the end are zero
        assert not res.nodes[b, res.n_nodes[b]:].any() and not res.best_path[b, res.best_len[b]:].any()


def test_device_resident_entry_point_equals_host_twin():
    """Engine.rrt_star (torch tensors in and out, uavac_rrt_star_dev) == rrt_star_batch (NumPy, uavac_rrt_star)."""
    import dataclasses
    from uav_ac.fleet import Engine
    from uav_ac.planning.rrt import draw_random_nodes_batch, rrt_star_batch
    rng = np.random.default_rng(9)
    B, max_iter = 70, 600
    lw, up = np.array([0.0, 0.0, 0.0]), np.array([10.0, 10.0, 4.0])
    starts, goals = rng.uniform(lw, [2, 10, 4], (B, 3)), rng.uniform([8, 0, 0], up, (B, 3))
    obstacles = np.array([[4.0, 5.0, 5.0, 11.0, -1.0, 5.0], [6.5, 7.0, -1.0, 4.0, -1.0, 5.0]])
    samples = draw_random_nodes_batch(np.arange(B) + 40, lw, up, np.round(goals, 2), max_iter)
    host = rrt_star_batch(starts, goals, 1.0, samples, obstacles)
    dev = Engine("cuda:0").rrt_star(starts, goals, 1.0, samples, obstacles).to_host()
    for f in dataclasses.fields(host):
        assert np.array_equal(getattr(host, f.name), getattr(dev, f.name)), f.name
    assert (host.status == 0).sum() > B // 4


def test_gpu_node_drawing_is_numpys_stream():
    """Engine.rrt_draw_nodes == 2 000 calls of the reference's _generate_random_node after np.random.seed (golden), and
    == the host replay of NumPy's generator for many seeds, including seeds that differ in the top bit and runs
    long enough to regenerate the MT19937 state dozens of times; `consumed` counts the stream's doubles."""
    from uav_ac.fleet import Engine
    from uav_ac.planning.rrt import draw_random_nodes, draw_random_nodes_batch
    eng = Engine("cuda:0")
    g = load("rrt_draws")
    got = eng.rrt_draw_nodes([int(g["seed"])], g["goal"][None], g["limits"][0], g["limits"][1], len(g["nodes"]))
    assert np.array_equal(got[0].cpu().numpy(), g["nodes"])
    seeds = np.array([0, 1, 2, 12345, 2**31 - 1, 2**31, 2**32 - 1, 987654321] + list(range(100, 164)))
    rng = np.random.default_rng(4)
    lw, up = np.array([-3.0, 0.5, -6.0]), np.array([24.0, 14.25, 0.0])
    goals = np.round(rng.uniform(lw, up, (len(seeds), 3)), 2)
    n = 5000
    samples, consumed = eng.rrt_draw_nodes(seeds, goals, lw, up, n, with_consumed=True)
    assert np.array_equal(samples.cpu().numpy(), draw_random_nodes_batch(seeds, lw, up, goals, n))
    ref_nodes, ref_consumed = draw_random_nodes(np.random.RandomState(int(seeds[3])).random_sample, lw, up, goals[3], n)
    assert np.array_equal(consumed[3].cpu().numpy(), ref_consumed)
    other = eng.rrt_draw_nodes(seeds[:4], goals[:4], lw, up, 300, epsilon=0.5)
    assert np.array_equal(other.cpu().numpy(), draw_random_nodes_batch(seeds[:4], lw, up, goals[:4], 300, epsilon=0.5))


def test_batched_simplify_matches_reference_and_facade():
    """Engine.rrt_simplify (one wavefront per path) == the reference's simplify_path of its own best paths (goldens),
    and == the facade's host loop on a batch of fresh problems."""
    from uav_ac.fleet import Engine
    from uav_ac.planning.rrt import RRTStar, draw_random_nodes_batch
    eng = Engine("cuda:0")
    for name in RUNS:
        g = load(name)
        if str(g["error"]):
            continue
        obstacles = g["obstacles"] if len(g["obstacles"]) else None
        res = eng.rrt_star(g["start"][None], g["goal"][None], float(g["step"]), padded(g)[None], obstacles)
        paths, lens = eng.rrt_simplify(res, obstacles)
        n = int(lens[0])
        assert np.array_equal(paths[0, :n].cpu().numpy(), g["simplified_path"]), name
        assert not paths[0, n:].any()
    rng = np.random.default_rng(12)
    B, max_iter = 96, 900
    lw, up = np.array([0.0, 0.0, -6.0]), np.array([24.0, 14.0, 0.0])
    obstacles = np.array([[3.7, 4.3, 4.0, 10.0, -3.4, -2.8], [10.7, 11.3, 4.0, 10.0, -2.2, 0.0],
                          [13.3, 14.7, 6.3, 7.7, -6.0, 0.0], [20.2, 20.8, 4.0, 10.0, -3.3, -2.7]])
    starts = np.round(rng.uniform([0.5, 1, -3], [2, 13, -1], (B, 3)), 2)
    goals = np.round(rng.uniform([22, 1, -3], [23.5, 13, -1], (B, 3)), 2)
    res = eng.rrt_star(starts, goals, 1.5, draw_random_nodes_batch(np.arange(B), lw, up, goals, max_iter), obstacles)
    paths, lens = eng.rrt_simplify(res, obstacles)
    host = res.to_host()
    helper = RRTStar(np.stack([lw, up]), starts[0], goals[0], 1.5, 1, obstacles)
    shorter = 0
    for b in range(B):
        n = int(lens[b])
        if host.status[b] != 0:
            assert n == 0
            continue
        ref = helper.simplify_path(host.path(b))
        assert np.array_equal(paths[b, :n].cpu().numpy(), ref)
        shorter += n < host.best_len[b]
    assert shorter > B // 2


def test_edge_cases_match_oracle():
    """Degenerate problems, each against the oracle: start == goal, goal walled in (no path), a single iteration, a step
    longer than the space, a zero-volume cuboid, a start inside an obstacle, trees that overflow the first-pass
    capacity by one node and by many."""
    from oracle import c_oracle as co
    from uav_ac.planning.rrt import draw_random_nodes_batch, rrt_star_batch
    lw, up = np.array([0.0, 0.0, 0.0]), np.array([6.0, 6.0, 3.0])
    box_in = np.array([[2.0, 4.0, 2.0, 4.0, -1.0, 4.0]])
    cases = [
        ("start == goal", [1.0, 1.0, 1.0], [1.0, 1.0, 1.0], 1.0, 60, None),
        ("goal walled in", [0.5, 0.5, 0.5], [3.0, 3.0, 1.5], 0.8, 300, box_in),
        ("one iteration", [0.5, 0.5, 0.5], [0.9, 0.5, 0.5], 1.0, 1, None),
        ("step longer than the space", [0.5, 0.5, 0.5], [5.5, 5.5, 2.5], 50.0, 40, None),
        ("zero-volume cuboid", [0.5, 0.5, 0.5], [5.5, 5.5, 2.5], 1.0, 200, np.array([[3.0, 3.0, 0.0, 6.0, 0.0, 3.0]])),
        ("start inside an obstacle", [3.0, 3.0, 1.5], [5.5, 5.5, 2.5], 0.8, 200, box_in),
        ("dense tree, small steps", [0.5, 0.5, 0.5], [5.5, 5.5, 2.5], 0.15, 2500, None),
    ]
    for name, start, goal, step, max_iter, obstacles in cases:
        start, goal = np.array(start), np.array(goal)
        samples = draw_random_nodes_batch([3, 4, 5], lw, up, np.stack([goal] * 3), max_iter)
        res = rrt_star_batch(np.stack([start] * 3), np.stack([goal] * 3), step, samples, obstacles)
        for b in range(3):
            assert_same_as_oracle(res, b, co.rrt_star(start, goal, step, samples[b], obstacles))
        if name == "goal walled in":
            assert (res.status == 1).all()
        if name == "dense tree, small steps":
            assert res.n_nodes.max() > 384                    # crossed the first-pass capacity


def test_large_tree_takes_the_scratch_path():
    """max_iterations too large for LDS (52 B per node > 160 KB): the same kernel on HBM scratch, same results."""
    from oracle import c_oracle as co
    from uav_ac.planning.rrt import draw_random_nodes, rrt_star_batch
    max_iter = 3600
    lw, up = np.array([0.0, 0.0, 0.0]), np.array([10.0, 10.0, 10.0])
    starts = np.array([[0.5, 0.5, 0.5], [9.0, 1.0, 2.0]])
    goals = np.array([[9.5, 9.5, 9.5], [1.0, 9.0, 8.0]])
    obstacles = np.array([[4.0, 6.0, -1.0, 8.0, -1.0, 11.0]])
    samples = np.stack([draw_random_nodes(np.random.RandomState(7 + b).random_sample, lw, up, goals[b], max_iter)[0]
                        for b in range(2)])
    res = rrt_star_batch(starts, goals, 0.5, samples, obstacles)
    for b in range(2):
        assert_same_as_oracle(res, b, co.rrt_star(starts[b], goals[b], 0.5, samples[b], obstacles))


def test_primitives_match_known_answers():
    from oracle import c_oracle as co
    from uav_ac.planning import rrt as R
    g = load("rrt_slab")
    # each segment against its own cuboid: one call per distinct cuboid would be slow; test E edges x 1 cuboid
    # through the any-hit entry point on a few cuboids, and the whole set against the oracle edge by edge
    for c in range(0, 6):
        hit = R._segment_hits(g["a"], g["b"], g["cuboid"][c])
        ref = np.array([co.segment_intersects_cuboid(a, b, g["cuboid"][c]) for a, b in zip(g["a"], g["b"])])
        assert np.array_equal(hit, ref)
    own = np.array([R.RRTStar._segment_intersects_cuboid(g["a"][i], g["b"][i], g["cuboid"][i]) for i in range(300)])
    assert np.array_equal(own, g["hit"][:300])
    # any-hit over several cuboids
    multi = R._segment_hits(g["a"], g["b"], g["cuboid"][:5])
    ref = np.zeros(len(g["a"]), bool)
    for c in range(5):
        ref |= np.array([co.segment_intersects_cuboid(a, b, g["cuboid"][c]) for a, b in zip(g["a"], g["b"])])
    assert np.array_equal(multi, ref)
    # edge lengths and steering, bit for bit
    assert np.array_equal(R._edge_lengths(g["a"], g["b"]), co.rrt_edge_lengths(g["a"], g["b"]))
    assert np.array_equal(R._edge_lengths(g["a"], g["b"][0]), co.rrt_edge_lengths(g["a"], g["b"][0]))
    rrt = R.RRTStar(np.array([[0, 0, 0], [10, 10, 10]]), np.array([0, 0, 0]), np.array([8, 8, 8]), 2, 1)
    for i in range(200):
        got = rrt._adapt_random_node_position(g["a"][i], g["b"][i])
        assert np.array_equal(got, co.rrt_steer(g["a"][i], g["b"][i], 2.0))


def test_segment_test_shortcut_never_changes_an_answer():
    """The kernel skips a cuboid's slab test when both end points lie beyond one of its faces by a margin
    (1e-6 + 1e-9 |d|).  Adversarial edges -- end points ON faces, within 1e-5 .. 1e-9 .. one ulp of them, inside,
    axis-parallel, degenerate, very long -- against the oracle, which always runs the full test."""
    from oracle import c_oracle as co
    from uav_ac.planning import rrt as R
    rng = np.random.default_rng(77)
    cub = np.array([[1.0, 2.0, -1.0, 3.0, 0.5, 0.75], [-4.0, -3.99, -10.0, 10.0, -10.0, 10.0],
                    [5.0, 5.0, 0.0, 1.0, 0.0, 1.0], [100.0, 250.0, 100.0, 250.0, -3.0, 3.0]])
    n = 200000
    which = rng.integers(0, len(cub), n)
    face = cub[which]
    lo, hi = face[:, 0::2], face[:, 1::2]
    a = rng.uniform(lo - 2.0, hi + 2.0)
    b = rng.uniform(lo - 2.0, hi + 2.0)
    offs = np.array([0.0, 1e-5, -1e-5, 1e-6, -1e-6, 2e-6, -2e-6, 1e-7, -1e-7, 1e-9, -1e-9, 1e-12, -1e-12])
    for pts in (a, b):                                      # snap a third of the coordinates onto / next to a face
        snap = rng.random((n, 3)) < 0.33
        side = np.where(rng.random((n, 3)) < 0.5, lo, hi)
        near = side + rng.choice(offs, (n, 3))
        ulp = np.nextafter(side, np.where(rng.random((n, 3)) < 0.5, np.inf, -np.inf))
        near = np.where(rng.random((n, 3)) < 0.2, ulp, near)
        pts[snap] = near[snap]
    par = rng.random((n, 3)) < 0.15
    b[par] = a[par]                                         # axis-parallel and degenerate segments
    far = rng.random(n) < 0.05
    b[far] = a[far] + rng.uniform(-1e6, 1e6, (int(far.sum()), 3))
    got = R._segment_hits(a, b, cub)
    ref = co.rrt_segment_hits(a, b, cub)
    assert np.array_equal(got, ref)
    assert 0.05 < ref.mean() < 0.95
    for c in cub:                                           # and cuboid by cuboid
        assert np.array_equal(R._segment_hits(a[:50000], b[:50000], c), co.rrt_segment_hits(a[:50000], b[:50000], c))


# ------------------------------------------------------------- upstream tests/unit/planning/test_rrt.py, replayed
@pytest.fixture
def rrt_object():
    from uav_ac.planning.rrt import RRTStar
    return RRTStar(space_limits=np.array([[0, 0, 0], [10, 10, 10]]), start=np.array([0, 0, 0]),
                   goal=np.array([8, 8, 8]), max_distance=2, max_iterations=1)


def test_reference_unit_tests_geometry(rrt_object):
    from uav_ac.planning.rrt import RRTStar
    assert np.isclose(RRTStar.path_cost(np.array([[1, 1, 1], [3, 3, 9], [11, 5, 5], [1, 1, 1]])), 29.1, atol=0.1)
    rrt_object.obstacles = None
    path = np.array([[0., 0., 0.], [2., 0., 0.], [4., 0., 0.], [6., 0., 0.]])
    assert rrt_object.simplify_path(path) == pytest.approx(np.array([[0., 0., 0.], [6., 0., 0.]]))
    rrt_object.obstacles = np.array([[5., 7., -1., 1., -1., 1.]])
    path = np.array([[0., 0., 0.], [4., 2., 0.], [8., 2., 0.], [12., 0., 0.]])
    result = rrt_object.simplify_path(path)
    assert len(result) > 2
    assert all(rrt_object._is_valid_connection(a, b) for a, b in zip(result[:-1], result[1:]))
    rrt_object.obstacles = None
    node = rrt_object._generate_random_node()
    assert np.all(node >= rrt_object.space_limits_lw) and np.all(node <= rrt_object.space_limits_up)
    rrt_object.all_nodes = np.array([[1, 1, 1], [3, 3, 9], [11, 5, 5], [1, 1, 1]])
    for q in ([1, 1, 1], [3, 3, 9], [11, 5, 5]):
        assert np.all(rrt_object._find_nearest_node(q) == q)
    rrt_object.step_size = 2
    assert np.all(rrt_object._adapt_random_node_position(np.array([1, 1, 1]), np.array([3, 3, 9])) ==
                  np.array([2.53, 2.53, 7.11]))
    rrt_object.all_nodes = np.array([[1, 1, 1], [2, 2, 2], [11, 5, 5], [1, 1, 1]])
    rrt_object.neighborhood_radius = 3
    assert np.all(rrt_object._find_valid_neighbors(np.array([1, 1, 1])) == np.array([[1, 1, 1], [2, 2, 2], [1, 1, 1]]))
    rrt_object.obstacles = np.array([[4.999, 5.001, -10., 10., -10., 10.]])
    assert not rrt_object._is_valid_connection(np.array([0., 0., 0.]), np.array([10., 0., 0.]))
    rrt_object.obstacles = np.array([[4., 6., 1., 2., -10., 10.]])
    assert rrt_object._is_valid_connection(np.array([0., 0., 0.]), np.array([10., 0., 0.]))


def test_reference_unit_tests_tree_bookkeeping(rrt_object):
    start = rrt_object.start = np.array([0., 0., 0.])
    assert rrt_object._cost_to_come(start) == pytest.approx(0.0)
    rrt_object.tree = {"[0.0, 0.0, 2.0]": start, "[0.0, 2.0, 2.0]": np.array([0., 0., 2.])}
    assert rrt_object._cost_to_come(np.array([0., 2., 2.])) == pytest.approx(4.0)
    # best neighbor: lowest cost through the tree
    near, detour, costly = np.array([0., 0., 2.]), np.array([0., 3., 0.]), np.array([0., 0., 1.])
    rrt_object.tree = {"[0.0, 0.0, 2.0]": start, "[0.0, 3.0, 0.0]": start, "[0.0, 0.0, 1.0]": detour}
    assert np.all(rrt_object._find_best_neighbor([costly, near], np.array([0., 0., 3.])) == near)
    # rewiring
    new_node = np.array([0., 0., 1.])
    rrt_object.tree = {"[0.0, 0.0, 5.0]": start, "[1.0, 0.0, 0.0]": np.array([0., 0., 5.]),
                       "[0.0, 1.0, 0.0]": np.array([0., 0., 5.]), "[0.0, 0.0, 1.0]": start}
    assert rrt_object._rewire_safely([np.array([1., 0., 0.]), np.array([0., 1., 0.])], new_node)
    assert np.all(rrt_object.tree["[1.0, 0.0, 0.0]"] == new_node)
    assert np.all(rrt_object.tree["[0.0, 1.0, 0.0]"] == new_node)
    rrt_object.tree = {"[0.0, 0.0, 1.0]": start}
    assert not rrt_object._rewire_safely([start], new_node)
    assert "[0.0, 0.0, 0.0]" not in rrt_object.tree
    # update keeps the cheaper parent
    rrt_object.tree = {"[0.0, 0.0, 2.0]": start, "[0.0, 3.0, 0.0]": start}
    rrt_object._update_tree(np.array([0., 3., 0.]), np.array([0., 0., 2.]))
    assert np.all(rrt_object.tree["[0.0, 0.0, 2.0]"] == start)


def test_rrt_entry_points_reject_bad_arguments():
    """Negative UAVAC_E* codes with a message, never a crash: sizes < 1, null pointers, a non-positive step,
    non-finite host inputs."""
    import ctypes as C
    from uav_ac import _native as nat
    ctx = nat.Context(0)
    lib = nat.lib()
    P = nat.np_ptr
    s = np.zeros((1, 3)); g = np.ones((1, 3)); smp = np.zeros((1, 4, 3)); cap = 5
    nodes = np.empty((1, cap, 3)); path = np.empty((1, cap, 3))
    canon = np.empty((1, cap), np.int32); par = np.empty((1, cap), np.int32); bpar = np.empty((1, cap), np.int32)
    counts = np.empty((1, 6), np.int32); cost = np.empty(1)
    outs = (P(nodes), P(canon), P(par), P(bpar), P(path), P(counts), P(cost))

    def call(*a):
        return lib.uavac_rrt_star(ctx._h, *a)
    assert call(P(s), P(g), 1, 1.0, 4, P(smp), None, 0, *outs) == nat.OK
    assert call(P(s), P(g), 0, 1.0, 4, P(smp), None, 0, *outs) == nat.EINVAL
    assert call(P(s), P(g), 1, 1.0, 0, P(smp), None, 0, *outs) == nat.EINVAL
    assert call(P(s), P(g), 1, 0.0, 4, P(smp), None, 0, *outs) == nat.EINVAL
    assert call(P(s), P(g), 1, 1.0, 4, P(smp), None, 2, *outs) == nat.EINVAL          # obstacles announced, none given
    assert call(None, P(g), 1, 1.0, 4, P(smp), None, 0, *outs) == nat.EINVAL
    assert b"" != lib.uavac_last_error(ctx._h)
    bad = smp.copy(); bad[0, 2, 1] = np.nan
    assert call(P(s), P(g), 1, 1.0, 4, P(bad), None, 0, *outs) == nat.ENONFINITE
    hit = np.empty(2, np.int32)
    assert lib.uavac_rrt_segment_hits(ctx._h, P(np.zeros((2, 3))), P(np.ones((2, 3))), -1, None, 0, P(hit)) == nat.EINVAL
    assert lib.uavac_rrt_segment_hits(ctx._h, P(np.zeros((2, 3))), P(np.ones((2, 3))), 2, None, 1, P(hit)) == nat.EINVAL
    assert lib.uavac_rrt_segment_hits(ctx._h, P(np.zeros((2, 3))), P(np.ones((2, 3))), 0, None, 0, P(hit)) == nat.OK
    out = np.empty(1)
    assert lib.uavac_rrt_path_cost(ctx._h, None, 3, P(out)) == nat.EINVAL
    assert lib.uavac_rrt_path_cost(ctx._h, None, 0, P(out)) == nat.OK and out[0] == 0.0
    lens = np.array([9], np.int32); outp = np.empty((1, cap, 3)); outl = np.empty(1, np.int32)
    assert lib.uavac_rrt_simplify(ctx._h, P(path), P(lens), 1, cap, None, 0, P(outp), P(outl)) == nat.EINVAL   # longer than cap
    ctx.close()
