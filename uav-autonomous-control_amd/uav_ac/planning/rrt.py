"""RRT* planner with the surface of the reference's `uav_ac/planning/rrt.py` (`RRTStar`), computing on the GPU
through the C ABI, plus the batched form (`rrt_star_batch`: B planning problems, one wavefront each).

What stays on the host: drawing the random nodes (NumPy's legacy generator, in the reference's call order, so that
a seeded run reproduces the reference's node for node) and dictionary / list bookkeeping of the facade's helper
methods.  Every distance, steering step, segment-vs-cuboid test and the whole of `run()` execute in
`libuavac.so` (`uavac_rrt_*`); there is no CPU fallback.

Differences from the reference, all in reporting: `run()` does not print per-iteration progress (the iterations
happen inside one kernel), only the final cost line.  Two exception details differ on purpose:
* when no path is found, the reference's `run()` reaches `_is_path_found(None)` and dies with an `AttributeError`
  before its own `raise Exception("No path found")` (rrt.py:72-73) can run; here that intended exception is what is raised;
* a dictionary lookup of a key that is not in the tree raises `KeyError` upstream only if the new node is not value-equal
  to the start (or a non-start neighbour exists); the kernel reports KEY_ERROR for every such lookup.  The two agree
  except when a node equals the start by value but not bit for bit (-0.0 against 0.0), which `np.round(..., 2)` of drawn
  coordinates can produce only for a start on a coordinate plane.
"""
from __future__ import annotations

import copy
from dataclasses import dataclass

import numpy as np

from .. import _native as nat
from .._single import ctx

STATUS_OK, STATUS_NO_PATH, STATUS_COST_INCREASED, STATUS_KEY_ERROR = 0, 1, 2, 3
EPSILON = 0.15                                   # goal bias, rrt.py:19


# ----------------------------------------------------------------------------------------------- random nodes
def draw_random_nodes(random_sample, limits_lw, limits_up, goal, n: int, epsilon: float = EPSILON):
    """`n` consecutive results of RRTStar._generate_random_node (rrt.py:118-127) taken from one stream of uniform
    [0, 1) doubles: `random_sample(k)` must return the next k doubles of the generator (np.random.random_sample,
    or RandomState(seed).random_sample).  np.random.uniform(lo, hi) is lo + (hi - lo) * next_double.

    Returns (nodes (n, 3), consumed (n,)): consumed[i] = doubles used by draws 0..i, so that a caller that ends up
    needing only the first k draws can put the generator where the reference would have left it."""
    lw = np.asarray(limits_lw, dtype=np.float64)
    up = np.asarray(limits_up, dtype=np.float64)
    goal = np.asarray(goal, dtype=np.float64)
    stream = np.asarray(random_sample(4 * n), dtype=np.float64)
    nodes = np.empty((n, 3))
    consumed = np.empty(n, dtype=np.int64)
    pos = 0
    for i in range(n):
        if 0.0 + (1.0 - 0.0) * stream[pos] < epsilon:
            nodes[i] = goal
            pos += 1
        else:
            nodes[i] = np.round(lw + (up - lw) * stream[pos + 1:pos + 4], 2)
            pos += 4
        consumed[i] = pos
    return nodes, consumed


def draw_random_nodes_batch(seeds, limits_lw, limits_up, goals, n: int, epsilon: float = EPSILON) -> np.ndarray:
    """draw_random_nodes for B problems at once: problem b gets what the reference draws after
    `np.random.seed(seeds[b])`.  Each stream comes from its own RandomState; the 1-or-4 stride through the streams
    is walked for all problems together.  -> (B, n, 3)."""
    lw = np.asarray(limits_lw, dtype=np.float64)
    up = np.asarray(limits_up, dtype=np.float64)
    goals = np.asarray(goals, dtype=np.float64).reshape(-1, 3)
    B = goals.shape[0]
    if len(seeds) != B:
        raise ValueError("one seed per problem")
    streams = np.stack([np.random.RandomState(int(sd)).random_sample(4 * n) for sd in seeds])
    streams = np.concatenate([streams, np.zeros((B, 3))], axis=1)          # reads past the last used double are masked
    rows = np.arange(B)
    pos = np.zeros(B, dtype=np.int64)
    nodes = np.empty((B, n, 3))
    for i in range(n):
        take_goal = 0.0 + (1.0 - 0.0) * streams[rows, pos] < epsilon
        xyz = np.stack([streams[rows, pos + 1], streams[rows, pos + 2], streams[rows, pos + 3]], axis=1)
        nodes[:, i] = np.where(take_goal[:, None], goals, np.round(lw + (up - lw) * xyz, 2))
        pos += np.where(take_goal, 1, 4)
    return nodes


# ---------------------------------------------------------------------------------------------------- batched
@dataclass
class RRTBatch:
    """Results of B runs (NumPy, host).  Row b: `nodes[b, :n_nodes[b]]` = all_nodes; `canon`, `parent`,
    `best_parent` as in include/uavac.h; `best_path[b, :best_len[b]]` start -> goal."""
    nodes: np.ndarray
    canon: np.ndarray
    parent: np.ndarray
    best_parent: np.ndarray
    best_path: np.ndarray
    n_nodes: np.ndarray
    iterations: np.ndarray
    status: np.ndarray
    best_n: np.ndarray
    best_len: np.ndarray
    dynamic_it_counter: np.ndarray
    best_cost: np.ndarray

    def path(self, b: int) -> np.ndarray:
        return self.best_path[b, :self.best_len[b]].copy()

    def tree(self, b: int, best: bool = False) -> dict:
        """The reference's `tree` / `best_tree` dict {text key: parent coordinates} of run b."""
        n = int(self.best_n[b] if best else self.n_nodes[b])
        par = self.best_parent[b] if best else self.parent[b]
        out = {}
        for e in range(n):
            if self.canon[b, e] == e and par[e] >= 0:
                out[RRTStar._node_key(self.nodes[b, e])] = self.nodes[b, par[e]].copy()
        return out


def rrt_star_batch(starts, goals, max_distance: float, samples, obstacles=None, context: nat.Context | None = None
                   ) -> RRTBatch:
    """B independent `RRTStar(..., start[b], goal[b], max_distance, max_iterations, obstacles).run()` on the node
    sequences `samples` (B, max_iterations, 3) (see draw_random_nodes)."""
    starts = nat.as_f64(starts)
    goals = nat.as_f64(goals)
    samples = nat.as_f64(samples)
    if starts.ndim != 2 or starts.shape[1] != 3 or goals.shape != starts.shape:
        raise ValueError("starts and goals must both have shape (B, 3)")
    B = starts.shape[0]
    if samples.ndim != 3 or samples.shape[0] != B or samples.shape[2] != 3 or samples.shape[1] < 1:
        raise ValueError("samples must have shape (B, max_iterations, 3)")
    max_iter = samples.shape[1]
    cub = None if obstacles is None else nat.as_f64(np.asarray(obstacles, dtype=np.float64).reshape(-1, 6))
    n_obs = 0 if cub is None else cub.shape[0]
    cap = max_iter + 1
    nodes = np.empty((B, cap, 3)); path = np.empty((B, cap, 3))
    canon = np.empty((B, cap), np.int32); parent = np.empty((B, cap), np.int32); bparent = np.empty((B, cap), np.int32)
    counts = np.empty((B, 6), np.int32); cost = np.empty(B)
    (context or ctx()).call("uavac_rrt_star", nat.np_ptr(starts), nat.np_ptr(goals), B, float(max_distance), max_iter,
                            nat.np_ptr(samples), nat.np_ptr(cub) if n_obs else None, n_obs, nat.np_ptr(nodes),
                            nat.np_ptr(canon), nat.np_ptr(parent), nat.np_ptr(bparent), nat.np_ptr(path),
                            nat.np_ptr(counts), nat.np_ptr(cost))
    return RRTBatch(nodes, canon, parent, bparent, path, counts[:, 0].copy(), counts[:, 1].copy(), counts[:, 2].copy(),
                    counts[:, 3].copy(), counts[:, 4].copy(), counts[:, 5].copy(), cost)


# --------------------------------------------------------------------------------------------- GPU primitives
def _edge_lengths(p0, p1) -> np.ndarray:
    """np.linalg.norm(p1 - p0) per row; p1 one point or one per row."""
    p0 = nat.as_f64(np.asarray(p0, dtype=np.float64).reshape(-1, 3))
    p1 = nat.as_f64(np.asarray(p1, dtype=np.float64))
    single = int(p1.ndim == 1)
    if not single and p1.shape != p0.shape:
        raise ValueError("p1 must be one point or one point per edge")
    out = np.empty(p0.shape[0])
    ctx().call("uavac_rrt_edge_lengths", nat.np_ptr(p0), nat.np_ptr(p1), single, p0.shape[0], nat.np_ptr(out))
    return out


def _polyline_length(points) -> float:
    """Edge lengths of the polyline summed in order (RRTStar.path_cost's loop), on the GPU."""
    pts = nat.as_f64(np.asarray(points, dtype=np.float64).reshape(-1, 3))
    out = np.zeros(1)
    ctx().call("uavac_rrt_path_cost", nat.np_ptr(pts), pts.shape[0], nat.np_ptr(out))
    return out[0]


def _segment_hits(p0, p1, cuboids) -> np.ndarray:
    p0 = nat.as_f64(np.asarray(p0, dtype=np.float64).reshape(-1, 3))
    p1 = nat.as_f64(np.asarray(p1, dtype=np.float64).reshape(-1, 3))
    cub = nat.as_f64(np.asarray(cuboids, dtype=np.float64).reshape(-1, 6))
    hit = np.zeros(p0.shape[0], dtype=np.int32)
    if cub.shape[0]:
        ctx().call("uavac_rrt_segment_hits", nat.np_ptr(p0), nat.np_ptr(p1), p0.shape[0], nat.np_ptr(cub), cub.shape[0],
                   nat.np_ptr(hit))
    return hit.astype(bool)


class RRTStar:
    """
    Rapidly-exploring Random Tree (RRT*) algorithm -- reference uav_ac/planning/rrt.py:7-35, same constructor,
    attributes and methods.
    """

    def __init__(self, space_limits, start, goal, max_distance, max_iterations, obstacles=None):
        lower, upper = space_limits[0], space_limits[1]
        self.space_limits_lw, self.space_limits_up = lower, upper
        self.start, self.goal = np.round(start, 2), np.round(goal, 2)        # nodes live on a 0.01 grid
        self.step_size, self.max_iterations, self.obstacles = max_distance, max_iterations, obstacles
        self.epsilon = EPSILON
        self.neighborhood_radius = 1.5 * max_distance
        self.dynamic_it_counter, self.dynamic_break_at = 0, max_iterations / 10
        self.all_nodes, self.tree = [self.start], {}
        self.best_path = self.best_tree = None
        assert self.neighborhood_radius > self.step_size, "Neighborhood radius must be larger than step size"
        for name, point in (("start", self.start), ("goal", self.goal)):
            assert lower[2] <= point[2] <= upper[2], f"The z location of the {name} must be within the z space limits"

    # ------------------------------------------------------------------------------------------------ run
    def run(self):
        """rrt.py:37-80 in one kernel launch.  Draws from the global NumPy generator exactly what the reference
        would have drawn (and leaves it in the same state)."""
        state = np.random.get_state()
        nodes, consumed = draw_random_nodes(np.random.random_sample, self.space_limits_lw, self.space_limits_up,
                                            self.goal, int(self.max_iterations), self.epsilon)
        res = rrt_star_batch(np.asarray(self.start, dtype=np.float64)[None], np.asarray(self.goal, dtype=np.float64)[None],
                             self.step_size, nodes[None], self.obstacles)
        iters = int(res.iterations[0])
        np.random.set_state(state)
        if iters > 0:
            np.random.random_sample(int(consumed[iters - 1]))          # where the reference's generator would be

        n = int(res.n_nodes[0])
        self.all_nodes = [res.nodes[0, e].copy() for e in range(n)]
        self.tree = res.tree(0)
        self.dynamic_it_counter = int(res.dynamic_it_counter[0])
        status = int(res.status[0])
        if status == STATUS_COST_INCREASED:
            raise Exception("Cost increased after rewiring")
        if status == STATUS_KEY_ERROR:
            raise KeyError("tree lookup failed for a node that is neither the start nor in the tree")
        if status == STATUS_NO_PATH:
            raise Exception("No path found")
        self.best_tree = res.tree(0, best=True)
        self.best_path = res.path(0).reshape(-1, 3)
        print("\nBest path found with cost: {}".format(res.best_cost[0]))

    def store_best_tree(self):
        """
        Update the best tree with the current tree if the cost is lower
        """
        self.best_tree = copy.deepcopy(self.tree)

    @staticmethod
    def path_cost(path):
        """
        Calculate the cost of the path (rrt.py:84-91)
        """
        return _polyline_length(path)

    def simplify_path(self, path: np.ndarray) -> np.ndarray:
        """
        Remove waypoints bypassed by a collision-free direct connection (rrt.py:93-116).  All candidate shortcuts
        from the current waypoint are tested in one call; the farthest clear one is taken.
        """
        if len(path) <= 2:
            return np.asarray(path)
        path = np.asarray(path)
        simplified_path = [path[0]]
        current_index = 0
        while current_index < len(path) - 1:
            next_index = current_index + 1
            candidates = np.arange(len(path) - 1, current_index + 1, -1)
            if len(candidates):
                clear = self._valid_connections(np.repeat(path[current_index][None], len(candidates), axis=0),
                                                path[candidates])
                if clear.any():
                    next_index = int(candidates[np.argmax(clear)])
            simplified_path.append(path[next_index])
            current_index = next_index
        return np.asarray(simplified_path)

    # -------------------------------------------------------------------------------------------- helpers
    def _generate_random_node(self):
        """rrt.py:118-127 (host: this IS the random stream): goal with probability epsilon, else a grid point of
        the space; one uniform() call, then one per axis."""
        if np.random.uniform(0, 1) < self.epsilon:
            return self.goal
        xyz = [np.random.uniform(lo, hi) for lo, hi in zip(self.space_limits_lw[:3], self.space_limits_up[:3])]
        return np.round(np.array(xyz), 2)

    def _find_nearest_node(self, new_node):
        distances = _edge_lengths(np.asarray(self.all_nodes, dtype=np.float64), np.asarray(new_node, dtype=np.float64))
        return self.all_nodes[int(np.argmin(distances))]

    def _adapt_random_node_position(self, new_node, nearest_node):
        """
        Adapt the random node position if it is too far from the nearest node (rrt.py:140-148)
        """
        a = nat.as_f64(np.asarray(new_node, dtype=np.float64).reshape(1, 3))
        b = nat.as_f64(np.asarray(nearest_node, dtype=np.float64).reshape(1, 3))
        out = np.empty((1, 3))
        ctx().call("uavac_rrt_steer", nat.np_ptr(a), nat.np_ptr(b), 1, float(self.step_size), nat.np_ptr(out))
        if np.array_equal(out[0], a[0]):
            return new_node                                             # untouched, like the reference
        return out[0]

    def _valid_connections(self, p0, p1) -> np.ndarray:
        if self.obstacles is None:
            return np.ones(len(p0), dtype=bool)
        return ~_segment_hits(p0, p1, self.obstacles)

    def _find_valid_neighbors(self, new_node):
        nodes = np.asarray(self.all_nodes, dtype=np.float64).reshape(-1, 3)
        new = np.asarray(new_node, dtype=np.float64)
        in_radius = _edge_lengths(nodes, new) <= self.neighborhood_radius
        valid = self._valid_connections(nodes, np.repeat(new[None], len(nodes), axis=0))
        return [node for node, keep in zip(self.all_nodes, in_radius & valid) if keep]

    @staticmethod
    def _node_key(node: np.ndarray) -> str:
        return str(np.round(node, 2).tolist())

    def _cost_to_come(self, node: np.ndarray) -> float:
        """
        Cost of the path from the start to `node` following the tree edges (rrt.py:163-173): the chain node ->
        start is read off the dict, its length is accumulated on the GPU in walking order.
        """
        chain = [np.asarray(node, dtype=float)]
        while not np.array_equal(chain[-1], self.start):
            chain.append(np.asarray(self.tree[RRTStar._node_key(chain[-1])], dtype=float))
        return float(_polyline_length(chain))

    def _find_best_neighbor(self, neighbors, new_node):
        """
        The neighbor with the cheapest cost-to-come plus edge to the new node (rrt.py:175-186)
        """
        edges = _edge_lengths(np.asarray(neighbors, dtype=np.float64), np.asarray(new_node, dtype=np.float64))
        costs = [self._cost_to_come(neighbor) + edge for neighbor, edge in zip(neighbors, edges)]
        return neighbors[int(np.argmin(costs))]

    def _edge(self, a, b) -> float:
        return _edge_lengths(np.asarray(a, dtype=float).reshape(1, 3), np.asarray(b, dtype=float).reshape(3))[0]

    def _update_tree(self, node, new_node):
        """rrt.py:188-205: `node` becomes the parent of `new_node` unless they coincide or the tree already reaches
        `new_node` at no greater cost."""
        parent = np.round(node, 2)
        if np.array_equal(parent, new_node):
            return
        key = RRTStar._node_key(new_node)
        if key in self.tree and self._cost_to_come(new_node) <= self._cost_to_come(parent) + self._edge(parent, new_node):
            return
        self.tree[key] = parent
        self.all_nodes.append(new_node)

    def _rewire_safely(self, neighbors, new_node):
        """rrt.py:207-229: every neighbor other than the start and the new node's own parent is re-parented to
        the new node when that shortens its path; True when any was."""
        through_new = self._cost_to_come(new_node)
        rewired = False
        for nb in neighbors:
            if np.array_equal(nb, self.start) or np.array_equal(nb, self.tree[RRTStar._node_key(new_node)]):
                continue
            if through_new + self._edge(new_node, nb) < self._cost_to_come(nb):
                self.tree[RRTStar._node_key(nb)] = np.round(new_node, 2)
                rewired = True
        return rewired

    def _is_valid_connection(self, node, new_node):
        """
        Check if the connection between the candidate node and the new node is collision-free (rrt.py:231-243)
        """
        if self.obstacles is None:
            return True
        return bool(self._valid_connections(np.asarray(node, dtype=float)[None], np.asarray(new_node, dtype=float)[None])[0])

    @staticmethod
    def _segment_intersects_cuboid(node1: np.ndarray, node2: np.ndarray, cuboid: np.ndarray) -> bool:
        """
        Exact segment vs axis-aligned cuboid intersection test (slab method, rrt.py:245-274)
        """
        return bool(_segment_hits(np.asarray(node1, dtype=float)[None], np.asarray(node2, dtype=float)[None],
                                  np.asarray(cuboid, dtype=float)[None])[0])

    def _is_path_found(self, tree):
        """rrt.py:276-281: the goal has a parent in `tree`."""
        return RRTStar._node_key(self.goal) in tree

    def get_path(self, tree):
        """rrt.py:283-301: (path start -> goal as (n, 3), its cost).  The walk is bounded by the tree size instead
        of the reference's 5 s timer."""
        walk = [self.goal]
        while not np.array_equal(walk[-1], self.start):
            if len(walk) > len(tree) + 1:
                raise Exception("A problem occurred while computing the path, please restart the algorithm")
            walk.append(tree[RRTStar._node_key(walk[-1])])
        return np.array(walk[::-1]).reshape(-1, 3), RRTStar.path_cost(walk)
