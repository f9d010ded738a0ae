/*
 * rrt_oracle.c -- scalar C restatement of the reference's RRT* planner (uav_ac/planning/rrt.py).
 * TEST INFRASTRUCTURE, NOT PRODUCT: only tests/ may load it (through oracle/c_oracle.py).
 *
 * The reference keeps the tree in a dict keyed by the TEXT of the rounded coordinates and the
 * node list (`all_nodes`) as a Python list that can hold the same coordinates several times.
 * Restated with indices:
 *   entry e      one element of all_nodes, in insertion order (entry 0 = start)
 *   canon[e]     first entry with bit-identical coordinates = the dict key of that entry
 *                (text keys tell -0.0 from 0.0, so key identity is BITWISE; np.array_equal, which
 *                the reference uses for "is the start" / "is the parent", is VALUE identity)
 *   parent[c]    for a key c: the key of tree[c], or -1 when the dict has no such key
 * Random numbers are not drawn here: the caller passes, per iteration, the node that
 * RRTStar._generate_random_node (rrt.py:118-127) returned.
 *
 * np.linalg.norm of a 3-vector is sqrt(x.dot(x)); the BLAS ddot the reference ran on when the golden
 * vectors were made accumulates with fused multiply-adds, fma(x2,x2,fma(x1,x1,x0*x0)) -- norm3() states
 * that sequence explicitly (checked against np.linalg.norm by tests/golden/make_golden.py).
 * np.round(x, 2) is rint(x*100)/100.
 *
 * Pinned by tests/test_oracle_rrt.py against tests/golden/rrt_*.npz (made by importing the reference).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

enum { RRT_OK = 0, RRT_NO_PATH = 1, RRT_COST_INCREASED = 2, RRT_KEY_ERROR = 3 };

static double norm3(double x, double y, double z) { return sqrt(fma(z, z, fma(y, y, x * x))); }
static double round2(double x) { return rint(x * 100.0) / 100.0; }
static int bits_equal(const double *a, const double *b) { return memcmp(a, b, 3 * sizeof(double)) == 0; }
static int value_equal(const double *a, const double *b) { return a[0] == b[0] && a[1] == b[1] && a[2] == b[2]; }

/* RRTStar._segment_intersects_cuboid (rrt.py:245-274): slab test, segment n1 -> n2 */
int oracle_segment_intersects_cuboid(const double *n1, const double *n2, const double *c) {
    double t_min = 0.0, t_max = 1.0;
    for (int a = 0; a < 3; ++a) {
        const double d = n2[a] - n1[a], low = c[2 * a], high = c[2 * a + 1];
        if (fabs(d) < 1e-12) {
            if (n1[a] < low || n1[a] > high) return 0;
            continue;
        }
        double t_low = (low - n1[a]) / d, t_high = (high - n1[a]) / d;
        if (t_low > t_high) { const double t = t_low; t_low = t_high; t_high = t; }
        if (t_low > t_min) t_min = t_low;          /* Python max(t_min, t_low) */
        if (t_high < t_max) t_max = t_high;        /* Python min(t_max, t_high) */
        if (t_min > t_max) return 0;
    }
    return 1;
}

/* RRTStar._is_valid_connection (rrt.py:231-243) */
static int valid_connection(const double *n1, const double *n2, const double *cuboids, int n_obs) {
    for (int o = 0; o < n_obs; ++o)
        if (oracle_segment_intersects_cuboid(n1, n2, cuboids + 6 * o)) return 0;
    return 1;
}

typedef struct {
    const double *start;
    double *nodes;      /* [cap][3] */
    int *canon, *parent;
    int n;
    int key_error;
} tree_t;

/* RRTStar._cost_to_come (rrt.py:163-173), from key c */
static double cost_to_come(tree_t *t, int c) {
    double cost = 0.0;
    while (!value_equal(t->nodes + 3 * c, t->start)) {
        const int p = t->parent[c];
        if (p < 0) { t->key_error = 1; return cost; }
        const double *a = t->nodes + 3 * c, *b = t->nodes + 3 * p;
        cost += norm3(a[0] - b[0], a[1] - b[1], a[2] - b[2]);
        c = p;
    }
    return cost;
}

static int find_key(const tree_t *t, const double *x) {
    for (int e = 0; e < t->n; ++e)
        if (bits_equal(t->nodes + 3 * e, x)) return e;          /* first entry = canon */
    return -1;
}

/*
 * RRTStar.__init__ + run (rrt.py:12-80).  samples [max_iter][3]; cuboids [n_obs][6]; cap = max_iter + 1.
 * Out: n_nodes, nodes [cap][3], canon [cap], parent [cap] (final tree), best_n / best_parent (the tree
 * stored at the last improvement: entries < best_n), best_path [cap][3] (start -> goal), best_len,
 * best_cost, iters (iterations begun), dynamic_it_counter.  Returns RRT_*.
 */
int oracle_rrt_star(const double *start_in, const double *goal_in, double step, int max_iter, const double *samples,
                    const double *cuboids, int n_obs, int *n_nodes, double *nodes, int *canon, int *parent,
                    int *best_n, int *best_parent, int *best_len, double *best_path, double *best_cost,
                    int *iters, int *dynamic_it_counter) {
    const int cap = max_iter + 1;
    double start[3], goal[3];
    for (int a = 0; a < 3; ++a) { start[a] = round2(start_in[a]); goal[a] = round2(goal_in[a]); }
    const double radius = 1.5 * step;
    const double break_at = (double)max_iter / 10.0;
    tree_t t = {start, nodes, canon, parent, 0, 0};
    int *nbr = (int *)malloc(sizeof(int) * (size_t)cap);
    for (int e = 0; e < cap; ++e) { canon[e] = -1; parent[e] = -1; best_parent[e] = -1; }
    memcpy(nodes, start, sizeof start);
    canon[0] = 0;
    t.n = 1;
    *best_n = 0; *best_len = 0; *best_cost = INFINITY;
    double old_cost = INFINITY;
    int counter = 0, status = RRT_OK, it = 0, have_best = 0;

    for (it = 0; it < max_iter; ++it) {
        double nw[3] = {samples[3 * it], samples[3 * it + 1], samples[3 * it + 2]};
        /* _find_nearest_node (rrt.py:129-134): np.argmin = first minimum */
        int nearest = 0;
        double dmin = INFINITY;
        for (int e = 0; e < t.n; ++e) {
            const double *p = nodes + 3 * e;
            const double d = norm3(nw[0] - p[0], nw[1] - p[1], nw[2] - p[2]);
            if (d < dmin) { dmin = d; nearest = e; }
        }
        /* _adapt_random_node_position (rrt.py:140-148) */
        if (dmin > step) {
            const double *p = nodes + 3 * nearest;
            for (int a = 0; a < 3; ++a) nw[a] = round2(p[a] + (nw[a] - p[a]) * step / dmin);
        }
        /* _find_valid_neighbors (rrt.py:150-156) */
        int n_nbr = 0;
        for (int e = 0; e < t.n; ++e) {
            const double *p = nodes + 3 * e;
            if (norm3(p[0] - nw[0], p[1] - nw[1], p[2] - nw[2]) <= radius && valid_connection(p, nw, cuboids, n_obs))
                nbr[n_nbr++] = e;
        }
        if (n_nbr == 0) continue;
        /* _find_best_neighbor (rrt.py:175-186) */
        int best = nbr[0];
        double cbest = INFINITY;
        for (int i = 0; i < n_nbr; ++i) {
            const double *p = nodes + 3 * nbr[i];
            const double c = cost_to_come(&t, canon[nbr[i]]) + norm3(p[0] - nw[0], p[1] - nw[1], p[2] - nw[2]);
            if (c < cbest) { cbest = c; best = nbr[i]; }
        }
        /* _update_tree (rrt.py:188-205) */
        int key = find_key(&t, nw);
        const double *bp = nodes + 3 * best;
        if (!value_equal(bp, nw)) {
            int link = 1;
            if (key >= 0 && parent[key] >= 0) {
                const double current = cost_to_come(&t, key);
                const double cand = cost_to_come(&t, canon[best]) + norm3(nw[0] - bp[0], nw[1] - bp[1], nw[2] - bp[2]);
                if (current <= cand) link = 0;
            }
            if (link) {
                memcpy(nodes + 3 * t.n, nw, sizeof nw);
                if (key < 0) key = t.n;
                canon[t.n] = key;
                ++t.n;
                parent[key] = canon[best];
            }
        }
        /* _rewire_safely (rrt.py:207-229) */
        int has_rewired = 0;
        if (key < 0) {
            /* new_node equals its best neighbour by value but not by key, and has no entry of its own */
            t.key_error = 1;
        } else {
            const double new_cost = cost_to_come(&t, key);
            for (int i = 0; i < n_nbr && !t.key_error; ++i) {
                const double *p = nodes + 3 * nbr[i];
                if (value_equal(p, start)) continue;
                if (parent[key] < 0) { t.key_error = 1; break; }            /* self.tree[key(new_node)] */
                if (value_equal(p, nodes + 3 * parent[key])) continue;
                const double current = cost_to_come(&t, canon[nbr[i]]);
                const double through = new_cost + norm3(p[0] - nw[0], p[1] - nw[1], p[2] - nw[2]);
                if (through < current) { parent[canon[nbr[i]]] = key; has_rewired = 1; }
            }
        }
        if (t.key_error) { status = RRT_KEY_ERROR; ++it; break; }
        /* _is_path_found + get_path (rrt.py:276-301) */
        const int gk = find_key(&t, goal);
        if (gk >= 0 && parent[gk] >= 0) {
            const double cost = cost_to_come(&t, gk);       /* == path_cost of the goal -> start walk */
            if (has_rewired && cost > old_cost) { status = RRT_COST_INCREASED; ++it; break; }
            if (cost < old_cost) {
                memcpy(best_parent, parent, sizeof(int) * (size_t)cap);      /* store_best_tree */
                *best_n = t.n;
                old_cost = cost;
                counter = 0;
                have_best = 1;
            } else {
                ++counter;
            }
            if ((double)counter >= break_at) { ++it; break; }
        }
    }
    *iters = it;
    *dynamic_it_counter = counter;
    *n_nodes = t.n;
    if (status == RRT_OK && !have_best) status = RRT_NO_PATH;
    if (status == RRT_OK) {
        /* get_path(best_tree): goal -> start, reversed */
        int len = 0, c = find_key(&t, goal);
        int *chain = nbr;
        while (1) {
            chain[len++] = c;
            if (value_equal(nodes + 3 * c, start)) break;
            c = best_parent[c];
        }
        double cost = 0.0;
        for (int i = 0; i + 1 < len; ++i) {
            const double *a = nodes + 3 * chain[i], *b = nodes + 3 * chain[i + 1];
            cost += norm3(b[0] - a[0], b[1] - a[1], b[2] - a[2]);
        }
        for (int i = 0; i < len; ++i) memcpy(best_path + 3 * i, nodes + 3 * chain[len - 1 - i], 3 * sizeof(double));
        *best_len = len;
        *best_cost = cost;
    }
    free(nbr);
    return status;
}

/* distances of `n` nodes to a query point, the way _find_nearest_node / _find_valid_neighbors take them */
void oracle_rrt_distances(const double *nodes, int n, const double *q, double *out) {
    for (int e = 0; e < n; ++e) out[e] = norm3(q[0] - nodes[3 * e], q[1] - nodes[3 * e + 1], q[2] - nodes[3 * e + 2]);
}

/* np.linalg.norm(p1[e] - p0[e]) for E edges (p1 one point for all when p1_single) */
void oracle_rrt_edge_lengths(const double *p0, const double *p1, int p1_single, int n, double *out) {
    for (int e = 0; e < n; ++e) {
        const double *a = p0 + 3 * e, *b = p1 + (p1_single ? 0 : 3 * e);
        out[e] = norm3(b[0] - a[0], b[1] - a[1], b[2] - a[2]);
    }
}

/* RRTStar._adapt_random_node_position (rrt.py:140-148) */
void oracle_rrt_steer(const double *sample, const double *nearest, double step, double *out) {
    const double d = norm3(sample[0] - nearest[0], sample[1] - nearest[1], sample[2] - nearest[2]);
    for (int a = 0; a < 3; ++a)
        out[a] = d > step ? round2(nearest[a] + (sample[a] - nearest[a]) * step / d) : sample[a];
}

/* hit[e] = 1 when segment p0[e] -> p1[e] crosses any of the n_obs cuboids (not a valid connection) */
void oracle_rrt_segment_hits(const double *p0, const double *p1, int n, const double *cuboids, int n_obs, int *hit) {
    for (int e = 0; e < n; ++e) hit[e] = !valid_connection(p0 + 3 * e, p1 + 3 * e, cuboids, n_obs);
}
