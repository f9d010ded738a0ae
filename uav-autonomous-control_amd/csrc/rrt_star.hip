// RRT* planner, batched: one wavefront per planning problem (gfx950).
//
// Restates uav_ac/planning/rrt.py (RRTStar.run and its helpers) with the tree held in arrays instead of a
// dict keyed by coordinate text:
//   entry e     one element of `all_nodes` in insertion order (entry 0 = start; coordinates may repeat)
//   canon[e]    first entry with BIT-identical coordinates = the dict key of that entry (text keys tell
//               -0.0 from 0.0; np.array_equal, used for "is the start" / "is the parent", is VALUE identity)
//   parent[c]   key of tree[c], -1 when the dict has no key c
// The 64 lanes share every O(all_nodes) step of an iteration: the nearest-node scan (rrt.py:129-134), the
// neighbourhood scan with the segment-vs-cuboid slab test per candidate edge (:150-156, :231-274), the
// cost-to-come walks of all neighbours at once (:163-186) and the re-wiring test (:207-229), which is applied in
// list order exactly as the sequential loop would (first improving neighbour, then the ones after it are
// re-evaluated on the modified tree).  Tree, edge lengths and the neighbour list live in LDS (52 B per node)
// when they fit, in HBM scratch otherwise.
//
// Arithmetic that decides branches is the reference's, operation for operation: np.linalg.norm of a 3-vector
// is sqrt(x.dot(x)) with the dot product accumulated by fused multiply-adds (BLAS ddot), np.round(x, 2) is
// rint(100 x) / 100, IEEE division and sqrt.  No FMA contraction anywhere else.
// Random numbers are drawn on the host (NumPy's legacy global generator, in the reference's call order): the
// kernel receives, per iteration, the node RRTStar._generate_random_node returned.
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "uavac_internal.h"

#pragma clang fp contract(off)

namespace {

constexpr int W = 64;
enum { RRT_OK = 0, RRT_NO_PATH = 1, RRT_COST_INCREASED = 2, RRT_KEY_ERROR = 3, RRT_OVERFLOW = 4 /* internal: rerun with full capacity */ };

__device__ __forceinline__ double norm3(double x, double y, double z) { return sqrt(fma(z, z, fma(y, y, x * x))); }
__device__ __forceinline__ double round2(double x) { return rint(x * 100.0) / 100.0; }
__device__ __forceinline__ bool bits_equal(double a, double b) {
    return __double_as_longlong(a) == __double_as_longlong(b);
}

// RRTStar._segment_intersects_cuboid (rrt.py:245-274)
__device__ __forceinline__ bool slab_hit(double a0, double a1, double a2, double b0, double b1, double b2,
                                         const double *__restrict__ c) {
    const double n1[3] = {a0, a1, a2}, n2[3] = {b0, b1, b2};
    double t_min = 0.0, t_max = 1.0;
    bool hit = true;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        if (!hit) break;
        const double d = n2[a] - n1[a], low = c[2 * a], high = c[2 * a + 1];
        if (fabs(d) < 1e-12) {
            if (n1[a] < low || n1[a] > high) hit = false;
            continue;
        }
        double t_low = (low - n1[a]) / d, t_high = (high - n1[a]) / d;
        if (t_low > t_high) { const double t = t_low; t_low = t_high; t_high = t; }
        if (t_low > t_min) t_min = t_low;
        if (t_high < t_max) t_max = t_high;
        if (t_min > t_max) hit = false;
    }
    return hit;
}

// Both end points beyond the same face of the cuboid by a clear margin: the slab test below is certain to say
// "no intersection" (t_low > 1 or t_high < 0 on that axis by far more than rounding can undo), so its six
// divisions are skipped.  The margin keeps the shortcut exact: inside it the full test decides.
__device__ __forceinline__ bool clearly_apart(double a, double b, double low, double high) {
    const double margin = 1e-6 + 1e-9 * fabs(b - a);
    return (a < low - margin && b < low - margin) || (a > high + margin && b > high + margin);
}

// RRTStar._is_valid_connection (rrt.py:231-243)
__device__ __forceinline__ bool valid_connection(double a0, double a1, double a2, double b0, double b1, double b2,
                                                 const double *__restrict__ cuboids, int n_obs) {
    for (int o = 0; o < n_obs; ++o) {
        const double *c = cuboids + 6 * o;
        if (clearly_apart(a0, b0, c[0], c[1]) || clearly_apart(a1, b1, c[2], c[3]) || clearly_apart(a2, b2, c[4], c[5]))
            continue;
        if (slab_hit(a0, a1, a2, b0, b1, b2, c)) return false;
    }
    return true;
}

struct Tree {
    double *nodes;   // [cap][3]
    int *canon;      // [cap]
    int *par;        // [cap] the dict (HBM, the `parent` output): parent key of a key, -1 when there is no such key
    int *up;         // [cap] walk link of a key: parent key, -1 no entry, -2 the key is (by value) the start
    double *elen;    // [cap] |key - parent(key)|
    double *cc;      // [cap] cost-to-come of a key as last walked, NaN = not known for the present tree
    int *nbr;        // [cap] neighbour list of the iteration
    int cap;
};

// RRTStar._cost_to_come (rrt.py:163-173) from key c; key_err: the dict lookup would have raised
__device__ __forceinline__ double cost_to_come(const Tree &t, int c, bool &key_err) {
    double cost = 0.0;
    for (int hops = 0;; ++hops) {
        const int u = t.up[c];
        if (u == -2) break;
        if (u < 0 || hops > t.cap) { key_err = true; break; }      // (a cycle cannot form; the bound is a fuse)
        cost += t.elen[c];
        c = u;
    }
    return cost;
}

// The same value without the walk when this key was walked since the tree last changed under an existing key.
// (A cost-to-come is the edge lengths summed from the node towards the start, so it cannot be derived from the
// parent's value -- different association -- but it can be remembered while the chain stays as it is.)
__device__ __forceinline__ double cost_to_come_cached(const Tree &t, int c, bool &key_err) {
    double v = t.cc[c];
    if (v == v) return v;
    v = cost_to_come(t, c, key_err);
    t.cc[c] = v;
    return v;
}

__device__ __forceinline__ void forget_costs(const Tree &t, int n, int lane) {
    __syncthreads();
    for (int e = lane; e < n; e += W) t.cc[e] = __builtin_nan("");
    __syncthreads();
}

__device__ __forceinline__ int first_lane(unsigned long long m) { return __ffsll((long long)m) - 1; }

// (value, index) arg-min over the wave, first index on ties
__device__ __forceinline__ void wave_argmin(double &d, int &e) {
#pragma unroll
    for (int s = 1; s < W; s <<= 1) {
        const double od = __shfl_xor(d, s, W);
        const int oe = __shfl_xor(e, s, W);
        if (od < d || (od == d && oe < e)) { d = od; e = oe; }
    }
}

template <bool USE_LDS>
__global__ void __launch_bounds__(W)
rrt_star_kernel(const double *__restrict__ starts, const double *__restrict__ goals, int B, double step, int max_iter,
                const double *__restrict__ samples, const double *__restrict__ cuboids, int n_obs,
                double *__restrict__ g_nodes, int32_t *__restrict__ g_canon, int32_t *__restrict__ g_parent,
                int32_t *__restrict__ g_best_parent, double *__restrict__ g_best_path, int32_t *__restrict__ counts,
                double *__restrict__ best_cost_out, double *__restrict__ scratch, int node_cap, int pass) {
    // Two passes share this kernel.  Pass 0 gives every problem LDS for `node_cap` nodes only (a typical tree uses a
    // fraction of max_iter + 1), so several times more problems are resident per CU; a problem whose tree outgrows
    // that stops with RRT_OVERFLOW.  Pass 1 (full capacity) reruns exactly those from the start; every other
    // workgroup leaves at once.  Results do not depend on the split: a run is deterministic.
    constexpr bool use_lds = USE_LDS;
    if (pass == 1 && counts[6 * blockIdx.x + 2] != RRT_OVERFLOW) return;
    extern __shared__ double lds[];
    const int b = blockIdx.x, lane = threadIdx.x;
    const int cap = max_iter + 1;
    const size_t capz = (size_t)cap;

    Tree t;
    t.cap = cap;
    int32_t *best_parent = g_best_parent + b * capz;
    t.par = g_parent + b * capz;
    if (use_lds) {
        const size_t ncz = (size_t)node_cap;
        t.nodes = lds;
        t.elen = lds + 3 * ncz;
        t.cc = lds + 4 * ncz;
        t.canon = reinterpret_cast<int *>(lds + 5 * ncz);
        t.up = t.canon + ncz;
        t.nbr = t.up + ncz;
    } else {
        double *ws = scratch + (size_t)b * 3 * capz;       // elen, cc [cap] f64, up, nbr [cap] i32
        t.nodes = g_nodes + b * 3 * capz;
        t.canon = g_canon + b * capz;
        t.elen = ws;
        t.cc = ws + capz;
        t.up = reinterpret_cast<int *>(ws + 2 * capz);
        t.nbr = t.up + capz;
    }
    int *parent = t.par;

    double start[3], goal[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) { start[a] = round2(starts[3 * b + a]); goal[a] = round2(goals[3 * b + a]); }
    const double radius = 1.5 * step;
    const double break_at = (double)max_iter / 10.0;

    for (int e = lane; e < cap; e += W) { parent[e] = -1; best_parent[e] = -1; }
    for (int e = lane; e < node_cap; e += W) { t.up[e] = -1; t.canon[e] = -1; }
    __syncthreads();
    if (lane == 0) {
        t.nodes[0] = start[0]; t.nodes[1] = start[1]; t.nodes[2] = start[2];
        t.canon[0] = 0;
        t.up[0] = -2;
        t.elen[0] = 0.0;
        t.cc[0] = 0.0;
    }
    __syncthreads();

    int n = 1;                                             // entries in all_nodes
    int goal_key = (bits_equal(start[0], goal[0]) && bits_equal(start[1], goal[1]) && bits_equal(start[2], goal[2])) ? 0 : -1;
    double old_cost = INFINITY;
    int counter = 0, status = RRT_OK, best_n = 0, it = 0;
    bool have_best = false, goal_linked = false;
    const double *smp = samples + (size_t)b * max_iter * 3;

    for (it = 0; it < max_iter; ++it) {
        double nw0 = smp[3 * it], nw1 = smp[3 * it + 1], nw2 = smp[3 * it + 2];

        // ---- _find_nearest_node (rrt.py:129-134): first minimum
        double dmin = INFINITY;
        int nearest = 0x7fffffff;
        for (int e = lane; e < n; e += W) {
            const double *p = t.nodes + 3 * e;
            const double d = norm3(nw0 - p[0], nw1 - p[1], nw2 - p[2]);
            if (d < dmin) { dmin = d; nearest = e; }
        }
        wave_argmin(dmin, nearest);
        if (nearest == 0x7fffffff) nearest = 0;           // every distance NaN: np.argmin of all-NaN is 0

        // ---- _adapt_random_node_position (rrt.py:140-148)
        if (dmin > step) {
            const double *p = t.nodes + 3 * nearest;
            nw0 = round2(p[0] + (nw0 - p[0]) * step / dmin);
            nw1 = round2(p[1] + (nw1 - p[1]) * step / dmin);
            nw2 = round2(p[2] + (nw2 - p[2]) * step / dmin);
        }

        // ---- _find_valid_neighbors (rrt.py:150-156) + the dict key of the new node.  Two passes: the radius test
        // over all entries compacts the few candidates (in list order) into t.nbr, then the segment-vs-cuboid
        // tests run on full lanes of candidates only and compact in place.
        int n_cand = 0, key = -1;
        for (int base = 0; base < n; base += W) {
            const int e = base + lane;
            bool in = false, same = false;
            if (e < n) {
                const double *p = t.nodes + 3 * e;
                const double p0 = p[0], p1 = p[1], p2 = p[2];
                same = bits_equal(p0, nw0) && bits_equal(p1, nw1) && bits_equal(p2, nw2);
                in = norm3(p0 - nw0, p1 - nw1, p2 - nw2) <= radius;
            }
            const unsigned long long m_in = __ballot(in), m_same = __ballot(same);
            if (in) t.nbr[n_cand + __popcll(m_in & ((1ull << lane) - 1ull))] = e;
            n_cand += __popcll(m_in);
            if (key < 0 && m_same) key = base + first_lane(m_same);
        }
        __syncthreads();
        int n_nbr = n_cand;
        if (n_obs > 0) {
            n_nbr = 0;
            for (int base = 0; base < n_cand; base += W) {
                const int i = base + lane;
                bool ok = false;
                int e = 0;
                if (i < n_cand) {
                    e = t.nbr[i];
                    const double *p = t.nodes + 3 * e;
                    ok = valid_connection(p[0], p[1], p[2], nw0, nw1, nw2, cuboids, n_obs);
                }
                const unsigned long long m_ok = __ballot(ok);
                __syncthreads();                                  // every lane holds its e before slots are rewritten
                if (ok) t.nbr[n_nbr + __popcll(m_ok & ((1ull << lane) - 1ull))] = e;
                n_nbr += __popcll(m_ok);
            }
        }
        __syncthreads();
        if (n_nbr == 0) continue;

        // ---- _find_best_neighbor (rrt.py:175-186): first minimum of cost-to-come + edge
        bool kerr = false;
        double cbest = INFINITY;
        int ibest = 0x7fffffff;
        for (int base = 0; base < n_nbr; base += W) {
            const int i = base + lane;
            if (i < n_nbr) {
                const int e = t.nbr[i];
                const double *p = t.nodes + 3 * e;
                const double c = cost_to_come_cached(t, t.canon[e], kerr) + norm3(p[0] - nw0, p[1] - nw1, p[2] - nw2);
                if (c < cbest) { cbest = c; ibest = i; }
            }
        }
        wave_argmin(cbest, ibest);
        __syncthreads();                                  // costs remembered by other lanes are visible from here on
        if (ibest == 0x7fffffff) ibest = 0;
        const int best = t.nbr[ibest];
        const double bp0 = t.nodes[3 * best], bp1 = t.nodes[3 * best + 1], bp2 = t.nodes[3 * best + 2];
        const int best_key = t.canon[best];

        // ---- _update_tree (rrt.py:188-205).  pk = key of tree[key(new_node)] afterwards (-1: no such entry)
        int pk = -1;
        bool linked = false;
        if (!(bp0 == nw0 && bp1 == nw1 && bp2 == nw2)) {
            bool link = true;
            const double edge = norm3(nw0 - bp0, nw1 - bp1, nw2 - bp2);
            if (key >= 0 && parent[key] >= 0) {
                const double current = cost_to_come_cached(t, key, kerr);
                const double cand = cost_to_come_cached(t, best_key, kerr) + edge;
                if (current <= cand) link = false;
            }
            if (link && n >= node_cap) { status = RRT_OVERFLOW; break; }      // pass 0 only: node_cap == cap otherwise
            if (link) {
                const bool fresh = key < 0;
                if (fresh) key = n;
                __syncthreads();
                if (lane == 0) {
                    t.nodes[3 * n] = nw0; t.nodes[3 * n + 1] = nw1; t.nodes[3 * n + 2] = nw2;
                    t.canon[n] = key;
                    parent[key] = best_key;
                    t.up[key] = (nw0 == start[0] && nw1 == start[1] && nw2 == start[2]) ? -2 : best_key;
                    t.elen[key] = edge;
                    t.cc[key] = __builtin_nan("");
                }
                if (fresh && bits_equal(nw0, goal[0]) && bits_equal(nw1, goal[1]) && bits_equal(nw2, goal[2]) && goal_key < 0)
                    goal_key = key;
                ++n;
                __syncthreads();
                if (!fresh) forget_costs(t, n, lane);                 // an existing key changed parent
                linked = true;
                pk = best_key;
            }
        }
        if (!linked && key >= 0) pk = parent[key];
        if (key >= 0 && key == goal_key && pk >= 0) goal_linked = true;

        // ---- _rewire_safely (rrt.py:207-229)
        bool has_rewired = false;
        if (key < 0) {
            kerr = true;            // the new node equals its best neighbour by value, not by key: no dict entry
        } else {
            const double new_cost = cost_to_come_cached(t, key, kerr);
            double q0 = 0.0, q1 = 0.0, q2 = 0.0;
            if (pk >= 0) { q0 = t.nodes[3 * pk]; q1 = t.nodes[3 * pk + 1]; q2 = t.nodes[3 * pk + 2]; }
            for (int base = 0; base < n_nbr; base += W) {
                const int i = base + lane;
                bool cand = false;
                int c = 0;
                double through = 0.0, edge = 0.0;
                if (i < n_nbr) {
                    const int e = t.nbr[i];
                    const double *p = t.nodes + 3 * e;
                    const double p0 = p[0], p1 = p[1], p2 = p[2];
                    const bool is_start = (p0 == start[0] && p1 == start[1] && p2 == start[2]);
                    if (!is_start) {
                        if (pk < 0) kerr = true;                      // KeyError on the first neighbour that is not the start
                        else if (!(p0 == q0 && p1 == q1 && p2 == q2)) {
                            cand = true;
                            c = t.canon[e];
                            edge = norm3(p0 - nw0, p1 - nw1, p2 - nw2);
                            through = new_cost + edge;
                        }
                    }
                }
                int done_upto = -1;                                   // lanes <= done_upto have had their turn
                while (true) {
                    bool want = false;
                    if (cand && lane > done_upto) want = through < cost_to_come_cached(t, c, kerr);
                    const unsigned long long m = __ballot(want);
                    if (!m) break;
                    const int L = first_lane(m);
                    __syncthreads();
                    if (lane == L) { parent[c] = key; t.up[c] = key; t.elen[c] = edge; }
                    has_rewired = true;
                    done_upto = L;
                    forget_costs(t, n, lane);                         // an existing key changed parent
                }
            }
        }
        if (__ballot(kerr)) { status = RRT_KEY_ERROR; ++it; break; }

        // ---- _is_path_found + get_path (rrt.py:276-301)
        if (goal_linked) {
            const double cost = cost_to_come_cached(t, goal_key, kerr);
            if (has_rewired && cost > old_cost) { status = RRT_COST_INCREASED; ++it; break; }
            if (cost < old_cost) {
                __syncthreads();
                for (int e = lane; e < n; e += W) best_parent[e] = parent[e];         // store_best_tree
                best_n = n;
                old_cost = cost;
                counter = 0;
                have_best = true;
                __syncthreads();
            } else {
                ++counter;
            }
            if ((double)counter >= break_at) { ++it; break; }
        }
    }

    if (status == RRT_OVERFLOW) {               // nothing of this attempt is kept
        if (lane == 0) counts[6 * b + 2] = RRT_OVERFLOW;
        return;
    }
    if (status == RRT_OK && !have_best) status = RRT_NO_PATH;
    __syncthreads();
    if (use_lds) {
        double *gn = g_nodes + b * 3 * capz;
        for (int i = lane; i < 3 * n; i += W) gn[i] = t.nodes[i];
        for (int i = 3 * n + lane; i < 3 * cap; i += W) gn[i] = 0.0;
        int32_t *gc = g_canon + b * capz;
        for (int e = lane; e < cap; e += W) gc[e] = e < node_cap ? t.canon[e] : -1;
    } else {
        for (int i = 3 * n + lane; i < 3 * cap; i += W) t.nodes[i] = 0.0;
    }
    __syncthreads();

    int best_len = 0;
    double best_cost = INFINITY;
    double *path = g_best_path + b * 3 * capz;
    if (status == RRT_OK) {
        // get_path(best_tree): goal -> start, then reversed; path_cost sums in goal -> start order
        if (lane == 0) {
            int len = 0, c = goal_key;
            double cost = 0.0;
            while (len < cap) {
                t.nbr[len++] = c;
                if (t.up[c] == -2) break;
                const int p = best_parent[c];
                const double *x = t.nodes + 3 * c, *y = t.nodes + 3 * p;
                cost += norm3(y[0] - x[0], y[1] - x[1], y[2] - x[2]);
                c = p;
            }
            best_len = len;
            best_cost = cost;
        }
        best_len = __shfl(best_len, 0, W);
        best_cost = __shfl(best_cost, 0, W);
        __syncthreads();
        for (int i = lane; i < best_len; i += W) {
            const int c = t.nbr[best_len - 1 - i];
            path[3 * i] = t.nodes[3 * c]; path[3 * i + 1] = t.nodes[3 * c + 1]; path[3 * i + 2] = t.nodes[3 * c + 2];
        }
    }
    for (int i = 3 * best_len + lane; i < 3 * cap; i += W) path[i] = 0.0;
    if (lane == 0) {
        counts[6 * b + 0] = n;
        counts[6 * b + 1] = it;
        counts[6 * b + 2] = status;
        counts[6 * b + 3] = best_n;
        counts[6 * b + 4] = best_len;
        counts[6 * b + 5] = counter;
        best_cost_out[b] = best_cost;
    }
}

// E candidate edges x n_obs cuboids -> hit[e] = 1 when edge e crosses any cuboid (not a valid connection)
__global__ void rrt_segment_hits_kernel(const double *__restrict__ p0, const double *__restrict__ p1, int E,
                                        const double *__restrict__ cuboids, int n_obs, int32_t *__restrict__ hit) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= E) return;
    const double *a = p0 + 3 * (size_t)e, *b = p1 + 3 * (size_t)e;
    hit[e] = valid_connection(a[0], a[1], a[2], b[0], b[1], b[2], cuboids, n_obs) ? 0 : 1;
}

// len[e] = np.linalg.norm(p1[e] - p0[e]); p1 is one point for all edges when p1_stride == 0
__global__ void rrt_edge_lengths_kernel(const double *__restrict__ p0, const double *__restrict__ p1, int p1_stride,
                                        int E, double *__restrict__ out) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= E) return;
    const double *a = p0 + 3 * (size_t)e, *b = p1 + (size_t)p1_stride * e;
    out[e] = norm3(b[0] - a[0], b[1] - a[1], b[2] - a[2]);
}

// The node sequence RRTStar._generate_random_node (rrt.py:118-127) produces after np.random.seed(seed), drawn on the
// GPU: NumPy's legacy generator is MT19937 seeded by init_genrand (integer seeds), random_sample() is the 53-bit
// double built from two outputs, uniform(lo, hi) = lo + (hi - lo) * random_sample(), np.round(x, 2) = rint(100 x) / 100.
// One wavefront per problem: the 64 lanes regenerate the 624-word state together (the three dependency-free
// ranges of the recurrence), temper and pair the words into 312 doubles in LDS, and every lane then walks the
// stream in step (one double for the goal bias, three more unless the goal was drawn) -- uniform control flow, lane 0
// writes.  All integer / exactly rounded arithmetic: the result is NumPy's, bit for bit.
__device__ __forceinline__ unsigned mt_mix(unsigned u, unsigned v, unsigned far) {
    const unsigned y = (u & 0x80000000u) | (v & 0x7fffffffu);
    return far ^ (y >> 1) ^ ((v & 1u) ? 0x9908b0dfu : 0u);
}

__global__ void __launch_bounds__(W)
rrt_draw_nodes_kernel(const unsigned *__restrict__ seeds, const double *__restrict__ goals, int B, int n, double lw0,
                      double lw1, double lw2, double up0, double up1, double up2, double epsilon,
                      double *__restrict__ samples, int64_t *__restrict__ consumed) {
    __shared__ unsigned mt[624];
    __shared__ double dbl[312];
    const int b = blockIdx.x, lane = threadIdx.x;
    // init_genrand: sequential by nature (lane 0), 624 steps once per problem
    if (lane == 0) {
        unsigned sd = seeds[b];
        for (int i = 0; i < 624; ++i) { mt[i] = sd; sd = 1812433253u * (sd ^ (sd >> 30)) + (unsigned)i + 1u; }
    }
    __syncthreads();
    const double g0 = goals[3 * b], g1 = goals[3 * b + 1], g2 = goals[3 * b + 2];
    const double r0 = up0 - lw0, r1 = up1 - lw1, r2 = up2 - lw2;
    double *out = samples + (size_t)b * n * 3;
    int64_t *cons = consumed ? consumed + (size_t)b * n : nullptr;
    int pos = 312;                                   // doubles of the current block already used (312 = none left)
    int64_t used = 0;
    auto refill = [&]() {
        // new[i] = old[i+397] ^ f(old[i], old[i+1]) for i < 227; new[i] = new[i-227] ^ f(old[i], old[i+1]) beyond
        for (int base = 0; base < 227; base += W) {
            const int i = base + lane;
            unsigned v = 0;
            if (i < 227) v = mt_mix(mt[i], mt[i + 1], mt[i + 397]);
            __syncthreads();
            if (i < 227) mt[i] = v;
            __syncthreads();
        }
        for (int lo = 227; lo < 623; lo += 227) {                     // [227, 454), [454, 623): each reads the range before it
            const int hi = lo + 227 < 623 ? lo + 227 : 623;
            for (int base = lo; base < hi; base += W) {
                const int i = base + lane;
                unsigned v = 0;
                if (i < hi) v = mt_mix(mt[i], mt[i + 1], mt[i - 227]);
                __syncthreads();
                if (i < hi) mt[i] = v;
                __syncthreads();
            }
        }
        if (lane == 0) mt[623] = mt_mix(mt[623], mt[0], mt[396]);
        __syncthreads();
        for (int k = lane; k < 312; k += W) {                         // genrand_res53 on tempered pairs
            unsigned a = mt[2 * k], c = mt[2 * k + 1];
            a ^= a >> 11; a ^= (a << 7) & 0x9d2c5680u; a ^= (a << 15) & 0xefc60000u; a ^= a >> 18;
            c ^= c >> 11; c ^= (c << 7) & 0x9d2c5680u; c ^= (c << 15) & 0xefc60000u; c ^= c >> 18;
            dbl[k] = ((double)(a >> 5) * 67108864.0 + (double)(c >> 6)) / 9007199254740992.0;
        }
        __syncthreads();
        pos = 0;
    };
    auto next = [&]() {
        if (pos >= 312) refill();
        ++used;
        return dbl[pos++];
    };
    for (int it = 0; it < n; ++it) {
        const double u = next();
        double x = g0, y = g1, z = g2;
        if (!(0.0 + (1.0 - 0.0) * u < epsilon)) {
            x = round2(lw0 + r0 * next());
            y = round2(lw1 + r1 * next());
            z = round2(lw2 + r2 * next());
        }
        if (lane == 0) {
            out[3 * it] = x; out[3 * it + 1] = y; out[3 * it + 2] = z;
            if (cons) cons[it] = used;
        }
    }
}

// RRTStar.simplify_path (rrt.py:93-116) for B paths, one wavefront each: from the current waypoint the lanes test the
// direct connection to the last, second-to-last, ... waypoint at once; the farthest clear one is the next waypoint.
__global__ void __launch_bounds__(W)
rrt_simplify_kernel(const double *__restrict__ paths, const int32_t *__restrict__ lens, int cap,
                    const double *__restrict__ cuboids, int n_obs, double *__restrict__ out, int32_t *__restrict__ out_lens) {
    const int b = blockIdx.x, lane = threadIdx.x;
    const double *path = paths + (size_t)b * cap * 3;
    double *dst = out + (size_t)b * cap * 3;
    const int L = lens[b];
    int n_out = 0;
    if (L <= 2) {
        for (int i = lane; i < 3 * L; i += W) dst[i] = path[i];
        n_out = L > 0 ? L : 0;
    } else {
        int cur = 0;
        if (lane == 0) { dst[0] = path[0]; dst[1] = path[1]; dst[2] = path[2]; }
        n_out = 1;
        while (cur < L - 1) {
            const double a0 = path[3 * cur], a1 = path[3 * cur + 1], a2 = path[3 * cur + 2];
            int next = cur + 1;
            for (int top = L - 1; top > cur + 1; top -= W) {
                const int j = top - lane;
                bool clear = false;
                if (j > cur + 1) clear = valid_connection(a0, a1, a2, path[3 * j], path[3 * j + 1], path[3 * j + 2], cuboids, n_obs);
                const unsigned long long m = __ballot(clear);
                if (m) { next = top - first_lane(m); break; }
            }
            if (lane == 0) { dst[3 * n_out] = path[3 * next]; dst[3 * n_out + 1] = path[3 * next + 1]; dst[3 * n_out + 2] = path[3 * next + 2]; }
            ++n_out;
            cur = next;
        }
    }
    for (int i = 3 * n_out + lane; i < 3 * cap; i += W) dst[i] = 0.0;
    if (lane == 0) out_lens[b] = n_out;
}

// RRTStar.path_cost (rrt.py:84-91): edge lengths of the polyline summed in order (one lane: the order is the result)
__global__ void rrt_path_cost_kernel(const double *__restrict__ path, int n, double *__restrict__ out) {
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    double cost = 0.0;
    for (int i = 0; i + 1 < n; ++i) {
        const double *a = path + 3 * (size_t)i, *b = a + 3;
        cost += norm3(b[0] - a[0], b[1] - a[1], b[2] - a[2]);
    }
    out[0] = cost;
}

// RRTStar._adapt_random_node_position (rrt.py:140-148) for E (sample, nearest node) pairs
__global__ void rrt_steer_kernel(const double *__restrict__ sample, const double *__restrict__ nearest, int E,
                                 double step, double *__restrict__ out) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= E) return;
    const double *s = sample + 3 * (size_t)e, *p = nearest + 3 * (size_t)e;
    double n0 = s[0], n1 = s[1], n2 = s[2];
    const double d = norm3(n0 - p[0], n1 - p[1], n2 - p[2]);
    if (d > step) {
        n0 = round2(p[0] + (n0 - p[0]) * step / d);
        n1 = round2(p[1] + (n1 - p[1]) * step / d);
        n2 = round2(p[2] + (n2 - p[2]) * step / d);
    }
    out[3 * (size_t)e] = n0; out[3 * (size_t)e + 1] = n1; out[3 * (size_t)e + 2] = n2;
}

constexpr size_t kLdsBytesPerNode = 3 * 8 + 8 + 8 + 3 * 4;     // nodes, elen, cc, canon/up/nbr
constexpr size_t kLdsLimit = 160 * 1024;

size_t scratch_doubles_per_problem(int cap) { return 3 * (size_t)cap; }

struct DevBuf {
    void *p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 8); }
    template <class T> T *as() { return static_cast<T *>(p); }
};

bool finite_all(const double *p, size_t n) {
    for (size_t i = 0; i < n; ++i)
        if (!std::isfinite(p[i])) return false;
    return true;
}

}  // namespace

extern "C" {

int uavac_rrt_star_dev(uavac_ctx *ctx, const double *start, const double *goal, int B, double step, int max_iter,
                       const double *samples, const double *cuboids, int n_obs, double *nodes, int32_t *canon,
                       int32_t *parent, int32_t *best_parent, double *best_path, int32_t *counts,
                       double *best_cost) {
    UAVAC_ENTER(ctx);
    if (!start || !goal || !samples || !nodes || !canon || !parent || !best_parent || !best_path || !counts || !best_cost)
        return uavac_fail(ctx, UAVAC_EINVAL, "null pointer");
    if (B < 1 || max_iter < 1) return uavac_fail(ctx, UAVAC_EINVAL, "B and max_iter must be >= 1");
    if (!std::isfinite(step) || !(step > 0.0)) return uavac_fail(ctx, UAVAC_EINVAL, "max_distance must be finite and > 0");
    if (n_obs < 0 || (n_obs > 0 && !cuboids)) return uavac_fail(ctx, UAVAC_EINVAL, "bad obstacle list");
    const int cap = max_iter + 1;
    // first-pass capacity: 384 nodes = 20 KB of LDS = 8 problems per CU (lab obstacle set, 1 000 iterations allowed:
    // mean tree 210 nodes; measured at B = 16 384: one pass 49.6 ms, 256 / 384 / 512 nodes first 39.6 / 35.9 / 38.1 ms)
    constexpr int kFirstPassNodes = 384;
    const int small = kFirstPassNodes < cap ? kFirstPassNodes : cap;
    const size_t lds_full = kLdsBytesPerNode * (size_t)cap + 16, lds_small = kLdsBytesPerNode * (size_t)small + 16;
    const bool full_in_lds = lds_full <= kLdsLimit;
    if (!full_in_lds) {
        const size_t need = scratch_doubles_per_problem(cap) * (size_t)B;
        if (need > ctx->ws_cap) {
            if (ctx->d_ws) UAVAC_HIP(ctx, hipFree(ctx->d_ws));
            ctx->d_ws = nullptr;
            ctx->ws_cap = 0;
            UAVAC_HIP(ctx, hipMalloc(&ctx->d_ws, need * sizeof(double)));
            ctx->ws_cap = need;
        }
    }
    const size_t lds_max = full_in_lds ? lds_full : (lds_small <= kLdsLimit ? lds_small : 0);
    if (lds_max > 64 * 1024)
        UAVAC_HIP(ctx, hipFuncSetAttribute((const void *)rrt_star_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)lds_max));
#define UAVAC_RRT_ARGS start, goal, B, step, max_iter, samples, cuboids, n_obs, nodes, canon, parent, best_parent, best_path, counts, \
                       best_cost, ctx->d_ws
    if (small < cap && lds_small <= kLdsLimit) {
        // pass 0: small trees at high occupancy; pass 1: the overflowed problems with everything they may need
        hipLaunchKernelGGL(rrt_star_kernel<true>, dim3(B), dim3(W), lds_small, ctx->stream, UAVAC_RRT_ARGS, small, 0);
        if (full_in_lds)
            hipLaunchKernelGGL(rrt_star_kernel<true>, dim3(B), dim3(W), lds_full, ctx->stream, UAVAC_RRT_ARGS, cap, 1);
        else
            hipLaunchKernelGGL(rrt_star_kernel<false>, dim3(B), dim3(W), 0, ctx->stream, UAVAC_RRT_ARGS, cap, 1);
    } else if (full_in_lds) {
        hipLaunchKernelGGL(rrt_star_kernel<true>, dim3(B), dim3(W), lds_full, ctx->stream, UAVAC_RRT_ARGS, cap, 0);
    } else {
        hipLaunchKernelGGL(rrt_star_kernel<false>, dim3(B), dim3(W), 0, ctx->stream, UAVAC_RRT_ARGS, cap, 0);
    }
#undef UAVAC_RRT_ARGS
    UAVAC_HIP(ctx, hipGetLastError());
    return UAVAC_OK;
}

int uavac_rrt_segment_hits_dev(uavac_ctx *ctx, const double *p0, const double *p1, int E, const double *cuboids,
                               int n_obs, int32_t *hit) {
    UAVAC_ENTER(ctx);
    if (E < 0 || n_obs < 0) return uavac_fail(ctx, UAVAC_EINVAL, "negative count");
    if (E == 0) return UAVAC_OK;
    if (!p0 || !p1 || !hit || (n_obs > 0 && !cuboids)) return uavac_fail(ctx, UAVAC_EINVAL, "null pointer");
    hipLaunchKernelGGL(rrt_segment_hits_kernel, dim3((E + 255) / 256), dim3(256), 0, ctx->stream, p0, p1, E, cuboids,
                       n_obs, hit);
    UAVAC_HIP(ctx, hipGetLastError());
    return UAVAC_OK;
}

int uavac_rrt_edge_lengths_dev(uavac_ctx *ctx, const double *p0, const double *p1, int p1_is_single, int E,
                               double *out) {
    UAVAC_ENTER(ctx);
    if (E < 0) return uavac_fail(ctx, UAVAC_EINVAL, "negative count");
    if (E == 0) return UAVAC_OK;
    if (!p0 || !p1 || !out) return uavac_fail(ctx, UAVAC_EINVAL, "null pointer");
    hipLaunchKernelGGL(rrt_edge_lengths_kernel, dim3((E + 255) / 256), dim3(256), 0, ctx->stream, p0, p1,
                       p1_is_single ? 0 : 3, E, out);
    UAVAC_HIP(ctx, hipGetLastError());
    return UAVAC_OK;
}

int uavac_rrt_draw_nodes_dev(uavac_ctx *ctx, const uint32_t *seeds, const double *goals, int B, int n,
                             const double *limits_lw_host, const double *limits_up_host, double epsilon, double *samples,
                             int64_t *consumed) {
    UAVAC_ENTER(ctx);
    if (B < 1 || n < 1) return uavac_fail(ctx, UAVAC_EINVAL, "B and n must be >= 1");
    if (!seeds || !goals || !limits_lw_host || !limits_up_host || !samples) return uavac_fail(ctx, UAVAC_EINVAL, "null pointer");
    for (int a = 0; a < 3; ++a)
        if (!std::isfinite(limits_lw_host[a]) || !std::isfinite(limits_up_host[a]))
            return uavac_fail(ctx, UAVAC_ENONFINITE, "non-finite space limit");
    hipLaunchKernelGGL(rrt_draw_nodes_kernel, dim3(B), dim3(W), 0, ctx->stream, seeds, goals, B, n, limits_lw_host[0],
                       limits_lw_host[1], limits_lw_host[2], limits_up_host[0], limits_up_host[1], limits_up_host[2], epsilon,
                       samples, consumed);
    UAVAC_HIP(ctx, hipGetLastError());
    return UAVAC_OK;
}

int uavac_rrt_simplify_dev(uavac_ctx *ctx, const double *paths, const int32_t *lens, int B, int cap, const double *cuboids,
                           int n_obs, double *out_paths, int32_t *out_lens) {
    UAVAC_ENTER(ctx);
    if (B < 1 || cap < 1 || n_obs < 0) return uavac_fail(ctx, UAVAC_EINVAL, "B and cap must be >= 1");
    if (!paths || !lens || !out_paths || !out_lens || (n_obs > 0 && !cuboids)) return uavac_fail(ctx, UAVAC_EINVAL, "null pointer");
    hipLaunchKernelGGL(rrt_simplify_kernel, dim3(B), dim3(W), 0, ctx->stream, paths, lens, cap, cuboids, n_obs, out_paths,
                       out_lens);
    UAVAC_HIP(ctx, hipGetLastError());
    return UAVAC_OK;
}

int uavac_rrt_path_cost_dev(uavac_ctx *ctx, const double *path, int n, double *cost) {
    UAVAC_ENTER(ctx);
    if (n < 0) return uavac_fail(ctx, UAVAC_EINVAL, "negative count");
    if (!cost || (n > 0 && !path)) return uavac_fail(ctx, UAVAC_EINVAL, "null pointer");
    hipLaunchKernelGGL(rrt_path_cost_kernel, dim3(1), dim3(64), 0, ctx->stream, path, n, cost);
    UAVAC_HIP(ctx, hipGetLastError());
    return UAVAC_OK;
}

int uavac_rrt_steer_dev(uavac_ctx *ctx, const double *sample, const double *nearest, int E, double step, double *out) {
    UAVAC_ENTER(ctx);
    if (E < 0) return uavac_fail(ctx, UAVAC_EINVAL, "negative count");
    if (E == 0) return UAVAC_OK;
    if (!sample || !nearest || !out) return uavac_fail(ctx, UAVAC_EINVAL, "null pointer");
    hipLaunchKernelGGL(rrt_steer_kernel, dim3((E + 255) / 256), dim3(256), 0, ctx->stream, sample, nearest, E, step, out);
    UAVAC_HIP(ctx, hipGetLastError());
    return UAVAC_OK;
}

// ------------------------------------------------------------------------------------------ host twins
int uavac_rrt_star(uavac_ctx *ctx, const double *start, const double *goal, int B, double step, int max_iter,
                   const double *samples, const double *cuboids, int n_obs, double *nodes, int32_t *canon,
                   int32_t *parent, int32_t *best_parent, double *best_path, int32_t *counts, double *best_cost) {
    UAVAC_ENTER(ctx);
    if (!start || !goal || !samples || !nodes || !canon || !parent || !best_parent || !best_path || !counts || !best_cost)
        return uavac_fail(ctx, UAVAC_EINVAL, "null pointer");
    if (B < 1 || max_iter < 1) return uavac_fail(ctx, UAVAC_EINVAL, "B and max_iter must be >= 1");
    if (n_obs < 0 || (n_obs > 0 && !cuboids)) return uavac_fail(ctx, UAVAC_EINVAL, "bad obstacle list");
    const size_t cap = (size_t)max_iter + 1, zB = (size_t)B;
    if (!finite_all(start, 3 * zB) || !finite_all(goal, 3 * zB) || !finite_all(samples, 3 * zB * max_iter) ||
        (n_obs > 0 && !finite_all(cuboids, 6 * (size_t)n_obs)))
        return uavac_fail(ctx, UAVAC_ENONFINITE, "non-finite start, goal, sample or cuboid");
    DevBuf ds, dg, dsm, dc, dn, dca, dp, dbp, dpath, dcnt, dcost;
    UAVAC_HIP(ctx, ds.alloc(24 * zB));
    UAVAC_HIP(ctx, dg.alloc(24 * zB));
    UAVAC_HIP(ctx, dsm.alloc(24 * zB * max_iter));
    UAVAC_HIP(ctx, dc.alloc(48 * (size_t)n_obs));
    UAVAC_HIP(ctx, dn.alloc(24 * zB * cap));
    UAVAC_HIP(ctx, dca.alloc(4 * zB * cap));
    UAVAC_HIP(ctx, dp.alloc(4 * zB * cap));
    UAVAC_HIP(ctx, dbp.alloc(4 * zB * cap));
    UAVAC_HIP(ctx, dpath.alloc(24 * zB * cap));
    UAVAC_HIP(ctx, dcnt.alloc(24 * zB));
    UAVAC_HIP(ctx, dcost.alloc(8 * zB));
    UAVAC_HIP(ctx, hipMemcpyAsync(ds.p, start, 24 * zB, hipMemcpyHostToDevice, ctx->stream));
    UAVAC_HIP(ctx, hipMemcpyAsync(dg.p, goal, 24 * zB, hipMemcpyHostToDevice, ctx->stream));
    UAVAC_HIP(ctx, hipMemcpyAsync(dsm.p, samples, 24 * zB * max_iter, hipMemcpyHostToDevice, ctx->stream));
    if (n_obs > 0) UAVAC_HIP(ctx, hipMemcpyAsync(dc.p, cuboids, 48 * (size_t)n_obs, hipMemcpyHostToDevice, ctx->stream));
    if (int rc = uavac_rrt_star_dev(ctx, ds.as<double>(), dg.as<double>(), B, step, max_iter, dsm.as<double>(),
                                    n_obs > 0 ? dc.as<double>() : nullptr, n_obs, dn.as<double>(), dca.as<int32_t>(),
                                    dp.as<int32_t>(), dbp.as<int32_t>(), dpath.as<double>(), dcnt.as<int32_t>(),
                                    dcost.as<double>())) return rc;
    UAVAC_HIP(ctx, hipMemcpyAsync(nodes, dn.p, 24 * zB * cap, hipMemcpyDeviceToHost, ctx->stream));
    UAVAC_HIP(ctx, hipMemcpyAsync(canon, dca.p, 4 * zB * cap, hipMemcpyDeviceToHost, ctx->stream));
    UAVAC_HIP(ctx, hipMemcpyAsync(parent, dp.p, 4 * zB * cap, hipMemcpyDeviceToHost, ctx->stream));
    UAVAC_HIP(ctx, hipMemcpyAsync(best_parent, dbp.p, 4 * zB * cap, hipMemcpyDeviceToHost, ctx->stream));
    UAVAC_HIP(ctx, hipMemcpyAsync(best_path, dpath.p, 24 * zB * cap, hipMemcpyDeviceToHost, ctx->stream));
    UAVAC_HIP(ctx, hipMemcpyAsync(counts, dcnt.p, 24 * zB, hipMemcpyDeviceToHost, ctx->stream));
    UAVAC_HIP(ctx, hipMemcpyAsync(best_cost, dcost.p, 8 * zB, hipMemcpyDeviceToHost, ctx->stream));
    UAVAC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return UAVAC_OK;
}

int uavac_rrt_segment_hits(uavac_ctx *ctx, const double *p0, const double *p1, int E, const double *cuboids, int n_obs,
                           int32_t *hit) {
    UAVAC_ENTER(ctx);
    if (E < 0 || n_obs < 0) return uavac_fail(ctx, UAVAC_EINVAL, "negative count");
    if (E == 0) return UAVAC_OK;
    if (!p0 || !p1 || !hit || (n_obs > 0 && !cuboids)) return uavac_fail(ctx, UAVAC_EINVAL, "null pointer");
    const size_t zE = (size_t)E;
    DevBuf d0, d1, dc, dh;
    UAVAC_HIP(ctx, d0.alloc(24 * zE));
    UAVAC_HIP(ctx, d1.alloc(24 * zE));
    UAVAC_HIP(ctx, dc.alloc(48 * (size_t)n_obs));
    UAVAC_HIP(ctx, dh.alloc(4 * zE));
    UAVAC_HIP(ctx, hipMemcpyAsync(d0.p, p0, 24 * zE, hipMemcpyHostToDevice, ctx->stream));
    UAVAC_HIP(ctx, hipMemcpyAsync(d1.p, p1, 24 * zE, hipMemcpyHostToDevice, ctx->stream));
    if (n_obs > 0) UAVAC_HIP(ctx, hipMemcpyAsync(dc.p, cuboids, 48 * (size_t)n_obs, hipMemcpyHostToDevice, ctx->stream));
    if (int rc = uavac_rrt_segment_hits_dev(ctx, d0.as<double>(), d1.as<double>(), E, dc.as<double>(), n_obs,
                                            dh.as<int32_t>())) return rc;
    UAVAC_HIP(ctx, hipMemcpyAsync(hit, dh.p, 4 * zE, hipMemcpyDeviceToHost, ctx->stream));
    UAVAC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return UAVAC_OK;
}

int uavac_rrt_edge_lengths(uavac_ctx *ctx, const double *p0, const double *p1, int p1_is_single, int E, double *out) {
    UAVAC_ENTER(ctx);
    if (E < 0) return uavac_fail(ctx, UAVAC_EINVAL, "negative count");
    if (E == 0) return UAVAC_OK;
    if (!p0 || !p1 || !out) return uavac_fail(ctx, UAVAC_EINVAL, "null pointer");
    const size_t zE = (size_t)E, n1 = p1_is_single ? 1 : zE;
    DevBuf d0, d1, dout;
    UAVAC_HIP(ctx, d0.alloc(24 * zE));
    UAVAC_HIP(ctx, d1.alloc(24 * n1));
    UAVAC_HIP(ctx, dout.alloc(8 * zE));
    UAVAC_HIP(ctx, hipMemcpyAsync(d0.p, p0, 24 * zE, hipMemcpyHostToDevice, ctx->stream));
    UAVAC_HIP(ctx, hipMemcpyAsync(d1.p, p1, 24 * n1, hipMemcpyHostToDevice, ctx->stream));
    if (int rc = uavac_rrt_edge_lengths_dev(ctx, d0.as<double>(), d1.as<double>(), p1_is_single, E, dout.as<double>()))
        return rc;
    UAVAC_HIP(ctx, hipMemcpyAsync(out, dout.p, 8 * zE, hipMemcpyDeviceToHost, ctx->stream));
    UAVAC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return UAVAC_OK;
}

int uavac_rrt_simplify(uavac_ctx *ctx, const double *paths, const int32_t *lens, int B, int cap, const double *cuboids,
                       int n_obs, double *out_paths, int32_t *out_lens) {
    UAVAC_ENTER(ctx);
    if (B < 1 || cap < 1 || n_obs < 0) return uavac_fail(ctx, UAVAC_EINVAL, "B and cap must be >= 1");
    if (!paths || !lens || !out_paths || !out_lens || (n_obs > 0 && !cuboids)) return uavac_fail(ctx, UAVAC_EINVAL, "null pointer");
    for (int b = 0; b < B; ++b)
        if (lens[b] < 0 || lens[b] > cap) return uavac_fail(ctx, UAVAC_EINVAL, "path length outside [0, cap]");
    const size_t np = (size_t)B * cap * 3;
    DevBuf dp, dl, dc, dout, dol;
    UAVAC_HIP(ctx, dp.alloc(8 * np));
    UAVAC_HIP(ctx, dl.alloc(4 * (size_t)B));
    UAVAC_HIP(ctx, dc.alloc(48 * (size_t)n_obs));
    UAVAC_HIP(ctx, dout.alloc(8 * np));
    UAVAC_HIP(ctx, dol.alloc(4 * (size_t)B));
    UAVAC_HIP(ctx, hipMemcpyAsync(dp.p, paths, 8 * np, hipMemcpyHostToDevice, ctx->stream));
    UAVAC_HIP(ctx, hipMemcpyAsync(dl.p, lens, 4 * (size_t)B, hipMemcpyHostToDevice, ctx->stream));
    if (n_obs > 0) UAVAC_HIP(ctx, hipMemcpyAsync(dc.p, cuboids, 48 * (size_t)n_obs, hipMemcpyHostToDevice, ctx->stream));
    if (int rc = uavac_rrt_simplify_dev(ctx, dp.as<double>(), dl.as<int32_t>(), B, cap, n_obs > 0 ? dc.as<double>() : nullptr,
                                        n_obs, dout.as<double>(), dol.as<int32_t>())) return rc;
    UAVAC_HIP(ctx, hipMemcpyAsync(out_paths, dout.p, 8 * np, hipMemcpyDeviceToHost, ctx->stream));
    UAVAC_HIP(ctx, hipMemcpyAsync(out_lens, dol.p, 4 * (size_t)B, hipMemcpyDeviceToHost, ctx->stream));
    UAVAC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return UAVAC_OK;
}

int uavac_rrt_path_cost(uavac_ctx *ctx, const double *path, int n, double *cost) {
    UAVAC_ENTER(ctx);
    if (n < 0) return uavac_fail(ctx, UAVAC_EINVAL, "negative count");
    if (!cost || (n > 0 && !path)) return uavac_fail(ctx, UAVAC_EINVAL, "null pointer");
    DevBuf dp, dc;
    UAVAC_HIP(ctx, dp.alloc(24 * (size_t)n));
    UAVAC_HIP(ctx, dc.alloc(8));
    if (n > 0) UAVAC_HIP(ctx, hipMemcpyAsync(dp.p, path, 24 * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
    if (int rc = uavac_rrt_path_cost_dev(ctx, dp.as<double>(), n, dc.as<double>())) return rc;
    UAVAC_HIP(ctx, hipMemcpyAsync(cost, dc.p, 8, hipMemcpyDeviceToHost, ctx->stream));
    UAVAC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return UAVAC_OK;
}

int uavac_rrt_steer(uavac_ctx *ctx, const double *sample, const double *nearest, int E, double step, double *out) {
    UAVAC_ENTER(ctx);
    if (E < 0) return uavac_fail(ctx, UAVAC_EINVAL, "negative count");
    if (E == 0) return UAVAC_OK;
    if (!sample || !nearest || !out) return uavac_fail(ctx, UAVAC_EINVAL, "null pointer");
    const size_t zE = (size_t)E;
    DevBuf d0, d1, dout;
    UAVAC_HIP(ctx, d0.alloc(24 * zE));
    UAVAC_HIP(ctx, d1.alloc(24 * zE));
    UAVAC_HIP(ctx, dout.alloc(24 * zE));
    UAVAC_HIP(ctx, hipMemcpyAsync(d0.p, sample, 24 * zE, hipMemcpyHostToDevice, ctx->stream));
    UAVAC_HIP(ctx, hipMemcpyAsync(d1.p, nearest, 24 * zE, hipMemcpyHostToDevice, ctx->stream));
    if (int rc = uavac_rrt_steer_dev(ctx, d0.as<double>(), d1.as<double>(), E, step, dout.as<double>())) return rc;
    UAVAC_HIP(ctx, hipMemcpyAsync(out, dout.p, 24 * zE, hipMemcpyDeviceToHost, ctx->stream));
    UAVAC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return UAVAC_OK;
}

}  // extern "C"
