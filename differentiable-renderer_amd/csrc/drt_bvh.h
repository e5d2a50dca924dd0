// drt_bvh.h -- host-side BVH construction for the triangle-mesh extension (all meshes of a scene
// share ONE tree).  Binned-SAH top-down binary build with a bounded depth, COLLAPSED to a 4-wide
// tree: every emitted node carries the boxes of up to four children (one fetch on the device
// yields four box tests and their near-to-far order, and the tree is half as deep -- the walk is
// bound by dependent node fetches); leaves are not nodes, a child link is either an interior
// node index or a triangle range.  The first `top` nodes are the top of the tree in breadth-first order (K2 stages
// them in LDS), the rest follow depth-first (subtrees contiguous).
#pragma once

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <queue>
#include <vector>

namespace drt_bvh {

struct Tri {
    double v0[3], e1[3], e2[3];
    double n[3];          // normalize(cross(e1, e2)), computed like the oracle does
    uint32_t global;      // index among all triangles of the scene (meshes in scene order)
    uint32_t flat;        // position in the flattened scene (tie order, pathtracer.hpp:80)
    uint32_t ids;         // colour parameter (0xFFFF = no BxDF) | material << 16 (0xFF = none) | emitter << 24 (0xFF = none)
};

// child link: bit 31 clear = interior node index; bit 31 set = leaf, (first << 3) | count in the
// low bits (count 0 = empty child, its box is inverted so it is never entered)
constexpr int kWidth = 4;
struct Node {
    double lo[kWidth][3], hi[kWidth][3];
    uint32_t child[kWidth];
};

constexpr uint32_t kLeaf = 0x80000000u;
constexpr int kMaxLeaf = 4;
constexpr int kMaxDepth = 20;     // bound of the binary build's depth.  The collapse below splits the child with the
                                  // LARGEST box, so a child that is never re-split advances one binary level per wide
                                  // level: the wide depth is NOT kMaxDepth / 2 in general.  What the device needs is
                                  // Built::stack_need <= its per-lane stack (DRT_BVH_STACK); build() measures it on the
                                  // finished tree and rebuilds with a smaller depth bound until it holds.
#ifndef DRT_BVH_STACK
#define DRT_BVH_STACK 30
#endif
constexpr int kStackEntries = DRT_BVH_STACK; // = DRT_BVH_STACK (drt_device.h; static_assert in drt_hip.hip)

struct Built {
    std::vector<Node> nodes;      // final order, root = 0
    std::vector<uint32_t> order;  // triangle indices (into the input) in leaf order
    uint32_t top = 0;             // nodes [0, top) are the breadth-first top of the tree
    int depth = 0;                // binary depth of the SAH tree
    int wide_depth = 0;           // levels of 4-wide nodes
    int stack_need = 0;           // worst number of entries the ordered walk can hold at once: along a root-to-leaf
                                  // path every node pushes at most (children - 1) links before descending
    int sah_splits = 0, median_splits = 0;   // how the interior nodes were split (diagnostics / tests)
};

namespace detail {

struct Tmp {
    double lo[3], hi[3];
    int left = -1, right = -1;    // children in the temporary tree
    uint32_t first = 0, count = 0;
};

inline void grow(double lo[3], double hi[3], const double p[3])
{
    for (int a = 0; a < 3; ++a) {
        lo[a] = std::min(lo[a], p[a]);
        hi[a] = std::max(hi[a], p[a]);
    }
}

inline double area(const double lo[3], const double hi[3])
{
    const double dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
    return (dx < 0 || dy < 0 || dz < 0) ? 0.0 : 2.0 * (dx * dy + dy * dz + dz * dx);
}

struct Builder {
    const std::vector<Tri>& tris;
    std::vector<double> cen;          // centroids, 3 per triangle
    std::vector<double> blo, bhi;     // bounds, 3 per triangle
    std::vector<uint32_t> idx;
    std::vector<Tmp> tmp;

    explicit Builder(const std::vector<Tri>& t) : tris(t)
    {
        const size_t n = t.size();
        cen.resize(n * 3); blo.resize(n * 3); bhi.resize(n * 3); idx.resize(n);
        for (size_t i = 0; i < n; ++i) {
            idx[i] = (uint32_t)i;
            for (int a = 0; a < 3; ++a) {
                const double p0 = t[i].v0[a], p1 = p0 + t[i].e1[a], p2 = p0 + t[i].e2[a];
                blo[i * 3 + a] = std::min(p0, std::min(p1, p2));
                bhi[i * 3 + a] = std::max(p0, std::max(p1, p2));
                cen[i * 3 + a] = (p0 + p1 + p2) / 3.0;
            }
        }
    }

    int max_depth_seen = 0;
    int depth_bound = kMaxDepth;
    int sah_splits = 0, median_splits = 0;

    int build(uint32_t first, uint32_t count, int depth = 0)
    {
        max_depth_seen = std::max(max_depth_seen, depth);
        const int me = (int)tmp.size();
        tmp.emplace_back();
        for (int a = 0; a < 3; ++a) { tmp[me].lo[a] = INFINITY; tmp[me].hi[a] = -INFINITY; }
        double clo[3] = {INFINITY, INFINITY, INFINITY}, chi[3] = {-INFINITY, -INFINITY, -INFINITY};
        for (uint32_t i = first; i < first + count; ++i) {
            grow(tmp[me].lo, tmp[me].hi, &blo[idx[i] * 3]);
            grow(tmp[me].lo, tmp[me].hi, &bhi[idx[i] * 3]);
            grow(clo, chi, &cen[idx[i] * 3]);
        }
        tmp[me].first = first;
        tmp[me].count = count;
        if (count <= 2)
            return me;
        // depth bound: once the remaining levels are only enough for a balanced tree, split by median
        int need = 0;
        while ((1u << need) * (uint32_t)kMaxLeaf < count) ++need;
        const bool force_median = depth + need + 1 >= depth_bound;
        // binned SAH over the widest centroid axis first, then the others
        constexpr int B = 16;
        double best = INFINITY;
        int best_axis = -1, best_bin = -1;
        for (int a = 0; a < 3 && !force_median; ++a) {
            const double ext = chi[a] - clo[a];
            if (!(ext > 0))
                continue;
            double lo[B][3], hi[B][3];
            uint32_t cnt[B] = {0};
            for (int b = 0; b < B; ++b)
                for (int c = 0; c < 3; ++c) { lo[b][c] = INFINITY; hi[b][c] = -INFINITY; }
            for (uint32_t i = first; i < first + count; ++i) {
                int b = (int)((cen[idx[i] * 3 + a] - clo[a]) / ext * B);
                b = b < 0 ? 0 : (b >= B ? B - 1 : b);
                cnt[b]++;
                grow(lo[b], hi[b], &blo[idx[i] * 3]);
                grow(lo[b], hi[b], &bhi[idx[i] * 3]);
            }
            double rlo[B][3], rhi[B][3];
            uint32_t rc[B];
            double alo[3] = {INFINITY, INFINITY, INFINITY}, ahi[3] = {-INFINITY, -INFINITY, -INFINITY};
            uint32_t acc = 0;
            // (an empty bin holds the inverted box (+inf, -inf): growing by its corners would blow the running box
            // up to (-inf, +inf), make every cost infinite and silently turn the whole build into median splits)
            for (int b = B - 1; b >= 0; --b) {
                if (cnt[b]) { grow(alo, ahi, lo[b]); grow(alo, ahi, hi[b]); }
                acc += cnt[b];
                std::memcpy(rlo[b], alo, sizeof alo); std::memcpy(rhi[b], ahi, sizeof ahi);
                rc[b] = acc;
            }
            double llo[3] = {INFINITY, INFINITY, INFINITY}, lhi[3] = {-INFINITY, -INFINITY, -INFINITY};
            uint32_t lc = 0;
            for (int b = 0; b < B - 1; ++b) {
                if (cnt[b]) { grow(llo, lhi, lo[b]); grow(llo, lhi, hi[b]); }
                lc += cnt[b];
                if (lc == 0 || rc[b + 1] == 0)
                    continue;
                const double cost = lc * area(llo, lhi) + rc[b + 1] * area(rlo[b + 1], rhi[b + 1]);
                if (cost < best) { best = cost; best_axis = a; best_bin = b; }
            }
        }
        const double leaf_cost = count * area(tmp[me].lo, tmp[me].hi);
        if (count <= (uint32_t)kMaxLeaf && (best_axis < 0 || best >= leaf_cost))
            return me;
        uint32_t mid;
        (best_axis >= 0 ? sah_splits : median_splits)++;
        if (best_axis >= 0) {
            const double ext = chi[best_axis] - clo[best_axis];
            auto it = std::partition(idx.begin() + first, idx.begin() + first + count, [&](uint32_t t) {
                int b = (int)((cen[t * 3 + best_axis] - clo[best_axis]) / ext * B);
                b = b < 0 ? 0 : (b >= B ? B - 1 : b);
                return b <= best_bin;
            });
            mid = (uint32_t)(it - idx.begin());
        } else {
            // median split along the widest centroid axis (balanced: bounds the depth)
            int ax = 0;
            for (int a2 = 1; a2 < 3; ++a2)
                if (chi[a2] - clo[a2] > chi[ax] - clo[ax]) ax = a2;
            mid = first + count / 2;
            std::nth_element(idx.begin() + first, idx.begin() + mid, idx.begin() + first + count,
                             [&](uint32_t x, uint32_t y) { return cen[x * 3 + ax] < cen[y * 3 + ax]; });
        }
        if (mid == first || mid == first + count)
            mid = first + count / 2;
        const int l = build(first, mid - first, depth + 1);
        const int r = build(mid, first + count - mid, depth + 1);
        tmp[me].left = l;
        tmp[me].right = r;
        tmp[me].count = 0;
        return me;
    }
};

} // namespace detail

// pad: boxes are grown by this much on every side (absorbs f32 rounding of boxes, rays and the
// traversal arithmetic; the closest hit is still exact, the box test only has to be conservative)
inline Built build_bounded(const std::vector<Tri>& tris, uint32_t max_top, double pad, int depth_bound)
{
    Built out;
    if (tris.empty())
        return out;
    detail::Builder b(tris);
    b.depth_bound = depth_bound;
    b.tmp.reserve(tris.size() * 2);
    b.build(0, (uint32_t)tris.size());
    out.depth = b.max_depth_seen;
    out.sah_splits = b.sah_splits;
    out.median_splits = b.median_splits;
    const std::vector<detail::Tmp>& t = b.tmp;
    out.order = b.idx;
    auto leaf_link = [&](int u) { return kLeaf | (t[u].first << 3) | t[u].count; };
    auto is_leaf = [&](int u) { return t[u].left < 0; };
    // children of wide node u: start from its two binary children, keep splitting the interior
    // child with the largest box until there are kWidth of them (or only leaves are left)
    auto wide_children = [&](int u, int out[kWidth]) {
        int n = 0;
        out[n++] = t[u].left;
        out[n++] = t[u].right;
        while (n < kWidth) {
            int best = -1;
            double best_area = -1;
            for (int i = 0; i < n; ++i)
                if (!is_leaf(out[i]) && detail::area(t[out[i]].lo, t[out[i]].hi) > best_area) {
                    best_area = detail::area(t[out[i]].lo, t[out[i]].hi);
                    best = i;
                }
            if (best < 0)
                break;
            const int c = out[best];
            out[best] = t[c].left;
            out[n++] = t[c].right;
        }
        return n;
    };
    auto set_child = [&](Node& nd, int side, int u, uint32_t link) {
        for (int a = 0; a < 3; ++a) { nd.lo[side][a] = t[u].lo[a] - pad; nd.hi[side][a] = t[u].hi[a] + pad; }
        nd.child[side] = link;
    };
    auto set_empty = [&](Node& nd, int side) {
        for (int a = 0; a < 3; ++a) { nd.lo[side][a] = INFINITY; nd.hi[side][a] = -INFINITY; }
        nd.child[side] = kLeaf;     // leaf with count 0; the inverted box is never entered
    };
    if (is_leaf(0)) {             // the whole scene is one leaf: a root with one real child
        Node nd;
        set_child(nd, 0, 0, leaf_link(0));
        for (int sde = 1; sde < kWidth; ++sde)
            set_empty(nd, sde);
        out.nodes.push_back(nd);
        out.top = 1;
        out.wide_depth = 1;
        out.stack_need = 0;
        return out;
    }
    // wide nodes = the binary nodes that survive the collapse; physical order: breadth-first prefix
    // of at most max_top, then depth-first
    const uint32_t n = (uint32_t)t.size();
    std::vector<uint32_t> pos(n, 0xFFFFFFFFu), at;
    std::vector<int> frontier;
    {
        std::queue<int> q;
        q.push(0);
        while (!q.empty() && at.size() < max_top) {
            const int u = q.front();
            q.pop();
            pos[u] = (uint32_t)at.size();
            at.push_back((uint32_t)u);
            int ch[kWidth];
            const int nc = wide_children(u, ch);
            for (int i = 0; i < nc; ++i)
                if (!is_leaf(ch[i])) q.push(ch[i]);
        }
        while (!q.empty()) { frontier.push_back(q.front()); q.pop(); }
    }
    out.top = (uint32_t)at.size();
    std::vector<int> stack(frontier.rbegin(), frontier.rend());
    while (!stack.empty()) {
        const int u = stack.back();
        stack.pop_back();
        pos[u] = (uint32_t)at.size();
        at.push_back((uint32_t)u);
        int ch[kWidth];
        const int nc = wide_children(u, ch);
        for (int i = nc - 1; i >= 0; --i)
            if (!is_leaf(ch[i])) stack.push_back(ch[i]);
    }
    out.nodes.resize(at.size());
    for (size_t i = 0; i < at.size(); ++i) {
        int ch[kWidth];
        const int nc = wide_children((int)at[i], ch);
        for (int c = 0; c < kWidth; ++c) {
            if (c < nc)
                set_child(out.nodes[i], c, ch[c], is_leaf(ch[c]) ? leaf_link(ch[c]) : pos[ch[c]]);
            else
                set_empty(out.nodes[i], c);
        }
    }
    // Worst stack occupancy of the device's ordered walk and the depth in wide levels, on the finished tree.
    // Children come after their parent in `at` (breadth-first prefix, then depth-first), so one backward
    // sweep sees every child before its parent.
    {
        std::vector<int> need(out.nodes.size(), 0), levels(out.nodes.size(), 1);
        for (size_t i = out.nodes.size(); i-- > 0;) {
            int nc = 0, deepest = 0, lv = 0;
            for (int c = 0; c < kWidth; ++c) {
                const uint32_t link = out.nodes[i].child[c];
                if (link == kLeaf)
                    continue;
                ++nc;
                if (!(link & kLeaf)) {
                    deepest = std::max(deepest, need[link]);
                    lv = std::max(lv, levels[link]);
                }
            }
            need[i] = (nc > 0 ? nc - 1 : 0) + deepest;
            levels[i] = 1 + lv;
        }
        out.stack_need = need[0];
        out.wide_depth = levels[0];
    }
    return out;
}

// The tree the device walks: binned SAH under the default depth bound; if the collapsed tree could hold more
// than `stack_entries` links on the walk's stack (deep, unbalanced trees: nested scales, long slivers), rebuild
// with a tighter bound -- more of the tree becomes balanced median splits -- until it fits.  A bound of
// ceil(log2(n / kMaxLeaf)) + 1 is a fully balanced tree; stack_entries >= 3 * that / 2 always terminates the loop.
inline Built build(const std::vector<Tri>& tris, uint32_t max_top, double pad, int stack_entries = kStackEntries)
{
    int bound = kMaxDepth;
    for (;;) {
        Built b = build_bounded(tris, max_top, pad, bound);
        if (b.stack_need <= stack_entries || bound <= 2)
            return b;
        --bound;
    }
}

// ---- device encoding: one 4-wide node in 64 bytes ---------------------------------------------
// The walk is bound by the address rate of per-lane divergent 16-byte loads, so bytes per visit
// are what count: child boxes are stored as 8-bit offsets on a per-node power-of-two grid,
//   word 0..2  origin.xyz (f32, <= every child's lo)       word 3   grid exponent bytes ex | ey<<8 | ez<<16
//   word 4..7  links of children 0..3
//   word 8..10 q_lo.x, q_lo.y, q_lo.z (one byte per child) word 11..13  q_hi.x, q_hi.y, q_hi.z
//   word 14,15 unused
// child c, axis a:  lo = origin[a] + q_lo[a][c] * 2^(e[a]-127),  hi likewise.  The encoder checks
// that the decoded box -- evaluated exactly AND as the device's f32 fma evaluates it -- contains
// the child's true box, so the f32 and the f64 kernels both stay conservative.
struct QNode {
    uint32_t w[16];
};

inline float grid_scale(uint32_t ebyte)
{
    const uint32_t bits = ebyte << 23;
    float f;
    memcpy(&f, &bits, 4);
    return f;
}

inline QNode quantise(const Node& n)
{
    QNode q;
    memset(&q, 0, sizeof q);
    uint32_t qlo[3] = {0, 0, 0}, qhi[3] = {0, 0, 0}, ebytes = 0;
    for (int a = 0; a < 3; ++a) {
        double lo = INFINITY, hi = -INFINITY;
        for (int c = 0; c < kWidth; ++c)
            if (n.child[c] != kLeaf) {
                lo = std::min(lo, n.lo[c][a]);
                hi = std::max(hi, n.hi[c][a]);
            }
        float origin = (float)lo;
        if ((double)origin > lo)
            origin = nextafterf(origin, -INFINITY);
        int e;
        (void)frexp(std::max(hi - (double)origin, 1e-30) / 255.0, &e);     // 2^e > extent / 255
        uint32_t eb = (uint32_t)std::min(std::max(e + 127, 1), 254);
        for (;;) {                                         // grow the grid until every child fits in 8 bits
            const float sc = grid_scale(eb);
            bool ok = true;
            qlo[a] = qhi[a] = 0;
            for (int c = 0; c < kWidth && ok; ++c) {
                if (n.child[c] == kLeaf)
                    continue;
                auto decoded_le = [&](long v, double x) { return (double)fmaf((float)v, sc, origin) <= x && (double)origin + (double)v * (double)sc <= x; };
                auto decoded_ge = [&](long v, double x) { return (double)fmaf((float)v, sc, origin) >= x && (double)origin + (double)v * (double)sc >= x; };
                long l = (long)floor((n.lo[c][a] - (double)origin) / (double)sc);
                l = std::min(std::max(l, 0L), 255L);
                while (l > 0 && !decoded_le(l, n.lo[c][a])) --l;
                long h = (long)ceil((n.hi[c][a] - (double)origin) / (double)sc);
                h = std::max(h, 0L);
                while (h <= 255 && !decoded_ge(h, n.hi[c][a])) ++h;
                if (h > 255 || !decoded_le(l, n.lo[c][a])) { ok = false; break; }
                qlo[a] |= (uint32_t)l << (8 * c);
                qhi[a] |= (uint32_t)h << (8 * c);
            }
            if (ok || eb >= 254)
                break;
            ++eb;
        }
        memcpy(&q.w[a], &origin, 4);
        ebytes |= eb << (8 * a);
    }
    q.w[3] = ebytes;
    for (int c = 0; c < kWidth; ++c)
        q.w[4 + c] = n.child[c];
    for (int a = 0; a < 3; ++a) {
        q.w[8 + a] = qlo[a];
        q.w[11 + a] = qhi[a];
    }
    return q;
}

// decoded child box as the device sees it (tests and debugging)
inline void decode(const QNode& q, int c, double lo[3], double hi[3])
{
    for (int a = 0; a < 3; ++a) {
        float origin;
        memcpy(&origin, &q.w[a], 4);
        const double sc = grid_scale((q.w[3] >> (8 * a)) & 0xFFu);
        lo[a] = (double)origin + (double)((q.w[8 + a] >> (8 * c)) & 0xFFu) * sc;
        hi[a] = (double)origin + (double)((q.w[11 + a] >> (8 * c)) & 0xFFu) * sc;
    }
}

} // namespace drt_bvh
