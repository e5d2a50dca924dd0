// drt_bvh.h -- host-side BVH construction for the triangle-mesh extension (all meshes of a scene
// share ONE tree).  Binned-SAH top-down build, then a THREADED layout: every node carries a
// hit link (first child, or the leaf's triangle range) and a miss link (where to continue when
// the box is missed or the leaf is done), so the device traverses without a stack.  Links are
// explicit, so the physical order is free: the first `top` nodes are the top of the tree in
// breadth-first order (K2 stages them in LDS), the rest follow depth-first (subtrees contiguous).
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
    uint32_t ids;         // material | emitter << 16 (0xFFFF = none)
};

struct Node {
    double lo[3], hi[3];
    uint32_t hit;         // interior: index of the first child; leaf: 0x80000000 | first << 3 | count
    uint32_t miss;        // next node when this one is missed / finished; 0xFFFFFFFF = done
};

constexpr uint32_t kEnd = 0xFFFFFFFFu;
constexpr int kMaxLeaf = 4;

struct Built {
    std::vector<Node> nodes;      // threaded, final order
    std::vector<uint32_t> order;  // triangle indices (into the input) in leaf order
    uint32_t top = 0;             // nodes [0, top) are the breadth-first top of the tree
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

    int build(uint32_t first, uint32_t count)
    {
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
        // binned SAH over the widest centroid axis first, then the others
        constexpr int B = 16;
        double best = INFINITY;
        int best_axis = -1, best_bin = -1;
        for (int a = 0; a < 3; ++a) {
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
            for (int b = B - 1; b >= 0; --b) {
                grow(alo, ahi, lo[b]); grow(alo, ahi, hi[b]);
                acc += cnt[b];
                std::memcpy(rlo[b], alo, sizeof alo); std::memcpy(rhi[b], ahi, sizeof ahi);
                rc[b] = acc;
            }
            double llo[3] = {INFINITY, INFINITY, INFINITY}, lhi[3] = {-INFINITY, -INFINITY, -INFINITY};
            uint32_t lc = 0;
            for (int b = 0; b < B - 1; ++b) {
                grow(llo, lhi, lo[b]); grow(llo, lhi, hi[b]);
                lc += cnt[b];
                if (lc == 0 || rc[b + 1] == 0)
                    continue;
                const double cost = lc * area(llo, lhi) + rc[b + 1] * area(rlo[b + 1], rhi[b + 1]);
                if (cost < best) { best = cost; best_axis = a; best_bin = b; }
            }
        }
        const double leaf_cost = count * area(tmp[me].lo, tmp[me].hi);
        if (best_axis < 0 || (count <= (uint32_t)kMaxLeaf && best >= leaf_cost))
            if (count <= (uint32_t)kMaxLeaf)
                return me;
        uint32_t mid;
        if (best_axis >= 0) {
            const double ext = chi[best_axis] - clo[best_axis];
            auto it = std::partition(idx.begin() + first, idx.begin() + first + count, [&](uint32_t t) {
                int b = (int)((cen[t * 3 + best_axis] - clo[best_axis]) / ext * B);
                b = b < 0 ? 0 : (b >= B ? B - 1 : b);
                return b <= best_bin;
            });
            mid = (uint32_t)(it - idx.begin());
        } else {
            mid = first + count / 2;      // coincident centroids: split the list
        }
        if (mid == first || mid == first + count)
            mid = first + count / 2;
        const int l = build(first, mid - first);
        const int r = build(mid, first + count - mid);
        tmp[me].left = l;
        tmp[me].right = r;
        tmp[me].count = 0;
        return me;
    }
};

} // namespace detail

// pad: boxes are grown by this much on every side (absorbs f32 rounding of boxes, rays and the
// traversal arithmetic; the closest hit is still exact, the box test only has to be conservative)
inline Built build(const std::vector<Tri>& tris, uint32_t max_top, double pad)
{
    Built out;
    if (tris.empty())
        return out;
    detail::Builder b(tris);
    b.tmp.reserve(tris.size() * 2);
    b.build(0, (uint32_t)tris.size());
    const std::vector<detail::Tmp>& t = b.tmp;
    const uint32_t n = (uint32_t)t.size();

    // physical order: breadth-first prefix of at most max_top nodes, then depth-first
    std::vector<uint32_t> pos(n, kEnd), at;
    at.reserve(n);
    std::vector<int> frontier;
    {
        std::queue<int> q;
        q.push(0);
        while (!q.empty() && at.size() + q.size() <= max_top) {
            const int u = q.front();
            q.pop();
            pos[u] = (uint32_t)at.size();
            at.push_back((uint32_t)u);
            if (t[u].left >= 0) { q.push(t[u].left); q.push(t[u].right); }
        }
        while (!q.empty()) { frontier.push_back(q.front()); q.pop(); }
    }
    out.top = (uint32_t)at.size();
    std::vector<int> stack;
    for (auto it = frontier.rbegin(); it != frontier.rend(); ++it)
        stack.push_back(*it);
    while (!stack.empty()) {
        const int u = stack.back();
        stack.pop_back();
        pos[u] = (uint32_t)at.size();
        at.push_back((uint32_t)u);
        if (t[u].left >= 0) { stack.push_back(t[u].right); stack.push_back(t[u].left); }
    }

    // links: miss(root) = end; left child's miss = right child; right child's miss = parent's miss
    std::vector<uint32_t> miss(n, kEnd);
    {
        std::vector<int> st{0};
        while (!st.empty()) {
            const int u = st.back();
            st.pop_back();
            if (t[u].left >= 0) {
                miss[t[u].left] = pos[t[u].right];
                miss[t[u].right] = miss[u];
                st.push_back(t[u].left);
                st.push_back(t[u].right);
            }
        }
    }
    out.order = b.idx;
    out.nodes.resize(n);
    for (uint32_t u = 0; u < n; ++u) {
        Node& nd = out.nodes[pos[u]];
        for (int a = 0; a < 3; ++a) { nd.lo[a] = t[u].lo[a] - pad; nd.hi[a] = t[u].hi[a] + pad; }
        nd.miss = miss[u];
        nd.hit = t[u].left >= 0 ? pos[t[u].left] : (0x80000000u | (t[u].first << 3) | t[u].count);
    }
    return out;
}

} // namespace drt_bvh
