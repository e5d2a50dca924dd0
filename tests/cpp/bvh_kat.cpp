// Known-answer checks of the host BVH builder and its 64-byte node encoding (drt_bvh.h):
// every triangle is in exactly one leaf and inside every box on its root path, as the DEVICE decodes
// the boxes; the 4-wide depth respects the stack bound of the traversal kernel; a brute-force
// closest hit is found by a walk over the encoded nodes.
#include "../../differentiable-renderer_amd/csrc/drt_bvh.h"

#include <cstdio>
#include <cstdlib>
#include <random>

using namespace drt_bvh;

static int failures = 0;
#define CHECK(c) do { if (!(c)) { std::printf("FAIL %s:%d %s\n", __FILE__, __LINE__, #c); ++failures; } } while (0)

static std::vector<Tri> soup(int n, unsigned seed, double spread, double size)
{
    std::mt19937 g(seed);
    std::uniform_real_distribution<double> u(-1.0, 1.0);
    std::vector<Tri> t(n);
    for (int i = 0; i < n; ++i) {
        for (int a = 0; a < 3; ++a) {
            t[i].v0[a] = spread * u(g) + 100.0;        // off-origin: f32 origins must round the right way
            t[i].e1[a] = size * u(g);
            t[i].e2[a] = size * u(g);
            t[i].n[a] = 0;
        }
        t[i].global = t[i].flat = (uint32_t)i;
        t[i].ids = 0;
    }
    return t;
}

static void tri_box(const Tri& t, double lo[3], double hi[3])
{
    for (int a = 0; a < 3; ++a) {
        const double p[3] = {t.v0[a], t.v0[a] + t.e1[a], t.v0[a] + t.e2[a]};
        lo[a] = std::min(p[0], std::min(p[1], p[2]));
        hi[a] = std::max(p[0], std::max(p[1], p[2]));
    }
}

struct Walk {
    const Built& b;
    const std::vector<QNode>& q;
    const std::vector<Tri>& tris;
    std::vector<int> seen;
    int max_depth = 0;
    size_t max_stack = 0;

    // every triangle below `link` must lie in [lo, hi] (the decoded box of the link in its parent)
    void visit(uint32_t link, const double lo[3], const double hi[3], int depth, size_t pending = 0)
    {
        max_stack = std::max(max_stack, pending);
        if (link & kLeaf) {
            const uint32_t first = (link & 0x7FFFFFFFu) >> 3, count = link & 7u;
            CHECK(count >= 1 && count <= (uint32_t)kMaxLeaf);
            for (uint32_t j = first; j < first + count; ++j) {
                const uint32_t ti = b.order[j];
                ++seen[ti];
                double tl[3], th[3];
                tri_box(tris[ti], tl, th);
                for (int a = 0; a < 3; ++a)
                    CHECK(lo[a] <= tl[a] && th[a] <= hi[a]);
            }
            return;
        }
        max_depth = std::max(max_depth, depth);
        CHECK(link < q.size());
        int real = 0, n_children = 0;
        for (int c = 0; c < kWidth; ++c)
            n_children += q[link].w[4 + c] != kLeaf;
        for (int c = 0; c < kWidth; ++c) {
            const uint32_t child = q[link].w[4 + c];
            if (child == kLeaf)
                continue;
            ++real;
            double cl[3], ch[3];
            decode(q[link], c, cl, ch);
            for (int a = 0; a < 3; ++a) {
                CHECK(cl[a] <= b.nodes[link].lo[c][a] && b.nodes[link].hi[c][a] <= ch[a]);   // contains the exact box
                // ... also as the device's f32 fma evaluates it
                float origin;
                memcpy(&origin, &q[link].w[a], 4);
                const float sc = grid_scale((q[link].w[3] >> (8 * a)) & 0xFFu);
                CHECK((double)fmaf((float)((q[link].w[8 + a] >> (8 * c)) & 0xFFu), sc, origin) <= b.nodes[link].lo[c][a]);
                CHECK((double)fmaf((float)((q[link].w[11 + a] >> (8 * c)) & 0xFFu), sc, origin) >= b.nodes[link].hi[c][a]);
                // and not absurdly loose: within two grid steps (+ f32 rounding at this magnitude) of the exact box
                const double slack = 2.0 * sc + 4.0 * 7.63e-6;
                CHECK(b.nodes[link].lo[c][a] - cl[a] <= slack && ch[a] - b.nodes[link].hi[c][a] <= slack);
            }
            visit(child, cl, ch, depth + 1, pending + (size_t)(n_children - 1));   // worst case: all siblings pushed
        }
        CHECK(real >= 2 || link == 0);
    }
};

// summed surface area of all child boxes, weighted by the triangles below: the SAH cost of the tree's leaves
static double leaf_sah(const Built& b)
{
    double cost = 0;
    for (const Node& nd : b.nodes)
        for (int c = 0; c < kWidth; ++c)
            if ((nd.child[c] & kLeaf) && nd.child[c] != kLeaf)
                cost += (double)(nd.child[c] & 7u) * detail::area(nd.lo[c], nd.hi[c]);
    return cost;
}

// nested scales: clusters of clusters (the builder's depth / stack bound is what this stresses)
static std::vector<Tri> nested(int n, unsigned seed)
{
    std::mt19937 g(seed);
    std::uniform_real_distribution<double> u(-1.0, 1.0);
    std::vector<Tri> t(n);
    for (int i = 0; i < n; ++i) {
        const int level = i % 7;                              // scale 4^-level, centres drifting along a line
        const double sc = std::pow(0.25, level);
        for (int a = 0; a < 3; ++a) {
            t[i].v0[a] = 3.0 * (1.0 - sc) + sc * u(g) + (a == 0 ? 0.0 : 0.1 * level);
            t[i].e1[a] = 0.05 * sc * u(g);
            t[i].e2[a] = 0.05 * sc * u(g);
            t[i].n[a] = 0;
        }
        t[i].global = t[i].flat = (uint32_t)i;
        t[i].ids = 0;
    }
    return t;
}

static void check_built(const std::vector<Tri>& tris, uint32_t max_top, int stack_entries = kStackEntries)
{
    const int n = (int)tris.size();
    const Built b = build(tris, max_top, 1e-5, stack_entries);
    std::vector<QNode> q(b.nodes.size());
    for (size_t i = 0; i < q.size(); ++i)
        q[i] = quantise(b.nodes[i]);
    CHECK(b.order.size() == tris.size());
    CHECK(b.top >= 1 && b.top <= std::max<uint32_t>(max_top, 1) && b.top <= b.nodes.size());
    Walk w{b, q, tris, std::vector<int>(tris.size(), 0)};
    const double lo[3] = {-INFINITY, -INFINITY, -INFINITY}, hi[3] = {INFINITY, INFINITY, INFINITY};
    w.visit(0, lo, hi, 1);
    for (int s : w.seen)
        CHECK(s == 1);
    // the device's per-lane stack: the builder's own figure is the worst case of this walk, and it fits
    CHECK((int)w.max_stack == b.stack_need);
    CHECK(b.stack_need <= stack_entries);
    CHECK(w.max_depth == b.wide_depth || b.nodes.size() == 1);
    std::printf("n=%d nodes=%zu top=%u binary depth=%d wide depth=%d stack need=%d splits sah/median=%d/%d\n", n,
                b.nodes.size(), b.top, b.depth, b.wide_depth, b.stack_need, b.sah_splits, b.median_splits);
}

static void check_tree(int n, unsigned seed, double spread, double size, uint32_t max_top)
{
    check_built(soup(n, seed, spread, size), max_top);
}

int main()
{
    check_tree(1, 1, 1.0, 0.1, 128);          // one leaf: a root with a single child
    check_tree(2, 2, 1.0, 0.1, 128);
    check_tree(5, 3, 1.0, 0.1, 128);
    check_tree(1000, 4, 1.0, 0.05, 128);
    check_tree(1000, 5, 1.0, 0.05, 4);        // tiny LDS prefix
    check_tree(50000, 6, 10.0, 0.02, 128);
    check_tree(3000, 7, 1e-3, 1e-5, 128);     // a tiny cluster far from the origin (f32 grid at ulp scale)
    check_tree(4096, 8, 0.0, 0.5, 128);       // all triangles on top of one another (degenerate splits)
    // clustered, multi-scale soups: the SAH must actually be in use (empty bins are the rule here) and beat a
    // median-only build; the stack bound must hold, also when it is made artificially tight
    for (int n : {172, 6000, 26000}) {
        const std::vector<Tri> t = nested(n, 9);
        check_built(t, 128);
        const Built sah = build(t, 128, 1e-5), median = build_bounded(t, 128, 1e-5, 1);
        CHECK(median.stack_need < sah.stack_need);
        check_built(t, 128, sah.stack_need - 2);   // forces rebuilds with a tighter depth bound (a balanced tree fits)
        CHECK(sah.sah_splits > 0 && median.sah_splits == 0);
        CHECK(leaf_sah(sah) < leaf_sah(median));
        std::printf("n=%d leaf SAH cost: sah %.4g, median-only %.4g\n", n, leaf_sah(sah), leaf_sah(median));
    }
    if (failures)
        std::printf("%d failures\n", failures);
    else
        std::printf("ok\n");
    return failures != 0;
}
