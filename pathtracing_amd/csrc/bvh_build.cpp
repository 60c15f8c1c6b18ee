// bvh_build.cpp — binned-SAH BVH2 build (16 bins/axis), optional SAH-area collapse to BVH4, breadth-first
// node layout (top levels contiguous => stageable in LDS, one 128-B line per BVH4 node). docs/SPEC.md §4.1.
#include "bvh_build.h"
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <limits>
#include <thread>

namespace ptrt {
namespace {

// Bins of the SAH sweep. Measured on the 1M-triangle Cornell (1080p / 64 spp): 16 / 32 / 64 bins = 7.72 / 7.63 / 7.60 node visits per
// ray, 17.60 / 17.48 / 17.37 ms per frame, 0.46 / 0.52 / 0.58 s per commit (the soup: 29.05 / 28.95 / 28.91 visits, 1.12 / 1.25 / 1.48 s).
#ifndef PT_SAH_BINS
#define PT_SAH_BINS 64
#endif
constexpr int kBins = PT_SAH_BINS;
constexpr uint32_t kMaxLeaf = 4;
constexpr int32_t kEmpty = 0x7fffffff;
constexpr float kInf = std::numeric_limits<float>::infinity();

struct Box {
    float lo[3], hi[3];
    void reset() { for (int k = 0; k < 3; ++k) { lo[k] = kInf; hi[k] = -kInf; } }
    void grow(const Box &b) { for (int k = 0; k < 3; ++k) { lo[k] = std::min(lo[k], b.lo[k]); hi[k] = std::max(hi[k], b.hi[k]); } }
    void grow(const float p[3]) { for (int k = 0; k < 3; ++k) { lo[k] = std::min(lo[k], p[k]); hi[k] = std::max(hi[k], p[k]); } }
    float area() const
    {
        const float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
        return (dx < 0.f) ? 0.f : 2.f * (dx * dy + dy * dz + dz * dx);
    }
};
struct Prim { Box box; float c[3]; };
struct Tmp { Box box; int32_t left, right; uint32_t first, count; }; // count > 0: leaf over idx[first, first+count)

inline float pad_of(float c) { return 1e-6f * std::max(1.0f, std::fabs(c)); }

inline Box tri_box(const float *verts9, uint32_t id) // the padded leaf box of docs/SPEC.md §4.1
{
    const float *p = verts9 + (size_t)id * 9;
    Box b;
    for (int k = 0; k < 3; ++k) {
        const float lo = std::min(p[k], std::min(p[3 + k], p[6 + k])), hi = std::max(p[k], std::max(p[3 + k], p[6 + k]));
        b.lo[k] = lo - pad_of(lo); b.hi[k] = hi + pad_of(hi);
    }
    return b;
}

struct Builder {
    const std::vector<Prim> &prims;
    std::vector<uint32_t> &idx;
    std::vector<Tmp> nodes;
    uint32_t max_leaf = kMaxLeaf; // 1: every primitive its own leaf (the top-level build over LBVH clusters)
    // Parallel build: ranges of more than `defer_above` primitives that reach depth `defer_depth` are not built but recorded in
    // `deferred` (a placeholder node holds their place); build_parallel() builds them on their own threads and splices them in.
    int defer_depth = -1; uint32_t defer_above = 0;
    struct Deferred { int32_t node; uint32_t b, e; int depth; };
    std::vector<Deferred> deferred;
    Builder(const std::vector<Prim> &p, std::vector<uint32_t> &i, size_t reserve) : prims(p), idx(i) { nodes.reserve(reserve); }
    Builder(const std::vector<Prim> &p, std::vector<uint32_t> &i) : Builder(p, i, p.size()) {}

    int32_t make_leaf(uint32_t b, uint32_t e, const Box &box)
    {
        Tmp t; t.box = box; t.left = t.right = -1; t.first = b; t.count = e - b;
        nodes.push_back(t);
        return (int32_t)nodes.size() - 1;
    }

    int32_t build(uint32_t b, uint32_t e, int depth)
    {
        const uint32_t n = e - b;
        if (depth == defer_depth && n > defer_above) {
            nodes.push_back(Tmp{});
            deferred.push_back(Deferred{ (int32_t)nodes.size() - 1, b, e, depth });
            return (int32_t)nodes.size() - 1;
        }
        Box box, cb;
        box.reset(); cb.reset();
        for (uint32_t i = b; i < e; ++i) { box.grow(prims[idx[i]].box); cb.grow(prims[idx[i]].c); }
        if (n == 1) return make_leaf(b, e, box);

        uint32_t mid = 0;
        bool have_split = false;
        if (depth < 40) {
            float best = kInf; int best_axis = -1, best_bin = -1;
            for (int ax = 0; ax < 3; ++ax) {
                const float ext = cb.hi[ax] - cb.lo[ax];
                if (!(ext > 0.f)) continue;
                Box bb[kBins]; uint32_t cnt[kBins];
                for (int k = 0; k < kBins; ++k) { bb[k].reset(); cnt[k] = 0; }
                const float sc = (float)kBins / ext;
                for (uint32_t i = b; i < e; ++i) {
                    const Prim &p = prims[idx[i]];
                    int k = (int)((p.c[ax] - cb.lo[ax]) * sc);
                    k = std::min(std::max(k, 0), kBins - 1);
                    bb[k].grow(p.box); cnt[k]++;
                }
                float ra[kBins]; uint32_t rc[kBins];
                Box acc; acc.reset(); uint32_t c = 0;
                for (int k = kBins - 1; k >= 1; --k) { acc.grow(bb[k]); c += cnt[k]; ra[k] = acc.area(); rc[k] = c; }
                acc.reset(); c = 0;
                for (int k = 0; k < kBins - 1; ++k) {
                    acc.grow(bb[k]); c += cnt[k];
                    if (c == 0 || rc[k + 1] == 0) continue;
                    const float cost = acc.area() * (float)c + ra[k + 1] * (float)rc[k + 1];
                    if (cost < best) { best = cost; best_axis = ax; best_bin = k; }
                }
            }
            const float leaf_cost = box.area() * (float)n;
            if (best_axis >= 0 && !(n <= max_leaf && best >= leaf_cost)) {
                const float ext = cb.hi[best_axis] - cb.lo[best_axis], sc = (float)kBins / ext, lo = cb.lo[best_axis];
                const int ax = best_axis, bin = best_bin;
                auto it = std::partition(idx.begin() + b, idx.begin() + e, [&](uint32_t id) {
                    int k = (int)((prims[id].c[ax] - lo) * sc);
                    k = std::min(std::max(k, 0), kBins - 1);
                    return k <= bin;
                });
                mid = (uint32_t)(it - idx.begin());
                have_split = mid > b && mid < e;
            } else if (n <= max_leaf) return make_leaf(b, e, box);
        }
        if (!have_split) {
            if (n <= max_leaf) return make_leaf(b, e, box);
            int ax = 0; // median split along the widest centroid axis (ties: index order)
            for (int k = 1; k < 3; ++k) if (cb.hi[k] - cb.lo[k] > cb.hi[ax] - cb.lo[ax]) ax = k;
            mid = b + n / 2;
            std::nth_element(idx.begin() + b, idx.begin() + mid, idx.begin() + e, [&](uint32_t x, uint32_t y) {
                const float cx = prims[x].c[ax], cy = prims[y].c[ax];
                return cx < cy || (cx == cy && x < y);
            });
        }
        const int32_t me = (int32_t)nodes.size();
        nodes.push_back(Tmp{});
        const int32_t l = build(b, mid, depth + 1);
        const int32_t r = build(mid, e, depth + 1);
        Tmp &t = nodes[me];
        t.box = box; t.left = l; t.right = r; t.first = 0; t.count = 0;
        return me;
    }
};

// The same tree as Builder::build(0, n, 0), built on several threads: the top levels serially down to depth `kParDepth`, every
// big range found there on a thread of its own (disjoint ranges of `idx`, a private node vector each), then spliced behind the top
// part with the indices shifted. Splits depend only on a range's own primitives, so the topology — and with it the emitted blob
// — is the serial one; only the order of the nodes in `nodes` differs, which nothing reads.
constexpr int kParDepth = 4;
int32_t build_parallel(Builder &B, uint32_t n, uint32_t serial_below = 1u << 16, uint32_t defer_above = 4096)
{
    const unsigned hw = std::thread::hardware_concurrency();
    if (n < serial_below || hw < 2) return B.build(0, n, 0);
    B.defer_depth = kParDepth; B.defer_above = defer_above;
    const int32_t root = B.build(0, n, 0);
    B.defer_depth = -1;
    const std::vector<Builder::Deferred> jobs = B.deferred;
    std::vector<Builder> subs;
    subs.reserve(jobs.size());
    for (const auto &j : jobs) { subs.emplace_back(B.prims, B.idx, (size_t)(j.e - j.b)); subs.back().max_leaf = B.max_leaf; }
    std::vector<int32_t> roots(jobs.size(), 0);
    std::vector<std::thread> th;
    for (size_t k = 0; k < jobs.size(); ++k)
        th.emplace_back([&, k] { roots[k] = subs[k].build(jobs[k].b, jobs[k].e, jobs[k].depth); });
    for (auto &t : th) t.join();
    for (size_t k = 0; k < jobs.size(); ++k) {
        const int32_t off = (int32_t)B.nodes.size();
        for (Tmp t : subs[k].nodes) {
            if (!t.count) { t.left += off; t.right += off; }
            B.nodes.push_back(t);
        }
        B.nodes[(size_t)jobs[k].node] = B.nodes[(size_t)(off + roots[k])]; // the placeholder becomes the subtree's root (children already shifted)
    }
    return root;
}

// Binary tree (tmp nodes, leaves = ranges of idx) -> blob: collapse to `width` children per node by opening the child of
// largest area, lay nodes out breadth-first, emit triangles in leaf order, compute depth and the worst-case stack need.
// octant_slots (width 8, layout BVH8O): a node's children are placed in the slot whose index names the corner of the node they sit in
// — bit k of the slot = child lies towards +axis k — so that `slot ^ (sign bits of the ray direction)` is a front-to-back order
// and the traversal needs no distance sort (the child-to-slot assignment of Ylitie, Karras & Laine, "Efficient incoherent ray
// traversal on GPUs through compressed wide BVHs", HPG 2017: greedy on the projection of the child's centre onto the slot's diagonal).
void emit_blob(const std::vector<Tmp> &tn, int32_t root, const std::vector<uint32_t> &idx, const float *verts9, const uint32_t *mats,
               uint32_t n_tris, uint32_t width, BvhBlob &out, bool octant_slots = false)
{
    struct Pending { int32_t kids[8]; int nk; }; // width <= 8; kids[c] < 0: empty slot (octant_slots leaves holes anywhere)
    std::vector<Pending> pend;
    pend.reserve(tn.size());
    // Which binary nodes become the children of a wide node: chosen by the SAH-optimal dynamic programme of Ylitie, Karras & Laine
    // (HPG 2017, §3.1) — round 1-2 opened the child of largest area until the node was full (PTRT_COLLAPSE=area keeps that for
    // comparison: 1M-triangle Cornell 7.60 -> 7.48 node visits per ray with 16 % fewer nodes, soup 28.92 -> 28.72).
    // cost[n][i-1] = least SAH cost of the subtree under binary node n when it may take up to i child slots of its parent (i = 1: n is
    // itself a node, or a leaf — the binary builder's leaves stay leaves, so the triangle term is a constant of the tree).
    const char *collapse_env = getenv("PTRT_COLLAPSE");
    const bool dp = !(collapse_env && std::strcmp(collapse_env, "area") == 0);
    const float c_tri = 0.6f; // a triangle test relative to a node visit
    std::vector<float> cost;
    auto C = [&](int32_t n, uint32_t i) -> float & { return cost[(size_t)n * width + (i - 1)]; };
    auto best_split = [&](int32_t n, uint32_t j, uint32_t &kbest) { // least cost of giving j >= 2 slots to the two children of n
        float best = kInf; kbest = 1;
        for (uint32_t k = 1; k < j; ++k) { const float v = C(tn[n].left, k) + C(tn[n].right, j - k); if (v < best) { best = v; kbest = k; } }
        return best;
    };
    if (dp) {
        cost.assign(tn.size() * width, 0.f);
        for (int64_t n = (int64_t)tn.size() - 1; n >= 0; --n) { // children have larger indices than their parents
            const float a = tn[n].box.area() / std::max(tn[root].box.area(), 1e-30f);
            if (tn[n].count) { for (uint32_t i = 1; i <= width; ++i) C((int32_t)n, i) = a * c_tri * (float)tn[n].count; continue; }
            if (tn[n].left < 0 || tn[n].right < 0) continue; // (placeholder of the parallel build: never reachable)
            uint32_t k;
            C((int32_t)n, 1) = a + best_split((int32_t)n, width, k);
            for (uint32_t i = 2; i <= width; ++i) C((int32_t)n, i) = std::min(best_split((int32_t)n, i, k), C((int32_t)n, i - 1));
        }
    }
    std::function<void(int32_t, uint32_t, Pending &)> gather = [&](int32_t n, uint32_t j, Pending &p) { // the <= j roots binary node n contributes
        if (tn[n].count || j == 1) { p.kids[p.nk++] = n; return; }
        uint32_t k;
        const float split = best_split(n, j, k);
        if (C(n, j - 1) <= split) { gather(n, j - 1, p); return; }
        gather(tn[n].left, k, p); gather(tn[n].right, j - k, p);
    };
    auto expand = [&](int32_t t) { // children of the output node made from tmp node t
        Pending p; p.nk = 0;
        for (int i = 0; i < 8; ++i) p.kids[i] = -1;
        if (tn[t].count) { p.kids[p.nk++] = t; return p; } // (root is a leaf) single child
        if (dp) {
            uint32_t k; (void)best_split(t, width, k);
            gather(tn[t].left, k, p); gather(tn[t].right, width - k, p);
        } else {
        p.kids[p.nk++] = tn[t].left; p.kids[p.nk++] = tn[t].right;
        }
        while (!dp && p.nk < (int)width) {
            int best = -1; float ba = -1.f;
            for (int i = 0; i < p.nk; ++i)
                if (!tn[p.kids[i]].count) { const float a = tn[p.kids[i]].box.area(); if (a > ba) { ba = a; best = i; } }
            if (best < 0) break;
            const int32_t c = p.kids[best];
            for (int i = p.nk; i > best + 1; --i) p.kids[i] = p.kids[i - 1];
            p.kids[best] = tn[c].left; p.kids[best + 1] = tn[c].right; p.nk++;
        }
        if (octant_slots && width == 8) {
            float cen[8][3], mid[3];
            Box all; all.reset();
            for (int i = 0; i < p.nk; ++i) all.grow(tn[p.kids[i]].box);
            for (int a = 0; a < 3; ++a) mid[a] = 0.5f * (all.lo[a] + all.hi[a]);
            for (int i = 0; i < p.nk; ++i) for (int a = 0; a < 3; ++a) cen[i][a] = 0.5f * (tn[p.kids[i]].box.lo[a] + tn[p.kids[i]].box.hi[a]) - mid[a];
            int32_t placed[8]; for (int sl = 0; sl < 8; ++sl) placed[sl] = -1;
            bool done[8] = {};
            for (int round = 0; round < p.nk; ++round) { // greedy: the (child, free slot) pair of largest projection; ties: lowest child, lowest slot
                int bc = -1, bs = -1; float bv = -kInf;
                for (int i = 0; i < p.nk; ++i) {
                    if (done[i]) continue;
                    for (int sl = 0; sl < 8; ++sl) {
                        if (placed[sl] >= 0) continue;
                        const float v = (sl & 1 ? cen[i][0] : -cen[i][0]) + (sl & 2 ? cen[i][1] : -cen[i][1]) + (sl & 4 ? cen[i][2] : -cen[i][2]);
                        if (v > bv) { bv = v; bc = i; bs = sl; }
                    }
                }
                placed[bs] = p.kids[bc]; done[bc] = true;
            }
            for (int sl = 0; sl < 8; ++sl) p.kids[sl] = placed[sl];
            p.nk = 8;
        } else for (int i = p.nk; i < 8; ++i) p.kids[i] = -1;
        return p;
    };
    pend.push_back(expand(root));
    out.tris.reserve(n_tris);
    out.slots.reserve(tn.size() * width);
    const float root_area = std::max(tn[root].box.area(), 1e-30f);
    double sah = 0.0;
    for (size_t i = 0; i < pend.size(); ++i) { // pend grows while we iterate: index i = output node i
        const Pending p = pend[i];
        BvhSlot s[8];
        for (uint32_t c = 0; c < width; ++c) { std::memset(&s[c], 0, sizeof(BvhSlot)); s[c].ref = kEmpty; }
        for (int c = 0; c < p.nk; ++c) {
            if (p.kids[c] < 0) continue;
            const Tmp &k = tn[p.kids[c]];
            for (int a = 0; a < 3; ++a) { s[c].lo[a] = k.box.lo[a]; s[c].hi[a] = k.box.hi[a]; }
            if (k.count) {
                const uint32_t first = (uint32_t)out.tris.size();
                for (uint32_t j = 0; j < k.count; ++j) {
                    const uint32_t id = idx[k.first + j];
                    const float *v = verts9 + (size_t)id * 9;
                    BvhTri t; std::memset(&t, 0, sizeof t);
                    for (int a = 0; a < 3; ++a) { t.v0[a] = v[a]; t.e1[a] = v[3 + a] - v[a]; t.e2[a] = v[6 + a] - v[a]; }
                    t.id = id; t.mat = mats ? mats[id] : 0u;
                    out.tris.push_back(t);
                }
                s[c].ref = (int32_t)~((first << 3) | (k.count - 1u));
                sah += (double)(k.box.area() / root_area) * k.count;
            } else {
                s[c].ref = (int32_t)pend.size();
                pend.push_back(expand(p.kids[c]));
                sah += (double)(k.box.area() / root_area);
            }
        }
        for (uint32_t c = 0; c < width; ++c) out.slots.push_back(s[c]);
    }
    out.n_nodes = (uint32_t)pend.size();
    out.sah_cost = (float)sah;

    // ---- depth and worst-case traversal-stack need (children have larger indices than parents)
    std::vector<uint32_t> depth(out.n_nodes, 1), need(out.n_nodes, 0);
    for (int64_t i = (int64_t)out.n_nodes - 1; i >= 0; --i) {
        uint32_t k = 0, dmax = 1, nmax = 0;
        for (uint32_t c = 0; c < width; ++c) {
            const int32_t r = out.slots[(size_t)i * width + c].ref;
            if (r == kEmpty) continue;
            ++k;
            if (r >= 0) { dmax = std::max(dmax, depth[r]); nmax = std::max(nmax, need[r]); }
        }
        depth[i] = dmax + 1;
        need[i] = (k ? k - 1 : 0) + nmax;
    }
    out.max_depth = depth[0];
    out.stack_need = need[0];
}

// Memory order of the node array (and optionally of the triangle array): pure renaming — refs change, the tree, the boxes and with
// them every picture and visit counter do not (docs/SPEC.md §4.1: the order of nodes and triangles is not part of the contract).
// Why it matters: beyond L2 the memory system moves 128-byte lines, a BVH4Q node is 64 bytes, so every node shares its line with
// one neighbour; which neighbour decides how many of a ray's node fetches are new lines (layout experiments).
//   order 0 : breadth-first as emitted (the children of a node are consecutive: a line holds two siblings)
//   order 1 : depth-first pre-order (a line holds a node and its first inner child, or two nodes of neighbouring subtrees)
//   order 2 : parent + largest child pairs, pairs in breadth-first order: a 128-byte line holds a node and the inner child of
//             largest surface area (the one a ray that visits the node most probably visits too); nodes without inner children
//             pair up among themselves in queue order
//   order 3 : the same pairs in depth-first order (pairs of one subtree contiguous: van-Emde-Boas-like blocks of two)
//   +16     : triangles re-emitted in the order the new node array references them
void reorder_blob(BvhBlob &b, uint32_t mode)
{
    const uint32_t W = b.width, n = b.n_nodes, order = mode & 15u;
    if (n < 2 || (order == 0 && !(mode & 16u))) return;
    std::vector<uint32_t> ord; // new position -> old index
    ord.reserve(n);
    auto slot = [&](uint32_t i, uint32_t c) -> const BvhSlot & { return b.slots[(size_t)i * W + c]; };
    if (order == 0) { for (uint32_t i = 0; i < n; ++i) ord.push_back(i); }
    else if (order == 1) {
        std::vector<uint32_t> st{ 0u };
        while (!st.empty()) {
            const uint32_t x = st.back(); st.pop_back();
            ord.push_back(x);
            for (int c = (int)W - 1; c >= 0; --c) { const int32_t r = slot(x, (uint32_t)c).ref; if (r >= 0 && r != kEmpty) st.push_back((uint32_t)r); }
        }
    } else {
        std::vector<uint32_t> heads{ 0u }; // FIFO (order 2: read index) or LIFO (order 3)
        size_t rd = 0;
        int64_t pending = -1;
        while (order == 2 ? rd < heads.size() : !heads.empty()) {
            uint32_t x;
            if (order == 2) x = heads[rd++]; else { x = heads.back(); heads.pop_back(); }
            int best = -1; float ba = -1.f;
            for (uint32_t c = 0; c < W; ++c) {
                const BvhSlot &k = slot(x, c);
                if (k.ref < 0 || k.ref == kEmpty) continue;
                Box bx; for (int a = 0; a < 3; ++a) { bx.lo[a] = k.lo[a]; bx.hi[a] = k.hi[a]; }
                const float ar = bx.area();
                if (ar > ba) { ba = ar; best = (int)c; }
            }
            if (best < 0) { // no inner child: shares a line with the next node of its kind
                if (pending >= 0) { ord.push_back((uint32_t)pending); ord.push_back(x); pending = -1; } else pending = x;
                continue;
            }
            const uint32_t y = (uint32_t)slot(x, (uint32_t)best).ref;
            ord.push_back(x); ord.push_back(y);
            std::vector<uint32_t> next;
            for (uint32_t c = 0; c < W; ++c) { const int32_t r = slot(x, c).ref; if (r >= 0 && r != kEmpty && (int)c != best) next.push_back((uint32_t)r); }
            for (uint32_t c = 0; c < W; ++c) { const int32_t r = slot(y, c).ref; if (r >= 0 && r != kEmpty) next.push_back((uint32_t)r); }
            if (order == 2) heads.insert(heads.end(), next.begin(), next.end());
            else heads.insert(heads.end(), next.rbegin(), next.rend());
        }
        if (pending >= 0) ord.push_back((uint32_t)pending);
    }
    if (ord.size() != n) return; // (cannot happen for a tree) leave the blob as it is
    std::vector<uint32_t> pos(n);
    for (uint32_t i = 0; i < n; ++i) pos[ord[i]] = i;
    std::vector<BvhSlot> ns((size_t)n * W);
    for (uint32_t i = 0; i < n; ++i)
        for (uint32_t c = 0; c < W; ++c) {
            BvhSlot k = slot(ord[i], c);
            if (k.ref >= 0 && k.ref != kEmpty) k.ref = (int32_t)pos[(uint32_t)k.ref];
            ns[(size_t)i * W + c] = k;
        }
    b.slots.swap(ns);
    if (mode & 16u) {
        std::vector<BvhTri> nt;
        nt.reserve(b.tris.size());
        for (size_t i = 0; i < b.slots.size(); ++i) {
            BvhSlot &k = b.slots[i];
            if (k.ref >= 0) continue; // inner node or kEmpty (0x7fffffff)
            const uint32_t enc = (uint32_t)~k.ref, first = enc >> 3, cnt = (enc & 7u) + 1u, nf = (uint32_t)nt.size();
            for (uint32_t j = 0; j < cnt; ++j) nt.push_back(b.tris[first + j]);
            k.ref = (int32_t)~((nf << 3) | (cnt - 1u));
        }
        b.tris.swap(nt);
    }
}

} // namespace

void build_bvh(const float *verts9, const uint32_t *mats, uint32_t n_tris, uint32_t width, BvhBlob &out, bool octant_slots)
{
    const auto t0 = std::chrono::steady_clock::now();
    out = BvhBlob{};
    out.width = width;
    if (n_tris == 0) return;

    std::vector<Prim> prims(n_tris);
    std::vector<uint32_t> idx(n_tris);
    for (uint32_t i = 0; i < n_tris; ++i) {
        const float *p = verts9 + (size_t)i * 9;
        Prim &pr = prims[i];
        for (int k = 0; k < 3; ++k) {
            const float lo = std::min(p[k], std::min(p[3 + k], p[6 + k])), hi = std::max(p[k], std::max(p[3 + k], p[6 + k]));
            pr.box.lo[k] = lo - pad_of(lo);
            pr.box.hi[k] = hi + pad_of(hi);
            pr.c[k] = 0.5f * (lo + hi);
        }
        idx[i] = i;
    }
    Builder B(prims, idx);
    const int32_t root = build_parallel(B, n_tris);
    emit_blob(B.nodes, root, idx, verts9, mats, n_tris, width, out, octant_slots);
    if (const char *e = getenv("PTRT_NODE_ORDER")) reorder_blob(out, (uint32_t)atoi(e)); // developer aid: layout experiments (tools/exp_order.py)
    out.build_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}

// A binary LBVH built on the device (lbvh.hip) -> blob, in two storeys:
//   top    : the LBVH is cut where a subtree holds at most kClusterTris triangles; the host's binned-SAH builder makes a binary tree
//            over those clusters (65 k boxes for 1M triangles, built on threads: a few milliseconds). Every ray crosses the top levels, and Morton splits are at
//            their worst there (1M-triangle Cornell: 9.15 -> 7.9 node visits per ray).
//   bottom : inside a cluster the device's topology and boxes are kept; subtrees of at most kMaxLeaf triangles become leaves unless
//            splitting them lowers the SAH cost — the leaf rule of Builder::build (their triangles are contiguous in Morton order).
#ifndef PT_LBVH_CLUSTER
#define PT_LBVH_CLUSTER 32
#endif
constexpr uint32_t kClusterTris = PT_LBVH_CLUSTER; // 0: no SAH storey, the LBVH as it is
void build_bvh_from_binary(const BinaryBvh &bt, const float *verts9, const uint32_t *mats, uint32_t n_tris, uint32_t width, BvhBlob &out, bool octant_slots)
{
    const auto t0 = std::chrono::steady_clock::now();
    out = BvhBlob{};
    out.width = width;
    if (n_tris < 2 || bt.order.size() != n_tris) return;
    auto node_box = [&](int32_t c) {
        Box b;
        if (c >= 0) for (int k = 0; k < 3; ++k) { b.lo[k] = bt.box[(size_t)c * 6 + k]; b.hi[k] = bt.box[(size_t)c * 6 + 3 + k]; }
        else b = tri_box(verts9, bt.order[(uint32_t)~c]); // single-triangle leaf: its padded box, exactly as the device made it
        return b;
    };
    auto node_count = [&](int32_t c) { return c >= 0 ? bt.last[c] - bt.first[c] + 1 : 1u; };

    // ---- cut: cluster roots, in Morton order
    std::vector<int32_t> clusters;
    {
        std::vector<int32_t> st{ 0 };
        while (!st.empty()) {
            const int32_t c = st.back(); st.pop_back();
            if (c < 0 || kClusterTris == 0 || node_count(c) <= kClusterTris) clusters.push_back(c);
            else { st.push_back(bt.right[c]); st.push_back(bt.left[c]); }
        }
    }
    // ---- top storey: binned SAH over the cluster boxes, one cluster per leaf
    std::vector<Prim> prims(clusters.size());
    std::vector<uint32_t> cidx(clusters.size());
    for (size_t i = 0; i < clusters.size(); ++i) {
        prims[i].box = node_box(clusters[i]);
        for (int k = 0; k < 3; ++k) prims[i].c[k] = 0.5f * (prims[i].box.lo[k] + prims[i].box.hi[k]);
        cidx[i] = (uint32_t)i;
    }
    Builder top(prims, cidx);
    top.max_leaf = 1;
    const int32_t root = top.build(0, (uint32_t)clusters.size(), 0);
    std::vector<Tmp> tn = top.nodes;
    tn.reserve(tn.size() + (size_t)n_tris * 2);

    // ---- bottom storey: every top leaf is replaced, in place, by its cluster's LBVH subtree (iterative: deep chains stay off the C stack)
    struct Work { int32_t node; int32_t at; }; // convert LBVH node `node` into tn[at]
    std::vector<Work> work;
    const size_t n_top = tn.size();
    for (size_t i = 0; i < n_top; ++i)
        if (tn[i].count) work.push_back({ clusters[cidx[tn[i].first]], (int32_t)i });
    while (!work.empty()) {
        const Work w = work.back(); work.pop_back();
        Tmp t; t.left = t.right = -1; t.first = 0; t.count = 0;
        t.box = node_box(w.node);
        if (w.node >= 0) {
            const uint32_t f = bt.first[w.node], cnt = node_count(w.node);
            if (cnt <= kMaxLeaf) {
                float split_cost = 0.f;
                for (const int32_t c : { bt.left[w.node], bt.right[w.node] }) split_cost += node_box(c).area() * (float)node_count(c);
                if (!(split_cost < t.box.area() * (float)cnt)) { t.first = f; t.count = cnt; }
            }
            if (t.count == 0) {
                t.left = (int32_t)tn.size(); t.right = t.left + 1;
                tn.push_back(Tmp{}); tn.push_back(Tmp{});
                work.push_back({ bt.left[w.node], t.left });
                work.push_back({ bt.right[w.node], t.right });
            }
        } else { t.first = (uint32_t)~w.node; t.count = 1; }
        tn[w.at] = t;
    }
    const auto t1 = std::chrono::steady_clock::now();
    emit_blob(tn, root, bt.order, verts9, mats, n_tris, width, out, octant_slots);
    if (getenv("PTRT_TIMING")) // developer aid
        fprintf(stderr, "ptrt commit: lbvh device %.2f ms, cut + SAH top + conversion %.2f ms, emit_blob %.2f ms\n", bt.device_ms,
                std::chrono::duration<double, std::milli>(t1 - t0).count(), std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t1).count());
    out.build_ms = bt.device_ms + std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}

void build_sah_over_boxes(const float *boxes6, uint32_t n, std::vector<int32_t> &left, std::vector<int32_t> &right, std::vector<float> &node_boxes6, int32_t &root)
{
    std::vector<Prim> prims(n);
    std::vector<uint32_t> idx(n);
    for (uint32_t i = 0; i < n; ++i) {
        for (int k = 0; k < 3; ++k) { prims[i].box.lo[k] = boxes6[(size_t)i * 6 + k]; prims[i].box.hi[k] = boxes6[(size_t)i * 6 + 3 + k]; }
        for (int k = 0; k < 3; ++k) prims[i].c[k] = 0.5f * (prims[i].box.lo[k] + prims[i].box.hi[k]);
        idx[i] = i;
    }
    Builder b(prims, idx);
    b.max_leaf = 1;
    const int32_t r = build_parallel(b, n, 8192, 256);
    // internal Tmp nodes -> compact internal numbering; leaves -> ~box index
    std::vector<int32_t> number(b.nodes.size(), -1);
    int32_t n_int = 0;
    for (size_t i = 0; i < b.nodes.size(); ++i) if (!b.nodes[i].count) number[i] = n_int++;
    auto ref = [&](int32_t t) { return b.nodes[t].count ? ~(int32_t)idx[b.nodes[t].first] : number[t]; };
    left.assign(n_int, 0); right.assign(n_int, 0); node_boxes6.assign((size_t)n_int * 6, 0.f);
    for (size_t i = 0; i < b.nodes.size(); ++i) {
        if (b.nodes[i].count) continue;
        const int32_t k = number[i];
        left[k] = ref(b.nodes[i].left); right[k] = ref(b.nodes[i].right);
        for (int a = 0; a < 3; ++a) { node_boxes6[(size_t)k * 6 + a] = b.nodes[i].box.lo[a]; node_boxes6[(size_t)k * 6 + 3 + a] = b.nodes[i].box.hi[a]; }
    }
    root = ref(r);
}

// ---- BVH4Q: 64-byte nodes, child boxes quantised to 8 bits per coordinate on a per-node power-of-two grid
// (docs/SPEC.md §4.1). decode(q) = fma((float)q, scale, origin) must enclose the float box it replaces; the
// quantiser checks that with the very expression the traversal uses and nudges q outward when rounding bites.
namespace {
inline float scale_of(uint8_t e) { uint32_t b = (uint32_t)e << 23; float f; std::memcpy(&f, &b, 4); return f; }
} // namespace

// N = 4: 64-byte nodes (layout 68); N = 8: 128-byte nodes (layout 72), same scheme with 8-byte coordinate groups
template <int N>
static void quantize_nodes(const BvhBlob &in, std::vector<uint8_t> &out)
{
    constexpr size_t kStride = N == 4 ? 64 : 128, kQ = 16 + 4 * N; // quantised coordinates start after origin|exps and the refs
    out.assign((size_t)in.n_nodes * kStride, 0);
    auto range = [&](uint32_t i0, uint32_t i1) { // nodes are independent of one another
    for (uint32_t i = i0; i < i1; ++i) {
        const BvhSlot *s = &in.slots[(size_t)i * N];
        uint8_t *nd = &out[(size_t)i * kStride];
        float org[3]; uint8_t ex[3];
        uint8_t qlo[3][N] = {}, qhi[3][N] = {};
        for (int k = 0; k < 3; ++k) {
            float lo = kInf, hi = -kInf;
            for (int c = 0; c < N; ++c) if (s[c].ref != kEmpty) { lo = std::min(lo, s[c].lo[k]); hi = std::max(hi, s[c].hi[k]); }
            if (!(lo <= hi)) { lo = hi = 0.f; } // node without children (cannot happen for a built tree)
            org[k] = lo;
            int e = 1;
            { // smallest power of two with 255*scale >= extent
                const float ext = hi - lo;
                int ee; const float m = std::frexp(ext / 255.0f, &ee); // ext/255 = m * 2^ee, m in [0.5,1)
                e = (ext > 0.f) ? ee + 127 - (m == 0.5f ? 1 : 0) : 1;
                e = std::min(std::max(e, 1), 254);
            }
            for (;;) { // quantise; widen the grid if a coordinate does not fit in 8 bits
                const float sc = scale_of((uint8_t)e);
                bool ok = true;
                for (int c = 0; c < N && ok; ++c) {
                    if (s[c].ref == kEmpty) continue;
                    int ql = (int)std::floor((s[c].lo[k] - lo) / sc), qh = (int)std::ceil((s[c].hi[k] - lo) / sc);
                    ql = std::min(std::max(ql, 0), 255); qh = std::min(std::max(qh, 0), 255);
                    while (ql > 0 && !(std::fmaf((float)ql, sc, lo) <= s[c].lo[k])) --ql;
                    while (qh < 255 && !(std::fmaf((float)qh, sc, lo) >= s[c].hi[k])) ++qh;
                    if (!(std::fmaf((float)ql, sc, lo) <= s[c].lo[k]) || !(std::fmaf((float)qh, sc, lo) >= s[c].hi[k])) { ok = false; break; }
                    qlo[k][c] = (uint8_t)ql; qhi[k][c] = (uint8_t)qh;
                }
                if (ok || e >= 254) break;
                ++e;
            }
            ex[k] = (uint8_t)e;
        }
        std::memcpy(nd + 0, org, 12);
        nd[12] = ex[0]; nd[13] = ex[1]; nd[14] = ex[2]; nd[15] = 0;
        for (int c = 0; c < N; ++c) std::memcpy(nd + 16 + 4 * c, &s[c].ref, 4);
        for (int k = 0; k < 3; ++k) { std::memcpy(nd + kQ + N * k, qlo[k], N); std::memcpy(nd + kQ + N * (3 + k), qhi[k], N); }
    }
    };
    const uint32_t nt = in.n_nodes < (1u << 15) ? 1u : std::min(16u, std::max(1u, std::thread::hardware_concurrency()));
    if (nt == 1) { range(0, in.n_nodes); return; }
    std::vector<std::thread> th;
    for (uint32_t t = 0; t < nt; ++t) th.emplace_back(range, (uint32_t)((uint64_t)in.n_nodes * t / nt), (uint32_t)((uint64_t)in.n_nodes * (t + 1) / nt));
    for (auto &t : th) t.join();
}

void quantize_bvh4(const BvhBlob &in, std::vector<uint8_t> &out) { quantize_nodes<4>(in, out); }
void quantize_bvh8(const BvhBlob &in, std::vector<uint8_t> &out) { quantize_nodes<8>(in, out); }

} // namespace ptrt
