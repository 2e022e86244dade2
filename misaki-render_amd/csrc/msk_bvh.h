// msk_bvh.h — host-side BVH2 builder (binned SAH) producing the flattened layout the
// traversal kernels read.  Replaces rtcCommitScene (reference: src/librender/scene.cpp:201-212).
//
// Layout (all float4, 16-byte aligned, 64 B per record so one record = one half cache line):
//   node  n : [lo0.x lo1.x lo0.y lo1.y] [lo0.z lo1.z hi0.x hi1.x] [hi0.y hi1.y hi0.z hi1.z] [c0, c1, -, - (uint bits)]
//             (the two children's bounds interleaved: each pair feeds one v_pk_fma_f32 of the slab test)
//             child ref c: 0x80000000 | first_tri << 5 | count for a leaf (count <= 8),
//             otherwise the index of an inner node.
//   tri   t : [v0.xyz, prim (uint bits)] [e1.xyz, 0] [e2.xyz, 0] [Ng.xyz, 0]
//             in LEAF order; e1 = v0 - v1, e2 = v2 - v0, Ng = e2 x e1 — the precomputed form of
//             Embree 3's TriangleM / Moeller-Trumbore test; prim = scene-global triangle index.
//   bounds t: [lo.xyz, 0] [hi.xyz, 0], same order, a separate array (the 64-byte triangle records keep their cache-line
//             alignment): bounding box of (v0, v0 - e1, v0 + e2) grown by tri_pad — the bounds predicate of oracle
//             deviation D10 (an accepted hit point must lie inside), read only for accepted hits.
// Child boxes are padded by 2 tri_pad (1e-5 of the scene's scale = max(diagonal, largest |coordinate|): some forty times the
// slab tests' worst rounding error, which tests/test_padding_margin.py locates) so the slab test can never cull a
// triangle the (differently rounded) triangle test accepts: hit selection is by (t, prim) and
// therefore independent of the tree.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace mskbvh {

struct V3 { float x, y, z; };
static inline V3 vmin(V3 a, V3 b) { return {std::min(a.x, b.x), std::min(a.y, b.y), std::min(a.z, b.z)}; }
static inline V3 vmax(V3 a, V3 b) { return {std::max(a.x, b.x), std::max(a.y, b.y), std::max(a.z, b.z)}; }
struct Box {
    V3 lo{INFINITY, INFINITY, INFINITY}, hi{-INFINITY, -INFINITY, -INFINITY};
    void grow(V3 p) { lo = vmin(lo, p); hi = vmax(hi, p); }
    void grow(const Box &b) { lo = vmin(lo, b.lo); hi = vmax(hi, b.hi); }
    float area() const {
        float dx = hi.x - lo.x, dy = hi.y - lo.y, dz = hi.z - lo.z;
        return (dx < 0) ? 0.f : 2.f * (dx * dy + dy * dz + dz * dx);
    }
};
static inline float axis_of(V3 v, int a) { return a == 0 ? v.x : a == 1 ? v.y : v.z; }

struct Built {
    std::vector<float> nodes;   // 16 floats per node
    std::vector<float> tris;    // 16 floats per triangle, leaf order
    std::vector<float> bounds;  // 8 floats per triangle, leaf order
    uint32_t root_ref = 0;      // packed child ref of the root
    int max_depth = 0;
    // 4-wide form of the same tree for scenes that do not fit LDS (collapse4): 32 floats (128 B = one L2 line) per node,
    //   [lo.x x4] [lo.y x4] [lo.z x4] [hi.x x4] [hi.y x4] [hi.z x4] [child refs x4 (uint bits)] [unused]
    // child ref as above, or kEmpty4 for an unused slot.  Same padded boxes, same leaves, same triangles.
    std::vector<float> nodes4;
    uint32_t root_ref4 = 0;
    int max_depth4 = 0;
    // The same 4-wide nodes (same numbering, same child refs) with QUANTISED child boxes, 16 dwords (64 B) per node — what the
    // traversal of a tree in HBM/L2 reads: four 16-byte loads per visit instead of seven (the traversal is bound by the number
    // of per-lane load instructions, msk_kernels.h: node4q_step):
    //   dw 0-2 origin (low corner of the node's box), dw 3 scale.x   dw 4-5 scale.y, scale.z (extent / 255, rounded up)
    //   dw 6-8 lo.x lo.y lo.z, dw 9-11 hi.x hi.y hi.z: one byte per slot (slot s = byte s); child box = [origin + lo * scale,
    //          origin + hi * scale], rounded OUTWARDS around the padded box of nodes4 (checked in exact arithmetic)
    //   dw 12-15 child refs; an unused slot holds an inverted box (lo = 255, hi = 0) and the reference of an empty leaf
    // Empty (and nodes4 used instead) if a box cannot be quantised conservatively.
    std::vector<uint32_t> nodes4q;
    // The same nodes with HALF-FLOAT child boxes, 20 dwords (80 B) per node (trace mode 6, MSK_QUANT_BVH=2; round 5):
    //   dw 0-2 origin (low corner of the node's box), dw 3 scale (ONE power of two: the largest extent / scale lies in [1024, 2048))
    //   dw 4-5 lo.x, dw 6-7 lo.y, dw 8-9 lo.z, dw 10-11 hi.x, dw 12-13 hi.y, dw 14-15 hi.z: four fp16 each (slot s = half s),
    //          child box = [origin + lo * scale, origin + hi * scale], rounded OUTWARDS around the padded box of nodes4 (lo down,
    //          hi up and never a subnormal; checked in exact arithmetic); dw 16-19 child refs — a leaf's as everywhere, an inner
    //          child's as the BYTE offset of its node (index x 80); an unused slot holds lo = 65504, hi = 0 and the reference of an
    //          empty leaf.  Empty if a box cannot be represented (or the tree has 2^31 / 80 nodes or more).
    std::vector<uint32_t> nodes4h;
    // 8-wide form with QUANTISED child boxes (collapse8): 32 dwords (128 B = one L2 line) per node,
    //   dw 0-2  origin (the low corner of the node's box)      dw 3  meta: bits 0-1 ordering axis, bits 8-15 mask of used slots
    //   dw 4-6  scale (a power of two per axis)                dw 7  -
    //   dw 8-19 child boxes as bytes, slot s = byte s of a dword pair: lo.x[8] lo.y[8] lo.z[8] hi.x[8] hi.y[8] hi.z[8]
    //           child box = [origin + lo * scale, origin + hi * scale], rounded OUTWARDS (it contains the padded box above)
    //   dw 20-27 child refs (as above), kEmpty4 for an unused slot (whose box is inverted: lo = 255, hi = 0)
    // Slots are ordered by the children's centres along the ordering axis, so a ray visits them front to back by walking the
    // slots up or down according to the sign of its direction on that axis — no sorting of eight distances per visit.
    std::vector<float> nodes8;
    uint32_t root_ref8 = 0;
    int max_depth8 = 0;
};
static constexpr uint32_t kEmpty4 = 0xfffffffeu;

struct Builder {
    const float *pos;           // 9 floats per triangle: p0 p1 p2 (scene-global order)
    uint32_t n;
    std::vector<Box> tb;
    std::vector<V3> tc;
    std::vector<uint32_t> order;
    std::vector<uint32_t> leaf_order;
    float pad = 0;
    static constexpr int kBins = 16;
    int kLeaf = 2;              // max triangles per leaf (MSK_BVH_LEAF overrides for experiments)
    int kSweepLevels = 8;       // the first kSweepLevels levels evaluate EVERY centroid split of the three axes (a full SAH sweep over sorted
                                // centroids) instead of 16 bins over the centroids' range — which the room's few huge triangles span, so
                                // that a 146 k-triangle mesh in the middle of it sees three or four of the bins (MSK_BVH_SWEEP overrides;
                                // round 5: node visits per ray 12.9 -> 10.7 on the config-5-class scene, 8.4 -> 6.7 on config 3's)
    std::vector<uint32_t> sweep_idx; std::vector<float> sweep_area;
    struct Child { int ref; int count; Box box; };
    std::vector<float> nodes;
    int max_depth = 0;

    Box range_box(uint32_t first, uint32_t count) const {
        Box b; for (uint32_t i = first; i < first + count; ++i) b.grow(tb[order[i]]); return b;
    }
    Child make_leaf(uint32_t first, uint32_t count) {
        Child c; c.ref = (int) leaf_order.size(); c.count = (int) count; c.box = range_box(first, count);
        for (uint32_t i = first; i < first + count; ++i) leaf_order.push_back(order[i]);
        return c;
    }
    Child build(uint32_t first, uint32_t count, int depth) {
        max_depth = std::max(max_depth, depth);
        if (count <= (uint32_t) kLeaf) return make_leaf(first, count);
        Box cb; for (uint32_t i = first; i < first + count; ++i) { V3 c = tc[order[i]]; cb.grow(c); }
        Box nb = range_box(first, count);
        int best_axis = -1, best_bin = -1; float best_cost = INFINITY;
        for (int a = 0; a < 3; ++a) {
            float lo = axis_of(cb.lo, a), hi = axis_of(cb.hi, a);
            if (!(hi > lo)) continue;
            Box bins[kBins]; uint32_t cnt[kBins] = {0};
            float sc = kBins / (hi - lo);
            for (uint32_t i = first; i < first + count; ++i) {
                int b = std::min(kBins - 1, std::max(0, (int) ((axis_of(tc[order[i]], a) - lo) * sc)));
                bins[b].grow(tb[order[i]]); cnt[b]++;
            }
            float right_area[kBins]; uint32_t right_cnt[kBins];
            Box acc; uint32_t c = 0;
            for (int b = kBins - 1; b > 0; --b) { acc.grow(bins[b]); c += cnt[b]; right_area[b] = acc.area(); right_cnt[b] = c; }
            acc = Box(); c = 0;
            for (int b = 0; b < kBins - 1; ++b) {
                acc.grow(bins[b]); c += cnt[b];
                if (c == 0 || right_cnt[b + 1] == 0) continue;
                float cost = acc.area() * c + right_area[b + 1] * right_cnt[b + 1];
                if (cost < best_cost) { best_cost = cost; best_axis = a; best_bin = b; }
            }
        }
        uint32_t mid;
        if (depth < kSweepLevels && sweep_split(first, count, &mid)) {
            // order[first .. first + count) is now sorted along the chosen axis and cut at mid
        } else if (best_axis < 0) {
            mid = first + count / 2;   // all centroids coincide: split in the middle
        } else {
            float lo = axis_of(cb.lo, best_axis), hi = axis_of(cb.hi, best_axis), sc = kBins / (hi - lo);
            auto it = std::partition(order.begin() + first, order.begin() + first + count, [&](uint32_t t) {
                int b = std::min(kBins - 1, std::max(0, (int) ((axis_of(tc[t], best_axis) - lo) * sc)));
                return b <= best_bin;
            });
            mid = (uint32_t) (it - order.begin());
            if (mid == first || mid == first + count) mid = first + count / 2;
        }
        (void) nb;
        int me = (int) (nodes.size() / 16);
        nodes.resize(nodes.size() + 16, 0.f);
        Child l = build(first, mid - first, depth + 1);
        Child r = build(mid, first + count - mid, depth + 1);
        write_node(me, l, r);
        Child c; c.ref = me; c.count = 0; c.box = l.box; c.box.grow(r.box);
        return c;
    }
    // exact SAH over all count - 1 centroid-order splits of the three axes; leaves order[] sorted along the best axis
    bool sweep_split(uint32_t first, uint32_t count, uint32_t *mid) {
        int best_axis = -1; uint32_t best_k = 0; float best_cost = INFINITY;
        sweep_idx.resize(count); sweep_area.resize(count);
        for (int a = 0; a < 3; ++a) {
            std::copy(order.begin() + first, order.begin() + first + count, sweep_idx.begin());
            std::stable_sort(sweep_idx.begin(), sweep_idx.end(), [&](uint32_t x, uint32_t y) { return axis_of(tc[x], a) < axis_of(tc[y], a); });
            Box acc;
            for (uint32_t i = count; i-- > 1;) { acc.grow(tb[sweep_idx[i]]); sweep_area[i] = acc.area(); }      // right part [i, count)
            acc = Box();
            for (uint32_t k = 1; k < count; ++k) {                                                                // left part [0, k)
                acc.grow(tb[sweep_idx[k - 1]]);
                const float cost = acc.area() * k + sweep_area[k] * (count - k);
                if (cost < best_cost) { best_cost = cost; best_axis = a; best_k = k; }
            }
        }
        if (best_axis < 0) return false;
        std::stable_sort(order.begin() + first, order.begin() + first + count, [&](uint32_t x, uint32_t y) { return axis_of(tc[x], best_axis) < axis_of(tc[y], best_axis); });
        *mid = first + best_k;
        return true;
    }
    static uint32_t pack(const Child &c) {
        return c.count > 0 ? (0x80000000u | ((uint32_t) c.ref << 5) | (uint32_t) c.count) : (uint32_t) c.ref;
    }
    void write_node(int idx, const Child &l, const Child &r) {
        float *n = &nodes[(size_t) idx * 16];
        Box a = l.box, b = r.box;
        a.lo = {a.lo.x - pad, a.lo.y - pad, a.lo.z - pad}; a.hi = {a.hi.x + pad, a.hi.y + pad, a.hi.z + pad};
        b.lo = {b.lo.x - pad, b.lo.y - pad, b.lo.z - pad}; b.hi = {b.hi.x + pad, b.hi.y + pad, b.hi.z + pad};
        n[0] = a.lo.x; n[1] = b.lo.x; n[2] = a.lo.y; n[3] = b.lo.y;
        n[4] = a.lo.z; n[5] = b.lo.z; n[6] = a.hi.x; n[7] = b.hi.x;
        n[8] = a.hi.y; n[9] = b.hi.y; n[10] = a.hi.z; n[11] = b.hi.z;
        uint32_t meta[4] = {pack(l), pack(r), 0u, 0u};
        std::memcpy(&n[12], meta, 16);
    }
};

// pos: 9 floats per triangle.  fp32 edge/normal precomputation uses one rounded operation per
// arithmetic op (this TU is compiled with -ffp-contract=off), the same values a per-ray
// evaluation of e1, e2, Ng would produce.
static inline Built build(const float *pos, uint32_t n, float tri_pad) {
    Built out;
    Builder b; b.pos = pos; b.n = n;
    if (const char *e = getenv("MSK_BVH_LEAF")) b.kLeaf = std::max(1, std::min(8, atoi(e)));
    if (const char *e = getenv("MSK_BVH_SWEEP")) b.kSweepLevels = std::max(0, std::min(64, atoi(e)));
    b.tb.resize(n); b.tc.resize(n); b.order.resize(n);
    Box all;
    for (uint32_t i = 0; i < n; ++i) {
        const float *p = pos + (size_t) i * 9;
        Box bx; bx.grow(V3{p[0], p[1], p[2]}); bx.grow(V3{p[3], p[4], p[5]}); bx.grow(V3{p[6], p[7], p[8]});
        b.tb[i] = bx; b.tc[i] = {0.5f * (bx.lo.x + bx.hi.x), 0.5f * (bx.lo.y + bx.hi.y), 0.5f * (bx.lo.z + bx.hi.z)};
        b.order[i] = i; all.grow(bx);
    }
    b.pad = n ? 2.f * tri_pad : 0.f;
    if (n == 0) return out;
    Builder::Child root = b.build(0, n, 0);
    out.root_ref = Builder::pack(root);
    out.nodes = std::move(b.nodes);
    out.max_depth = b.max_depth;
    out.tris.resize((size_t) n * 16); out.bounds.resize((size_t) n * 8);
    for (uint32_t k = 0; k < n; ++k) {
        uint32_t prim = b.leaf_order[k];
        const float *p = pos + (size_t) prim * 9;
        float *t = &out.tris[(size_t) k * 16];
        float *bb = &out.bounds[(size_t) k * 8];
        float e1[3] = {p[0] - p[3], p[1] - p[4], p[2] - p[5]};        // v0 - v1
        float e2[3] = {p[6] - p[0], p[7] - p[1], p[8] - p[2]};        // v2 - v0
        float ng[3] = {e2[1] * e1[2] - e2[2] * e1[1], e2[2] * e1[0] - e2[0] * e1[2], e2[0] * e1[1] - e2[1] * e1[0]};
        t[0] = p[0]; t[1] = p[1]; t[2] = p[2]; std::memcpy(&t[3], &prim, 4);
        t[4] = e1[0]; t[5] = e1[1]; t[6] = e1[2]; t[7] = 0;
        t[8] = e2[0]; t[9] = e2[1]; t[10] = e2[2]; t[11] = 0;
        t[12] = ng[0]; t[13] = ng[1]; t[14] = ng[2]; t[15] = 0;
        for (int a = 0; a < 3; ++a) {                     // oracle D10, same fp32 operations as oracle.cpp's intersect_triangle
            const float w0 = p[a], w1 = p[a] - e1[a], w2 = p[a] + e2[a];
            bb[a] = std::min(w0, std::min(w1, w2)) - tri_pad;
            bb[4 + a] = std::max(w0, std::max(w1, w2)) + tri_pad;
        }
        bb[3] = 0; bb[7] = 0;
    }
    return out;
}

// Collapses the binary tree into 4-wide nodes: a node's two children are replaced by their own children, largest
// surface area first, until four slots are filled or only leaves remain.
struct Collapser {
    const std::vector<float> &n2;
    std::vector<float> out;
    int max_depth = 0;
    std::vector<uint32_t> out_q;        // the quantised twin (Built::nodes4q)
    bool ok_q = true;
    std::vector<uint32_t> out_h;        // the half-float twin (Built::nodes4h)
    bool ok_h = true;
    // v >= 0 as fp16 bits, rounded down or up (never to a subnormal when rounding up: a flushed subnormal would shrink the box)
    static uint16_t to_half(double v, bool up) {
        if (!(v > 0.0)) return 0;
        if (v >= 65504.0) return up && v > 65504.0 ? 0x7c00 : 0x7bff;
        int e; const double m = std::frexp(v, &e);                       // v = m 2^e, m in [0.5, 1)
        int he = e - 1 + 15;                                             // biased exponent of 1.f x 2^(e-1)
        if (he <= 0) {                                                   // below the smallest normal (2^-14)
            if (up) return 0x0400;
            const double q = std::floor(v * 16777216.0);                 // subnormal: q 2^-24
            return (uint16_t) q;
        }
        const double f = (m * 2.0 - 1.0) * 1024.0;                       // 10 fraction bits
        double q = up ? std::ceil(f) : std::floor(f);
        uint32_t h = ((uint32_t) he << 10) + (uint32_t) q;               // q = 1024 carries into the exponent
        return (uint16_t) h;
    }
    static double from_half(uint16_t h) {
        const int e = (h >> 10) & 31; const int f = h & 1023;
        if (e == 31) return INFINITY;
        return e ? std::ldexp(1.0 + f / 1024.0, e - 15) : std::ldexp((double) f, -24);
    }
    struct Slot { uint32_t ref; float lo[3], hi[3]; };
    void quantise(uint32_t me, const Slot *s, int ns, const uint32_t *refs);
    void quantise_h(uint32_t me, const Slot *s, int ns, const uint32_t *refs);
    static float area(const Slot &s) {
        float dx = s.hi[0] - s.lo[0], dy = s.hi[1] - s.lo[1], dz = s.hi[2] - s.lo[2];
        return 2.f * (dx * dy + dy * dz + dz * dx);
    }
    void children(uint32_t node, Slot *a, Slot *b) const {
        const float *n = &n2[(size_t) node * 16];
        uint32_t meta[4]; std::memcpy(meta, &n[12], 16);
        a->ref = meta[0]; b->ref = meta[1];
        a->lo[0] = n[0]; a->lo[1] = n[2]; a->lo[2] = n[4]; a->hi[0] = n[6]; a->hi[1] = n[8]; a->hi[2] = n[10];
        b->lo[0] = n[1]; b->lo[1] = n[3]; b->lo[2] = n[5]; b->hi[0] = n[7]; b->hi[1] = n[9]; b->hi[2] = n[11];
    }
    // Which descendants of a binary node become the (up to four) children of its wide node.  Greedy (dp empty): open the child
    // with the largest surface area until four slots are filled.  Optimal (plan() was run): the choice that minimises the summed
    // surface area of the wide nodes — a visit tests four boxes, used or not, so a wide node costs its area whatever it holds,
    // and the leaves are the binary tree's either way — by dynamic programming over the binary tree (the scheme of Ylitie,
    // Karras, Laine 2017 for 8-wide trees): cost[i - 1] = the cheapest way to hang a subtree into at most i slots.  Greedy
    // fills 2.97 of 4 slots on the 146 k-triangle mesh, the plan 3.54: 23 % fewer nodes, 5.5 % fewer node visits per ray
    // (tools/micro/bvh_stats.cpp).
    struct Plan { float cost[4]; uint8_t split[4]; uint8_t slots[4]; };    // index i - 1 for "at most i slots": slots = 0: n is one wide node (whose own
                                                                           // four slots are split[0] : 4 - split[0]); else n's children share `slots` slots, split : slots - split
    std::vector<Plan> dp;
    void plan(uint32_t root) {
        const size_t nn = n2.size() / 16;
        dp.assign(nn, Plan{});
        auto cost_of = [&](const Slot &c, int i) {                        // T[c][i] of a child reference
            if (c.ref & 0x80000000u) return 0.f;
            return dp[c.ref].cost[i - 1];
        };
        // children before their parent: a post-order walk from the root (the host builder numbers children above their parent,
        // the device builder — msk_lbvh.hip, Karras' internal-node indices — does not: an index sweep would read unplanned children)
        std::vector<uint32_t> order, todo;
        order.reserve(nn);
        if (!(root & 0x80000000u)) todo.push_back(root);
        while (!todo.empty()) {
            const uint32_t n = todo.back(); todo.pop_back();
            order.push_back(n);
            Slot l, r; children(n, &l, &r);
            if (!(l.ref & 0x80000000u)) todo.push_back(l.ref);
            if (!(r.ref & 0x80000000u)) todo.push_back(r.ref);
        }
        for (size_t k = order.size(); k-- > 0;) {                          // reversed pre-order: every node after its descendants
            const uint32_t n = order[k];
            Slot l, r; children(n, &l, &r);
            Plan &p = dp[n];
            float lo[3], hi[3];
            for (int a = 0; a < 3; ++a) { lo[a] = std::min(l.lo[a], r.lo[a]); hi[a] = std::max(l.hi[a], r.hi[a]); }
            Slot me{0, {lo[0], lo[1], lo[2]}, {hi[0], hi[1], hi[2]}};
            float dist[5]; uint8_t dsplit[5] = {0, 0, 0, 0, 0};           // dist[j]: n's two children hung into at most j slots (j >= 2)
            for (int j = 2; j <= 4; ++j) {
                dist[j] = INFINITY;
                for (int k2 = 1; k2 < j; ++k2) {
                    const float c = cost_of(l, k2) + cost_of(r, j - k2);
                    if (c < dist[j]) { dist[j] = c; dsplit[j] = (uint8_t) k2; }
                }
            }
            p.cost[0] = area(me) + dist[4]; p.split[0] = dsplit[4]; p.slots[0] = 0;     // one slot: n is a wide node
            for (int i = 2; i <= 4; ++i) {
                if (dist[i] < p.cost[i - 2]) { p.cost[i - 1] = dist[i]; p.split[i - 1] = dsplit[i]; p.slots[i - 1] = (uint8_t) i; }
                else { p.cost[i - 1] = p.cost[i - 2]; p.split[i - 1] = p.split[i - 2]; p.slots[i - 1] = p.slots[i - 2]; }
            }
        }
    }
    void hang(const Slot &c, int i, Slot *s, int &ns) const {             // c's subtree into at most i slots, as planned
        if ((c.ref & 0x80000000u) || dp[c.ref].slots[i - 1] == 0) { s[ns++] = c; return; }
        Slot l, r; children(c.ref, &l, &r);
        const int j = dp[c.ref].slots[i - 1], k = dp[c.ref].split[i - 1];
        hang(l, k, s, ns); hang(r, j - k, s, ns);
    }
    uint32_t collapse(uint32_t node, int depth) {
        max_depth = std::max(max_depth, depth);
        Slot s[4]; int ns = 2;
        children(node, &s[0], &s[1]);
        if (!dp.empty()) {
            const int k = dp[node].split[0];
            const Slot l = s[0], r = s[1];
            ns = 0;
            hang(l, k, s, ns); hang(r, 4 - k, s, ns);
        } else
        while (ns < 4) {
            int best = -1; float best_area = -1.f;
            for (int i = 0; i < ns; ++i)
                if (!(s[i].ref & 0x80000000u) && area(s[i]) > best_area) { best = i; best_area = area(s[i]); }
            if (best < 0) break;
            Slot a, b; children(s[best].ref, &a, &b);
            s[best] = a; s[ns++] = b;
        }
        const uint32_t me = (uint32_t) (out.size() / 32);
        out.resize(out.size() + 32, 0.f);
        uint32_t refs[4];
        for (int i = 0; i < 4; ++i) {
            if (i >= ns) refs[i] = kEmpty4;
            else if (s[i].ref & 0x80000000u) refs[i] = s[i].ref;
            else refs[i] = collapse(s[i].ref, depth + 1);
        }
        float *o = &out[(size_t) me * 32];
        for (int i = 0; i < 4; ++i)
            for (int a = 0; a < 3; ++a) {
                // an unused slot holds an inverted box: the slab test misses it like any other box (node4_step has no other test)
                o[a * 4 + i] = i < ns ? s[i].lo[a] : 3e38f;
                o[12 + a * 4 + i] = i < ns ? s[i].hi[a] : -3e38f;
            }
        std::memcpy(&o[24], refs, 16);
        quantise(me, s, ns, refs);
        quantise_h(me, s, ns, refs);
        return me;
    }
};
inline void Collapser::quantise(uint32_t me, const Slot *s, int ns, const uint32_t *refs) {
    if (out_q.size() < ((size_t) me + 1) * 16) out_q.resize(((size_t) me + 1) * 16, 0u);
    uint32_t *q = &out_q[(size_t) me * 16];
    float origin[3], scale[3];
    uint32_t lo_b[3] = {0, 0, 0}, hi_b[3] = {0, 0, 0};
    for (int a = 0; a < 3; ++a) {
        double lo = INFINITY, hi = -INFINITY;
        for (int i = 0; i < ns; ++i) { lo = std::min(lo, (double) s[i].lo[a]); hi = std::max(hi, (double) s[i].hi[a]); }
        origin[a] = (float) lo;
        const double ext = hi - (double) origin[a];
        // the finest grid that still spans the node: scale = ext / 255 rounded up to a float (a power of two would be up to twice
        // as coarse and cost ~10 % more leaf visits); a degenerate axis gets a tiny positive scale
        float sf = ext > 0 ? (float) (ext / 255.0) : 1e-30f;
        if (!(sf > 0.f) || !std::isfinite(sf)) { ok_q = false; sf = 1.f; }
        while (255.0 * (double) sf < ext) sf = std::nextafterf(sf, INFINITY);
        scale[a] = sf;
        const double sc = (double) sf;
        for (int i = 0; i < 4; ++i) {
            if (i >= ns) { lo_b[a] |= 255u << (8 * i); continue; }                 // inverted: lo = 255, hi = 0
            double ql = std::floor(((double) s[i].lo[a] - (double) origin[a]) / sc), qh = std::ceil(((double) s[i].hi[a] - (double) origin[a]) / sc);
            ql = std::max(0.0, std::min(255.0, ql)); qh = std::max(0.0, std::min(255.0, qh));
            // exact in double: the decoded box must contain the child's
            if (!((double) origin[a] + ql * sc <= (double) s[i].lo[a] && (double) origin[a] + qh * sc >= (double) s[i].hi[a])) ok_q = false;
            lo_b[a] |= (uint32_t) ql << (8 * i); hi_b[a] |= (uint32_t) qh << (8 * i);
        }
    }
    std::memcpy(&q[0], origin, 12); std::memcpy(&q[3], &scale[0], 4); std::memcpy(&q[4], &scale[1], 8);
    for (int a = 0; a < 3; ++a) { q[6 + a] = lo_b[a]; q[9 + a] = hi_b[a]; }
    for (int i = 0; i < 4; ++i) q[12 + i] = refs[i] == kEmpty4 ? 0x80000000u : refs[i];       // empty leaf: first 0, count 0
}
inline void Collapser::quantise_h(uint32_t me, const Slot *s, int ns, const uint32_t *refs) {
    if (out_h.size() < ((size_t) me + 1) * 20) out_h.resize(((size_t) me + 1) * 20, 0u);
    uint32_t *q = &out_h[(size_t) me * 20];
    float origin[3]; double ext = 0;
    for (int a = 0; a < 3; ++a) {
        double lo = INFINITY, hi = -INFINITY;
        for (int i = 0; i < ns; ++i) { lo = std::min(lo, (double) s[i].lo[a]); hi = std::max(hi, (double) s[i].hi[a]); }
        origin[a] = (float) lo;
        ext = std::max(ext, hi - (double) origin[a]);
    }
    int e = 0;
    if (ext > 0) { (void) std::frexp(ext, &e); e -= 11; }               // ext = m 2^e', m in [0.5, 1): ext / 2^(e' - 11) in [1024, 2048)
    if (e < -120 || e > 120 || !std::isfinite(ext)) { ok_h = false; e = 0; }
    const float scale = std::ldexp(1.f, e);
    const double sc = (double) scale;
    uint16_t h[6][4];
    for (int a = 0; a < 3; ++a)
        for (int i = 0; i < 4; ++i) {
            if (i >= ns) { h[a][i] = 0x7bff; h[3 + a][i] = 0; continue; }                // inverted: lo = 65504, hi = 0
            const double lo = ((double) s[i].lo[a] - (double) origin[a]) / sc, hi = ((double) s[i].hi[a] - (double) origin[a]) / sc;
            h[a][i] = to_half(lo, false); h[3 + a][i] = to_half(hi, true);
            // exact in double (a power-of-two scale): the decoded box must contain the child's
            if (!((double) origin[a] + from_half(h[a][i]) * sc <= (double) s[i].lo[a] && (double) origin[a] + from_half(h[3 + a][i]) * sc >= (double) s[i].hi[a]) ||
                h[3 + a][i] >= 0x7c00) ok_h = false;
        }
    std::memcpy(&q[0], origin, 12); std::memcpy(&q[3], &scale, 4);
    for (int p = 0; p < 6; ++p) { q[4 + 2 * p] = (uint32_t) h[p][0] | ((uint32_t) h[p][1] << 16); q[5 + 2 * p] = (uint32_t) h[p][2] | ((uint32_t) h[p][3] << 16); }
    // an inner child's reference is the BYTE offset of its node (index x 80: the visit's address without a multiplication);
    // collapse4 drops the half-float twin for a tree of 2^31 / 80 nodes or more
    for (int i = 0; i < 4; ++i) q[16 + i] = refs[i] == kEmpty4 ? 0x80000000u : (refs[i] & 0x80000000u) ? refs[i] : refs[i] * 80u;
}
static inline void collapse4(Built &b, bool optimal = true) {
    b.nodes4.clear(); b.nodes4q.clear(); b.nodes4h.clear(); b.root_ref4 = b.root_ref; b.max_depth4 = 0;
    if (b.root_ref & 0x80000000u) return;            // a single leaf: nothing to collapse
    Collapser c{b.nodes, {}, 0};
    if (optimal) c.plan(b.root_ref);
    b.root_ref4 = c.collapse(b.root_ref, 1);
    b.nodes4 = std::move(c.out);
    if (c.ok_q) b.nodes4q = std::move(c.out_q);
    if (c.ok_h && c.out_h.size() / 20 < (1u << 31) / 80u) b.nodes4h = std::move(c.out_h);
    b.max_depth4 = c.max_depth;
}

// Collapses the binary tree into 8-wide nodes with quantised child boxes (layout: Built::nodes8).
struct Collapser8 {
    const std::vector<float> &n2;
    std::vector<float> out;
    int max_depth = 0;
    bool ok = true;                      // false: a quantised box failed its containment check (extents beyond the exponent clamp)
    typedef Collapser::Slot Slot;
    void children(uint32_t node, Slot *a, Slot *b) const { Collapser c{n2, {}, 0}; c.children(node, a, b); }
    uint32_t collapse(uint32_t node, int depth) {
        max_depth = std::max(max_depth, depth);
        Slot s[8]; int ns = 2;
        children(node, &s[0], &s[1]);
        while (ns < 8) {                 // open the inner child with the largest surface area
            int best = -1; float best_area = -1.f;
            for (int i = 0; i < ns; ++i)
                if (!(s[i].ref & 0x80000000u) && Collapser::area(s[i]) > best_area) { best = i; best_area = Collapser::area(s[i]); }
            if (best < 0) break;
            Slot a, b; children(s[best].ref, &a, &b);
            s[best] = a; s[ns++] = b;
        }
        // ordering axis: the one along which the children's centres spread most; slots ascending along it
        float clo[3] = {INFINITY, INFINITY, INFINITY}, chi[3] = {-INFINITY, -INFINITY, -INFINITY};
        double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
        for (int i = 0; i < ns; ++i)
            for (int a = 0; a < 3; ++a) {
                const float c = 0.5f * (s[i].lo[a] + s[i].hi[a]);
                clo[a] = std::min(clo[a], c); chi[a] = std::max(chi[a], c);
                lo[a] = std::min(lo[a], (double) s[i].lo[a]); hi[a] = std::max(hi[a], (double) s[i].hi[a]);
            }
        int axis = 0;
        for (int a = 1; a < 3; ++a) if (chi[a] - clo[a] > chi[axis] - clo[axis]) axis = a;
        std::stable_sort(s, s + ns, [&](const Slot &x, const Slot &y) { return x.lo[axis] + x.hi[axis] < y.lo[axis] + y.hi[axis]; });
        const uint32_t me = (uint32_t) (out.size() / 32);
        out.resize(out.size() + 32, 0.f);
        uint32_t refs[8];
        for (int i = 0; i < 8; ++i) {
            if (i >= ns) refs[i] = kEmpty4;
            else if (s[i].ref & 0x80000000u) refs[i] = s[i].ref;
            else refs[i] = collapse(s[i].ref, depth + 1);
        }
        // quantisation frame: origin = low corner (a float), scale = the power of two with 255 * scale >= extent
        float origin[3], scale[3];
        uint8_t q[6][8];
        for (int a = 0; a < 3; ++a) {
            origin[a] = (float) lo[a];
            const double ext = hi[a] - (double) origin[a];
            int e = ext > 0 ? (int) std::ceil(std::log2(ext / 255.0)) : -60;
            e = std::max(-100, std::min(100, e));
            while (255.0 * std::ldexp(1.0, e) < ext) ++e;              // (log2 rounding)
            scale[a] = (float) std::ldexp(1.0, e);
            const double sc = std::ldexp(1.0, e);
            for (int i = 0; i < 8; ++i) {
                if (i >= ns) { q[a][i] = 255; q[3 + a][i] = 0; continue; }
                double ql = std::floor(((double) s[i].lo[a] - (double) origin[a]) / sc), qh = std::ceil(((double) s[i].hi[a] - (double) origin[a]) / sc);
                ql = std::max(0.0, std::min(255.0, ql)); qh = std::max(0.0, std::min(255.0, qh));
                // exact in double: origin + q * 2^e; the decoded box must contain the child's
                if ((double) origin[a] + ql * sc > (double) s[i].lo[a] || (double) origin[a] + qh * sc < (double) s[i].hi[a]) ok = false;
                q[a][i] = (uint8_t) ql; q[3 + a][i] = (uint8_t) qh;
            }
        }
        float *o = &out[(size_t) me * 32];
        const uint32_t meta = (uint32_t) axis | (((1u << ns) - 1u) << 8);
        std::memcpy(&o[0], origin, 12); std::memcpy(&o[3], &meta, 4);
        std::memcpy(&o[4], scale, 12);
        std::memcpy(&o[8], q, 48);
        std::memcpy(&o[20], refs, 32);
        return me;
    }
};
// false (and no 8-wide tree) when a box cannot be quantised conservatively: the caller keeps the full-precision 4-wide tree
static inline bool collapse8(Built &b) {
    b.nodes8.clear(); b.root_ref8 = b.root_ref; b.max_depth8 = 0;
    if (b.root_ref & 0x80000000u) return true;
    Collapser8 c{b.nodes, {}, 0, true};
    b.root_ref8 = c.collapse(b.root_ref, 1);
    if (!c.ok) { b.root_ref8 = b.root_ref; return false; }
    b.nodes8 = std::move(c.out);
    b.max_depth8 = c.max_depth;
    return true;
}

}  // namespace mskbvh
