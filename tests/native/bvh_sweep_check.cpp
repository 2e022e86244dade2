// The host BVH builder's full SAH sweep on the first levels (msk_bvh.h: Builder::sweep_split, MSK_BVH_SWEEP), compiled and run by
// tests/test_bvh_host.py: on a room of a few huge triangles around a dense mesh — the case 16 bins over the centroids' range handle
// badly — the swept tree must be a valid tree over the same triangles and cost less (summed surface area of its inner nodes'
// child boxes, the SAH's own measure) than the binned one.  usage: bvh_sweep_check <n_theta>   prints "ok cost_binned C cost_swept C ..."
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <set>
#include <vector>
#include "../../misaki-render_amd/csrc/msk_bvh.h"

static double tree_cost(const mskbvh::Built &b, size_t n_tris, bool *ok) {
    // sum over inner nodes of area(child box) x triangles under that child; every triangle must be in exactly one leaf
    std::vector<int> seen(n_tris, 0);
    double cost = 0;
    struct Item { uint32_t ref; };
    std::vector<uint32_t> todo;
    if (!(b.root_ref & 0x80000000u)) todo.push_back(b.root_ref);
    auto leaf_count = [&](uint32_t ref, auto &&self) -> size_t {
        if (ref & 0x80000000u) {
            const uint32_t first = (ref & 0x7fffffffu) >> 5, cnt = ref & 31u;
            for (uint32_t i = 0; i < cnt; ++i) { uint32_t prim; std::memcpy(&prim, &b.tris[(size_t) (first + i) * 16 + 3], 4); if (prim < n_tris) seen[prim]++; else *ok = false; }
            return cnt;
        }
        const float *n = &b.nodes[(size_t) ref * 16];
        uint32_t meta[4]; std::memcpy(meta, &n[12], 16);
        size_t total = 0;
        for (int c = 0; c < 2; ++c) {
            const double dx = n[6 + c] - n[0 + c], dy = n[8 + c] - n[2 + c], dz = n[10 + c] - n[4 + c];
            const size_t k = self(meta[c], self);
            cost += 2.0 * (dx * dy + dy * dz + dz * dx) * (double) k;
            total += k;
        }
        return total;
    };
    const size_t total = leaf_count(b.root_ref, leaf_count);
    if (total != n_tris) *ok = false;
    for (int s : seen) if (s != 1) *ok = false;
    return cost;
}

int main(int argc, char **argv) {
    const int nt = argc > 1 ? atoi(argv[1]) : 60;
    std::vector<float> pos;
    auto tri = [&](const float *a, const float *b, const float *c) { for (int k = 0; k < 3; ++k) pos.push_back(a[k]); for (int k = 0; k < 3; ++k) pos.push_back(b[k]); for (int k = 0; k < 3; ++k) pos.push_back(c[k]); };
    // the room: five walls of two triangles each, 556 units wide
    const float L = 556.f;
    const float q[5][4][3] = {{{0, 0, 0}, {L, 0, 0}, {L, 0, L}, {0, 0, L}}, {{0, L, 0}, {L, L, 0}, {L, L, L}, {0, L, L}}, {{0, 0, L}, {L, 0, L}, {L, L, L}, {0, L, L}},
                              {{0, 0, 0}, {0, L, 0}, {0, L, L}, {0, 0, L}}, {{L, 0, 0}, {L, L, 0}, {L, L, L}, {L, 0, L}}};
    for (auto &w : q) { tri(w[0], w[1], w[2]); tri(w[0], w[2], w[3]); }
    // a dense sphere of radius 160 in the middle
    const int np = 2 * nt;
    auto pt = [&](int i, int j, float *o) { const double th = M_PI * i / nt, ph = 2 * M_PI * (j % np) / np; o[0] = 278 + 160 * std::sin(th) * std::cos(ph); o[1] = 200 + 160 * std::cos(th); o[2] = 280 + 160 * std::sin(th) * std::sin(ph); };
    for (int i = 0; i < nt; ++i) for (int j = 0; j < np; ++j) { float a[3], b[3], c[3], d[3]; pt(i, j, a); pt(i + 1, j, b); pt(i + 1, j + 1, c); pt(i, j + 1, d); if (i > 0) tri(a, d, c); if (i < nt - 1) tri(a, c, b); }
    const uint32_t n = (uint32_t) (pos.size() / 9);
    setenv("MSK_BVH_SWEEP", "0", 1);
    mskbvh::Built binned = mskbvh::build(pos.data(), n, 0.005f);
    setenv("MSK_BVH_SWEEP", "8", 1);
    mskbvh::Built swept = mskbvh::build(pos.data(), n, 0.005f);
    unsetenv("MSK_BVH_SWEEP");
    mskbvh::Built dflt = mskbvh::build(pos.data(), n, 0.005f);
    bool ok = true;
    const double cb = tree_cost(binned, n, &ok), cs = tree_cost(swept, n, &ok), cd = tree_cost(dflt, n, &ok);
    if (!ok) { printf("FAIL a tree does not hold every triangle exactly once\n"); return 1; }
    if (dflt.nodes != swept.nodes) { printf("FAIL the default is not the sweep on eight levels\n"); return 1; }
    printf("%s triangles %u cost_binned %.6g cost_swept %.6g ratio %.4f depth %d / %d\n", cs < cb ? "ok" : "FAIL", n, cb, cs, cs / cb, binned.max_depth, swept.max_depth);
    (void) cd;
    return cs < cb ? 0 : 1;
}
