// Host-side traversal statistics of the 4-wide quantised tree (msk_bvh.h), used to plan the layout the traversal kernel reads:
//   g++ -O2 -std=c++17 -ffp-contract=off -I misaki-render_amd/csrc tools/micro/bvh_stats.cpp -o gpurun_scratch/bvh_stats
//   gpurun_scratch/bvh_stats positions.bin [n_rays]
// positions.bin: 9 floats per triangle (tools/dump_positions.py).  Rays: area-weighted surface points, cosine-distributed
// directions (the bounce rays of a path tracer), closest-hit traversal with the kernel's visiting order.
// Prints: node visits / leaf visits / triangle tests per ray, the share of node visits that fall on the first N nodes in
// breadth-first order (what an LDS-resident treetop of N nodes would serve), and a 4 MiB 16-way LRU model of one XCD's L2
// over the 64-byte node and 48-byte triangle records for the depth-first layout.
#include "msk_bvh.h"
#include <cstdio>
#include <random>
#include <queue>
#include <map>

using namespace mskbvh;

struct Ray { float o[3], d[3], tmin; };

static bool tri_hit(const float *p, const Ray &r, float tmax, float *t_out) {
    // plain Moeller-Trumbore (statistics only: not the bit-exact test)
    float e1[3] = {p[3] - p[0], p[4] - p[1], p[5] - p[2]}, e2[3] = {p[6] - p[0], p[7] - p[1], p[8] - p[2]};
    float pv[3] = {r.d[1] * e2[2] - r.d[2] * e2[1], r.d[2] * e2[0] - r.d[0] * e2[2], r.d[0] * e2[1] - r.d[1] * e2[0]};
    float det = e1[0] * pv[0] + e1[1] * pv[1] + e1[2] * pv[2];
    if (std::fabs(det) < 1e-12f) return false;
    float inv = 1.f / det;
    float tv[3] = {r.o[0] - p[0], r.o[1] - p[1], r.o[2] - p[2]};
    float u = (tv[0] * pv[0] + tv[1] * pv[1] + tv[2] * pv[2]) * inv;
    if (u < 0 || u > 1) return false;
    float qv[3] = {tv[1] * e1[2] - tv[2] * e1[1], tv[2] * e1[0] - tv[0] * e1[2], tv[0] * e1[1] - tv[1] * e1[0]};
    float v = (r.d[0] * qv[0] + r.d[1] * qv[1] + r.d[2] * qv[2]) * inv;
    if (v < 0 || u + v > 1) return false;
    float t = (e2[0] * qv[0] + e2[1] * qv[1] + e2[2] * qv[2]) * inv;
    if (t <= r.tmin || t > tmax) return false;
    *t_out = t;
    return true;
}

struct Lru {       // set-associative LRU over 128-byte lines
    size_t sets, ways; std::vector<uint64_t> tag; std::vector<uint32_t> age; uint32_t clock = 0; uint64_t hits = 0, misses = 0;
    Lru(size_t bytes, size_t w) : sets(bytes / 128 / w), ways(w), tag(sets * w, ~0ull), age(sets * w, 0) {}
    void touch(uint64_t addr) {
        const uint64_t line = addr >> 7; const size_t s = (size_t) (line % sets);
        ++clock;
        size_t victim = 0; uint32_t oldest = ~0u;
        for (size_t w = 0; w < ways; ++w) {
            if (tag[s * ways + w] == line) { age[s * ways + w] = clock; ++hits; return; }
            if (age[s * ways + w] < oldest) { oldest = age[s * ways + w]; victim = w; }
        }
        ++misses; tag[s * ways + victim] = line; age[s * ways + victim] = clock;
    }
};

int main(int argc, char **argv) {
    if (argc < 2) { std::fprintf(stderr, "usage: bvh_stats positions.bin [n_rays]\n"); return 2; }
    FILE *f = std::fopen(argv[1], "rb");
    if (!f) { std::perror(argv[1]); return 2; }
    std::fseek(f, 0, SEEK_END); const long bytes = std::ftell(f); std::fseek(f, 0, SEEK_SET);
    const uint32_t n = (uint32_t) (bytes / 36);
    std::vector<float> pos((size_t) n * 9);
    if (std::fread(pos.data(), 36, n, f) != n) return 2;
    std::fclose(f);
    const size_t n_rays = argc > 2 ? (size_t) atol(argv[2]) : 2000000;
    Box all; for (uint32_t i = 0; i < n * 3; ++i) all.grow(V3{pos[i * 3], pos[i * 3 + 1], pos[i * 3 + 2]});
    const float diag = std::sqrt((all.hi.x - all.lo.x) * (all.hi.x - all.lo.x) + (all.hi.y - all.lo.y) * (all.hi.y - all.lo.y) + (all.hi.z - all.lo.z) * (all.hi.z - all.lo.z));
    float amax = 0;
    for (float v : {all.lo.x, all.lo.y, all.lo.z, all.hi.x, all.hi.y, all.hi.z}) amax = std::max(amax, std::fabs(v));
    // msk_gpu.hip's rule (0.5e-5 of the scene's scale); PAD_SCALE=1e-4 reproduces rounds 1-3
    // SKIP_FIRST=k: the tree holds the triangles from k on only (the room's 14 come first: what does the mesh alone cost the same rays?)
    const uint32_t skip = getenv("SKIP_FIRST") ? (uint32_t) atoi(getenv("SKIP_FIRST")) : 0u;
    Built b = build(pos.data() + (size_t) skip * 9, n - skip, 0.5f * (getenv("PAD_SCALE") ? (float) atof(getenv("PAD_SCALE")) : 1e-5f) * std::max(diag, amax));
    collapse4(b, !getenv("GREEDY"));
    const uint32_t nn = (uint32_t) (b.nodes4.size() / 32);
    std::printf("%u triangles, %u 4-wide nodes (%.2f MB as 64-byte nodes), depth %d, triangles %.2f MB as 48-byte records\n", n, nn, nn * 64 / 1e6, b.max_depth4, n * 48 / 1e6);
    // breadth-first rank of every node
    std::vector<uint32_t> rank(nn, 0), depth(nn, 0);
    {
        std::queue<uint32_t> q; q.push(b.root_ref4); uint32_t r = 0;
        while (!q.empty()) {
            const uint32_t u = q.front(); q.pop(); rank[u] = r++;
            const uint32_t *refs = (const uint32_t *) &b.nodes4[(size_t) u * 32 + 24];
            for (int i = 0; i < 4; ++i) if (refs[i] != kEmpty4 && !(refs[i] & 0x80000000u)) { depth[refs[i]] = depth[u] + 1; q.push(refs[i]); }
        }
    }
    // triangle k of the leaf order -> scene-global index is in tris[k*16+3]
    std::vector<double> cdf(n);
    double acc = 0;
    for (uint32_t i = 0; i < n; ++i) {
        const float *p = &pos[(size_t) i * 9];
        const float e1[3] = {p[3] - p[0], p[4] - p[1], p[5] - p[2]}, e2[3] = {p[6] - p[0], p[7] - p[1], p[8] - p[2]};
        const float c[3] = {e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0]};
        acc += 0.5 * std::sqrt((double) c[0] * c[0] + (double) c[1] * c[1] + (double) c[2] * c[2]);
        cdf[i] = acc;
    }
    std::mt19937_64 rng(1234);
    std::uniform_real_distribution<double> U(0, 1);
    const bool quant = !b.nodes4q.empty();
    uint64_t node_visits = 0, leaf_visits = 0, tri_tests = 0, hits = 0, stale_nodes = 0, stale_leaves = 0, stale_tris = 0;
    std::vector<uint64_t> by_rank_bucket(32, 0);         // bucket k: rank < 2^k
    std::vector<uint64_t> by_depth(64, 0);
    Lru l2(4u << 20, 16);
    const uint64_t tri_base = (uint64_t) nn * 64 + 4096;
    uint64_t pairs_fan = 0, pairs_other = 0, singles = 0;
    // how many 2-triangle leaves are "fans" (tri1 = (v0, v2 of tri0, new vertex))?
    for (uint32_t u = 0; u < nn; ++u) {
        const uint32_t *refs = (const uint32_t *) &b.nodes4[(size_t) u * 32 + 24];
        for (int i = 0; i < 4; ++i) {
            if (refs[i] == kEmpty4 || !(refs[i] & 0x80000000u)) continue;
            const uint32_t first = (refs[i] & 0x7fffffffu) >> 5, cnt = refs[i] & 31u;
            if (cnt == 1) { ++singles; continue; }
            uint32_t pa, pb; std::memcpy(&pa, &b.tris[(size_t) first * 16 + 3], 4); std::memcpy(&pb, &b.tris[(size_t) (first + 1) * 16 + 3], 4);
            const float *A = &pos[(size_t) (pa + skip) * 9], *B = &pos[(size_t) (pb + skip) * 9];
            auto same = [](const float *x, const float *y) { return x[0] == y[0] && x[1] == y[1] && x[2] == y[2]; };
            int shared = 0;
            for (int x = 0; x < 3; ++x) for (int y = 0; y < 3; ++y) shared += same(A + 3 * x, B + 3 * y);
            if (shared >= 2) ++pairs_fan; else ++pairs_other;
        }
    }
    std::printf("leaves: %llu single, %llu two triangles sharing an edge, %llu two unrelated triangles\n", (unsigned long long) singles, (unsigned long long) pairs_fan, (unsigned long long) pairs_other);
    const float cx = 278, cy = 274, cz = 280;
    for (size_t k = 0; k < n_rays; ++k) {
        const uint32_t t = (uint32_t) (std::lower_bound(cdf.begin(), cdf.end(), U(rng) * acc) - cdf.begin());
        const float *p = &pos[(size_t) std::min(t, n - 1) * 9];
        float u = (float) U(rng), v = (float) U(rng);
        if (u + v > 1) { u = 1 - u; v = 1 - v; }
        Ray r;
        for (int a = 0; a < 3; ++a) r.o[a] = p[a] + u * (p[3 + a] - p[a]) + v * (p[6 + a] - p[a]);
        float e1[3] = {p[3] - p[0], p[4] - p[1], p[5] - p[2]}, e2[3] = {p[6] - p[0], p[7] - p[1], p[8] - p[2]};
        float nrm[3] = {e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0]};
        float len = std::sqrt(nrm[0] * nrm[0] + nrm[1] * nrm[1] + nrm[2] * nrm[2]);
        if (len == 0) continue;
        for (int a = 0; a < 3; ++a) nrm[a] /= len;
        // walls: towards the room; mesh (the last n - 12 triangles): 60 % outwards, 40 % inwards
        const bool wall = t < 12;
        const float toc[3] = {cx - r.o[0], cy - r.o[1], cz - r.o[2]};
        const bool points_in = nrm[0] * toc[0] + nrm[1] * toc[1] + nrm[2] * toc[2] > 0;
        const bool want_in = wall ? true : U(rng) < 0.4;
        if (points_in != want_in) for (int a = 0; a < 3; ++a) nrm[a] = -nrm[a];
        // cosine hemisphere
        const float r1 = (float) U(rng), r2 = (float) U(rng);
        const float sr = std::sqrt(r1), phi = 6.2831853f * r2;
        float lx = sr * std::cos(phi), ly = sr * std::sin(phi), lz = std::sqrt(std::max(0.f, 1 - r1));
        float s[3], tt[3];
        if (std::fabs(nrm[0]) > 0.9f) { s[0] = 0; s[1] = 1; s[2] = 0; } else { s[0] = 1; s[1] = 0; s[2] = 0; }
        float dp = s[0] * nrm[0] + s[1] * nrm[1] + s[2] * nrm[2];
        for (int a = 0; a < 3; ++a) s[a] -= dp * nrm[a];
        len = std::sqrt(s[0] * s[0] + s[1] * s[1] + s[2] * s[2]);
        for (int a = 0; a < 3; ++a) s[a] /= len;
        tt[0] = nrm[1] * s[2] - nrm[2] * s[1]; tt[1] = nrm[2] * s[0] - nrm[0] * s[2]; tt[2] = nrm[0] * s[1] - nrm[1] * s[0];
        for (int a = 0; a < 3; ++a) r.d[a] = lx * s[a] + ly * tt[a] + lz * nrm[a];
        r.tmin = 8.940697e-05f * (1.f + std::max(std::fabs(r.o[0]), std::max(std::fabs(r.o[1]), std::fabs(r.o[2]))));   // interaction.h:40-44
        float idir[3], oi[3];
        for (int a = 0; a < 3; ++a) { idir[a] = std::max(-1e25f, std::min(1e25f, 1.f / r.d[a])); oi[a] = r.o[a] * idir[a]; }
        float best = INFINITY;
        uint32_t stack[128]; float stack_t[128]; int sp = 0; bool popped_stale = false;
        uint32_t cur = b.root_ref4;
        const uint32_t DONE = 0xffffffffu;
        while (cur != DONE) {
            while (!(cur & 0x80000000u)) {
                ++node_visits; if (popped_stale) ++stale_nodes; popped_stale = false;
                for (int kb = 0; kb < 32; ++kb) if (rank[cur] < (1u << kb)) { ++by_rank_bucket[kb]; break; }
                ++by_depth[depth[cur]];
                l2.touch((uint64_t) cur * 64);
                float tn[4]; uint32_t rf[4]; int nh = 0;
                if (quant) {
                    const uint32_t *q = &b.nodes4q[(size_t) cur * 16];
                    float org[3], scl[3]; std::memcpy(org, q, 12); std::memcpy(&scl[0], q + 3, 4); std::memcpy(&scl[1], q + 4, 8);
                    for (int i = 0; i < 4; ++i) {
                        float t0 = r.tmin, t1 = best;
                        for (int a = 0; a < 3; ++a) {
                            const float lo = org[a] + ((q[6 + a] >> (8 * i)) & 255u) * scl[a], hi = org[a] + ((q[9 + a] >> (8 * i)) & 255u) * scl[a];
                            const float ta = lo * idir[a] - oi[a], tb = hi * idir[a] - oi[a];
                            t0 = std::max(t0, std::min(ta, tb)); t1 = std::min(t1, std::max(ta, tb));
                        }
                        if (t0 <= t1 * 1.0000004f && ((q[6] >> (8 * i)) & 255u) <= ((q[9] >> (8 * i)) & 255u)) { tn[nh] = t0; rf[nh] = q[12 + i]; ++nh; }
                    }
                } else {
                    const float *nd = &b.nodes4[(size_t) cur * 32];
                    const uint32_t *refs = (const uint32_t *) (nd + 24);
                    for (int i = 0; i < 4; ++i) {
                        if (refs[i] == kEmpty4) continue;
                        float t0 = r.tmin, t1 = best;
                        for (int a = 0; a < 3; ++a) {
                            const float ta = nd[a * 4 + i] * idir[a] - oi[a], tb = nd[12 + a * 4 + i] * idir[a] - oi[a];
                            t0 = std::max(t0, std::min(ta, tb)); t1 = std::min(t1, std::max(ta, tb));
                        }
                        if (t0 <= t1 * 1.0000004f) { tn[nh] = t0; rf[nh] = refs[i]; ++nh; }
                    }
                }
                for (int i = 1; i < nh; ++i) for (int j = i; j > 0 && tn[j] < tn[j - 1]; --j) { std::swap(tn[j], tn[j - 1]); std::swap(rf[j], rf[j - 1]); }
                if (nh == 0) { if (sp > 0) { cur = stack[--sp]; popped_stale = stack_t[sp] > best; } else cur = DONE; }
                else { for (int i = nh - 1; i >= 1; --i) { stack_t[sp] = tn[i]; stack[sp++] = rf[i]; } cur = rf[0]; }
            }
            if (cur == DONE) break;
            const uint32_t first = (cur & 0x7fffffffu) >> 5, cnt = cur & 31u;
            if (cnt) ++leaf_visits;
            if (popped_stale) { ++stale_leaves; stale_tris += cnt; } popped_stale = false;
            for (uint32_t i = 0; i < cnt; ++i) {
                ++tri_tests;
                l2.touch(tri_base + (uint64_t) (first + i) * 48); l2.touch(tri_base + (uint64_t) (first + i) * 48 + 47);
                uint32_t prim; std::memcpy(&prim, &b.tris[(size_t) (first + i) * 16 + 3], 4);
                float th;
                if (tri_hit(&pos[(size_t) (prim + skip) * 9], r, best, &th) && th < best) best = th;
            }
            if (sp > 0) { cur = stack[--sp]; popped_stale = stack_t[sp] > best; } else cur = DONE;
        }
        hits += best < INFINITY;
    }
    std::printf("%zu rays (%s boxes): %.2f node visits, %.2f leaf visits, %.2f triangle tests per ray; %.1f %% hit\n", n_rays, quant ? "quantised" : "full-precision",
                (double) node_visits / n_rays, (double) leaf_visits / n_rays, (double) tri_tests / n_rays, 100.0 * hits / n_rays);
    std::printf("popped behind the best hit (avoidable with entry distances on the stack): %.2f node visits, %.2f leaf visits, %.2f triangle tests per ray\n",
                (double) stale_nodes / n_rays, (double) stale_leaves / n_rays, (double) stale_tris / n_rays);
    uint64_t cum = 0;
    std::printf("share of node visits served by a breadth-first treetop of N nodes:\n");
    for (int kb = 0; kb < 32 && (1u << kb) <= 2 * nn; ++kb) { cum += by_rank_bucket[kb]; std::printf("  N = %7u (%6.1f KB): %5.1f %%\n", 1u << kb, (1u << kb) * 64 / 1024.0, 100.0 * cum / node_visits); }
    std::printf("by depth:"); for (int dd = 0; dd < 20; ++dd) std::printf(" %.2f", (double) by_depth[dd] / n_rays); std::printf("\n");
    std::printf("4 MiB 16-way LRU over 128-byte lines, depth-first layout: %.1f %% line hits (%.1f line requests per ray)\n", 100.0 * l2.hits / (l2.hits + l2.misses), (double) (l2.hits + l2.misses) / n_rays);
    return 0;
}
