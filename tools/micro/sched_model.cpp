// tools/micro/sched_model.cpp — what would another traversal SCHEDULER buy on the mesh configs?  A host-side model (no GPU time)
// that replays the GPU's own rays (MSK_DUMP_RAYS: the live slots of every 32nd region of one wavefront iteration, in the order
// k_trace_r takes them) through the product's tree (msk_bvh.h: the SAH build + the 4-wide collapse the library uploads) under
//   (i)   today's schedule: one wave per region, lane replacement once 16 lanes are idle, quanta of <= 3 inner-node steps and
//         one leaf (msk_kernels.h: trace_replace / trav_quantum) — checked against the instrumented build's counters;
//   (ii)  rays re-binned every quantum by the node they wait at: a pool of P rays, sorted by (leaf?, cursor), cut into waves
//         of 64 — P = one region, four regions, everything (the limit of any treelet / node-sorted scheduler: the moving of
//         ray state through LDS or HBM that the re-binning needs is NOT charged);
//   (iii) the shadow ray and the extension ray of a slot walked together by one lane (two cursors, one step of either per
//         step slot).
// Reported per scheme: wave-level inner-node steps and triangle steps per ray, and lanes busy per step.  A wave-level step is
// the unit the kernel pays for: its VALU issue is per wave, whatever the number of live lanes.
//   g++ -O2 -std=c++17 -ffp-contract=off -I misaki-render_amd/csrc tools/micro/sched_model.cpp -o gpurun_scratch/sched_model
//   gpurun_scratch/sched_model positions.bin rays.bin [max_inner = 3] [refill = 16]
// Statistics only: full-precision child boxes (the kernel walks half-float ones: slightly looser) and a plain Moeller-Trumbore.
#include "msk_bvh.h"
#include <algorithm>
#include <cstdio>
#include <cstring>

using namespace mskbvh;

static const uint32_t DONE = 0xffffffffu, LEAF = 0x80000000u;
static Built g_b;
static std::vector<float> g_pos;

struct RayIn { float o[3], tmin, d[3], tmax, s[3], smax; bool has_shadow; };
struct Trav {            // one ray's traversal state (TravState of the kernel)
    float o[3], d[3], idir[3], oi[3], tmin, tmax, best;
    uint32_t cur, stack[192]; int sp; bool any, found;
};

static void begin(Trav &t, const float *o, const float *d, float tmin, float tmax, bool any) {
    for (int a = 0; a < 3; ++a) { t.o[a] = o[a]; t.d[a] = d[a]; t.idir[a] = std::max(-1e25f, std::min(1e25f, 1.f / d[a])); t.oi[a] = o[a] * t.idir[a]; }
    t.tmin = tmin; t.tmax = tmax; t.best = tmax; t.sp = 0; t.any = any; t.found = false;
    t.cur = g_b.root_ref4;
    if (d[0] == 0.f && d[1] == 0.f && d[2] == 0.f) t.cur = DONE;        // a parked path (d = 0): hits nothing
}
static bool tri_hit(const float *p, const Trav &r, float tmax, float *t_out) {
    float e1[3] = {p[3] - p[0], p[4] - p[1], p[5] - p[2]}, e2[3] = {p[6] - p[0], p[7] - p[1], p[8] - p[2]};
    float pv[3] = {r.d[1] * e2[2] - r.d[2] * e2[1], r.d[2] * e2[0] - r.d[0] * e2[2], r.d[0] * e2[1] - r.d[1] * e2[0]};
    float det = e1[0] * pv[0] + e1[1] * pv[1] + e1[2] * pv[2];
    if (std::fabs(det) < 1e-12f) return false;
    float inv = 1.f / det, tv[3] = {r.o[0] - p[0], r.o[1] - p[1], r.o[2] - p[2]};
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
// one inner-node step (node4_step): slab test of the four children, nearest first, the others pushed
static void node_step(Trav &t) {
    const float *nd = &g_b.nodes4[(size_t) t.cur * 32];
    const uint32_t *refs = (const uint32_t *) (nd + 24);
    float tn[4]; uint32_t rf[4]; int nh = 0;
    for (int i = 0; i < 4; ++i) {
        if (refs[i] == kEmpty4) continue;
        float t0 = t.tmin, t1 = t.best;
        for (int a = 0; a < 3; ++a) {
            const float ta = nd[a * 4 + i] * t.idir[a] - t.oi[a], tb = nd[12 + a * 4 + i] * t.idir[a] - t.oi[a];
            t0 = std::max(t0, std::min(ta, tb)); t1 = std::min(t1, std::max(ta, tb));
        }
        if (t0 <= t1 * 1.0000004f) { tn[nh] = t0; rf[nh] = refs[i]; ++nh; }
    }
    for (int i = 1; i < nh; ++i) for (int j = i; j > 0 && tn[j] < tn[j - 1]; --j) { std::swap(tn[j], tn[j - 1]); std::swap(rf[j], rf[j - 1]); }
    if (nh == 0) t.cur = t.sp > 0 ? t.stack[--t.sp] : DONE;
    else { for (int i = nh - 1; i >= 1; --i) t.stack[t.sp++] = rf[i]; t.cur = rf[0]; }
}
// one leaf: returns the triangle tests made (an any-hit query stops at its hit)
static uint32_t leaf_step(Trav &t) {
    const uint32_t first = (t.cur & 0x7fffffffu) >> 5, cnt = t.cur & 31u;
    uint32_t tests = 0;
    for (uint32_t i = 0; i < cnt; ++i) {
        ++tests;
        uint32_t prim; std::memcpy(&prim, &g_b.tris[(size_t) (first + i) * 16 + 3], 4);
        float th;
        if (tri_hit(&g_pos[(size_t) (prim & 0x3ffffffu) * 9], t, t.best, &th)) {
            if (t.any) { t.found = true; break; }
            if (th < t.best) t.best = th;
        }
    }
    t.cur = (t.sp > 0 && !t.found) ? t.stack[--t.sp] : DONE;
    return tests;
}
static bool at_inner(const Trav &t) { return t.cur != DONE && !(t.cur & LEAF); }
static bool at_leaf(const Trav &t) { return t.cur != DONE && (t.cur & LEAF); }

struct Tally {
    unsigned long long rays = 0, quanta = 0, lanes_q = 0, node_w = 0, node_l = 0, tri_w = 0, tri_l = 0, leaf_l = 0, distinct_nodes = 0;
    void print(const char *name) const {
        const double r = (double) std::max(rays, 1ull);
        std::printf("%-44s node steps %.3f/ray at %4.1f lanes | triangle steps %.3f/ray at %4.1f lanes | wave steps %.3f/ray | lane: %.2f nodes %.2f leaves %.2f tris | %.1f distinct nodes per node step\n",
                    name, node_w / r, (double) node_l / std::max(node_w, 1ull), tri_w / r, (double) tri_l / std::max(tri_w, 1ull), (node_w + tri_w) / r,
                    node_l / r, leaf_l / r, tri_l / r, (double) distinct_nodes / std::max(node_w, 1ull));
    }
};

// a slot = the shadow ray (if any), then the extension ray; `phase` 0 shadow, 1 extension
struct Job { const RayIn *in; Trav t; int phase; bool done; };
static void job_begin(Job &j, const RayIn *in) {
    j.in = in; j.done = false;
    if (in->has_shadow) { j.phase = 0; begin(j.t, in->o, in->s, in->tmin, in->smax, true); }
    else { j.phase = 1; begin(j.t, in->o, in->d, in->tmin, in->tmax, false); }
}
// after a quantum: ray finished? -> next phase or done; returns rays completed (0 / 1)
static int job_advance(Job &j) {
    if (j.t.cur != DONE) return 0;
    if (j.phase == 0) { j.phase = 1; begin(j.t, j.in->o, j.in->d, j.in->tmin, j.in->tmax, false); return 1; }
    j.done = true;
    return 1;
}
// one quantum of a wave over `lanes` (pointers to jobs; nullptr = idle lane): the SIMT loop runs max-over-lanes iterations
static void wave_quantum(Job *const *lanes, int n, int max_inner, Tally &ty) {
    int active = 0;
    for (int l = 0; l < n; ++l) active += lanes[l] != nullptr;
    if (!active) return;
    ++ty.quanta; ty.lanes_q += active;
    for (int k = 0; k < max_inner; ++k) {
        int here = 0;
        uint32_t seen[64]; int ns = 0;
        for (int l = 0; l < n; ++l) if (lanes[l] && at_inner(lanes[l]->t)) {
            ++here;
            const uint32_t c = lanes[l]->t.cur; bool dup = false;
            for (int q = 0; q < ns; ++q) if (seen[q] == c) { dup = true; break; }
            if (!dup) seen[ns++] = c;
        }
        if (!here) break;
        ++ty.node_w; ty.node_l += here; ty.distinct_nodes += ns;
        for (int l = 0; l < n; ++l) if (lanes[l] && at_inner(lanes[l]->t)) node_step(lanes[l]->t);
    }
    uint32_t max_tests = 0;
    for (int l = 0; l < n; ++l) if (lanes[l] && at_leaf(lanes[l]->t)) {
        ++ty.leaf_l;
        const uint32_t tests = leaf_step(lanes[l]->t);
        ty.tri_l += tests; max_tests = std::max(max_tests, tests);
    }
    ty.tri_w += max_tests;
}

// (i) the kernel's schedule on one region
static void model_replace(const std::vector<RayIn> &rays, int max_inner, int refill, Tally &ty) {
    std::vector<Job> job(64);
    Job *lane[64] = {};
    size_t next = 0;
    for (;;) {
        int idle = 0;
        for (int l = 0; l < 64; ++l) idle += lane[l] == nullptr;
        if (next < rays.size() && idle && (idle >= refill || idle == 64)) {
            for (int l = 0; l < 64 && next < rays.size(); ++l) if (!lane[l]) { job_begin(job[l], &rays[next++]); lane[l] = &job[l]; }
        }
        bool any = false;
        for (int l = 0; l < 64; ++l) any |= lane[l] != nullptr;
        if (!any) break;
        wave_quantum(lane, 64, max_inner, ty);
        for (int l = 0; l < 64; ++l) if (lane[l]) { ty.rays += job_advance(*lane[l]); if (lane[l]->done) lane[l] = nullptr; }
    }
}

// (iv) shadow rays and extension rays as SEPARATE jobs of the region's wave (k_trace_q's scheme on a tree in HBM): the shadow rays
// first, then every slot's extension ray; a lane takes the next job when `refill` lanes are idle
static void model_jobs(const std::vector<RayIn> &rays, int max_inner, int refill, Tally &ty) {
    struct J { const RayIn *in; bool shadow; };
    std::vector<J> jobs;
    for (auto &r : rays) if (r.has_shadow) jobs.push_back(J{&r, true});
    for (auto &r : rays) jobs.push_back(J{&r, false});
    std::vector<Job> job(64);
    Job *lane[64] = {};
    size_t next = 0;
    for (;;) {
        int idle = 0;
        for (int l = 0; l < 64; ++l) idle += lane[l] == nullptr;
        if (next < jobs.size() && idle && (idle >= refill || idle == 64)) {
            for (int l = 0; l < 64 && next < jobs.size(); ++l) if (!lane[l]) {
                const J &j = jobs[next++];
                job[l].in = j.in; job[l].done = false; job[l].phase = 1;          // phase 1: finishing the ray finishes the job
                if (j.shadow) begin(job[l].t, j.in->o, j.in->s, j.in->tmin, j.in->smax, true);
                else begin(job[l].t, j.in->o, j.in->d, j.in->tmin, j.in->tmax, false);
                lane[l] = &job[l];
            }
        }
        bool any = false;
        for (int l = 0; l < 64; ++l) any |= lane[l] != nullptr;
        if (!any) break;
        wave_quantum(lane, 64, max_inner, ty);
        for (int l = 0; l < 64; ++l) if (lane[l]) { ty.rays += job_advance(*lane[l]); if (lane[l]->done) lane[l] = nullptr; }
    }
}

// (ii) a pool of P slots, re-binned every quantum: sort by (leaf?, cursor), waves of 64 consecutive jobs
static void model_rebin(const std::vector<const RayIn *> &queue, size_t P, int max_inner, bool split_types, Tally &ty) {
    std::vector<Job> pool(std::min(P, queue.size()));
    std::vector<Job *> live;
    size_t next = 0;
    for (auto &j : pool) { job_begin(j, queue[next++]); live.push_back(&j); }
    while (!live.empty()) {
        std::sort(live.begin(), live.end(), [](const Job *a, const Job *b) {
            const uint32_t ka = a->t.cur, kb = b->t.cur;
            return ka < kb;                         // inner nodes (bit 31 clear) first, by node; then leaves by leaf
        });
        // with split_types a wave never mixes rays waiting at an inner node with rays waiting at a leaf
        size_t i = 0;
        while (i < live.size()) {
            size_t e = std::min(live.size(), i + 64);
            if (split_types) { const bool leaf0 = at_leaf(live[i]->t); size_t k = i; while (k < e && at_leaf(live[k]->t) == leaf0) ++k; e = k; }
            wave_quantum(&live[i], (int) (e - i), max_inner, ty);
            i = e;
        }
        std::vector<Job *> keep;
        for (Job *j : live) {
            ty.rays += job_advance(*j);
            if (!j->done) keep.push_back(j);
            else if (next < queue.size()) { job_begin(*j, queue[next++]); keep.push_back(j); }
        }
        live.swap(keep);
    }
}

// (iii) both rays of a slot in one lane: per step slot the lane advances whichever of its two rays wants that kind of step
struct Pair { const RayIn *in; Trav a, b; bool has_a, done; };
static void model_pairs(const std::vector<RayIn> &rays, int max_inner, int refill, Tally &ty) {
    std::vector<Pair> pr(64);
    Pair *lane[64] = {};
    size_t next = 0;
    auto fin = [](const Pair &p) { return (!p.has_a || p.a.cur == DONE) && p.b.cur == DONE; };
    for (;;) {
        int idle = 0;
        for (int l = 0; l < 64; ++l) idle += lane[l] == nullptr;
        if (next < rays.size() && idle && (idle >= refill || idle == 64)) {
            for (int l = 0; l < 64 && next < rays.size(); ++l) if (!lane[l]) {
                Pair &p = pr[l]; p.in = &rays[next++]; p.has_a = p.in->has_shadow; p.done = false;
                if (p.has_a) begin(p.a, p.in->o, p.in->s, p.in->tmin, p.in->smax, true); else p.a.cur = DONE;
                begin(p.b, p.in->o, p.in->d, p.in->tmin, p.in->tmax, false);
                lane[l] = &p;
            }
        }
        int active = 0;
        for (int l = 0; l < 64; ++l) active += lane[l] != nullptr;
        if (!active) break;
        ++ty.quanta; ty.lanes_q += active;
        for (int k = 0; k < max_inner; ++k) {
            int here = 0;
            for (int l = 0; l < 64; ++l) if (lane[l]) {
                Trav *t = (lane[l]->has_a && at_inner(lane[l]->a)) ? &lane[l]->a : at_inner(lane[l]->b) ? &lane[l]->b : nullptr;
                if (t) { ++here; node_step(*t); }
            }
            if (!here) break;
            ++ty.node_w; ty.node_l += here;
        }
        uint32_t max_tests = 0;
        for (int l = 0; l < 64; ++l) if (lane[l]) {
            Trav *t = (lane[l]->has_a && at_leaf(lane[l]->a)) ? &lane[l]->a : at_leaf(lane[l]->b) ? &lane[l]->b : nullptr;
            if (t) { ++ty.leaf_l; const uint32_t tests = leaf_step(*t); ty.tri_l += tests; max_tests = std::max(max_tests, tests); }
        }
        ty.tri_w += max_tests;
        for (int l = 0; l < 64; ++l) if (lane[l] && fin(*lane[l])) { ty.rays += lane[l]->has_a ? 2 : 1; lane[l] = nullptr; }
    }
}

int main(int argc, char **argv) {
    if (argc < 3) { std::fprintf(stderr, "usage: sched_model positions.bin rays.bin [max_inner] [refill]\n"); return 2; }
    const int max_inner = argc > 3 ? atoi(argv[3]) : 3, refill = argc > 4 ? atoi(argv[4]) : 16;
    FILE *f = std::fopen(argv[1], "rb");
    if (!f) { std::perror(argv[1]); return 2; }
    std::fseek(f, 0, SEEK_END); const long bytes = std::ftell(f); std::fseek(f, 0, SEEK_SET);
    const uint32_t n = (uint32_t) (bytes / 36);
    g_pos.resize((size_t) n * 9);
    if (std::fread(g_pos.data(), 36, n, f) != n) return 2;
    std::fclose(f);
    Box all; for (uint32_t i = 0; i < n * 3; ++i) all.grow(V3{g_pos[i * 3], g_pos[i * 3 + 1], g_pos[i * 3 + 2]});
    const float diag = std::sqrt((all.hi.x - all.lo.x) * (all.hi.x - all.lo.x) + (all.hi.y - all.lo.y) * (all.hi.y - all.lo.y) + (all.hi.z - all.lo.z) * (all.hi.z - all.lo.z));
    float amax = 0;
    for (float v : {all.lo.x, all.lo.y, all.lo.z, all.hi.x, all.hi.y, all.hi.z}) amax = std::max(amax, std::fabs(v));
    g_b = build(g_pos.data(), n, 0.5f * 1e-5f * std::max(diag, amax));          // msk_gpu.hip's padding rule
    collapse4(g_b, true);
    std::printf("%u triangles, %zu 4-wide nodes, depth %d; quantum = %d inner-node steps + one leaf, refill at %d idle lanes\n", n, g_b.nodes4.size() / 32, g_b.max_depth4, max_inner, refill);

    f = std::fopen(argv[2], "rb");
    if (!f) { std::perror(argv[2]); return 2; }
    uint32_t head[4];
    if (std::fread(head, 4, 4, f) != 4 || head[0] != 0x524b534du) { std::fprintf(stderr, "not a ray dump\n"); return 2; }
    std::vector<std::vector<RayIn>> regions(head[1]);
    size_t slots = 0, shadows = 0;
    for (auto &rg : regions) {
        uint32_t cn[2];
        if (std::fread(cn, 4, 2, f) != 2) return 2;
        rg.resize(cn[0]);
        for (uint32_t c = 0; c < cn[0]; ++c) {
            float r[12];
            if (std::fread(r, 4, 12, f) != 12) return 2;
            RayIn &q = rg[c];
            std::memcpy(q.o, r, 12); q.tmin = r[3]; std::memcpy(q.d, r + 4, 12); q.tmax = r[7]; std::memcpy(q.s, r + 8, 12); q.smax = r[11];
            q.has_shadow = c < cn[1];
        }
        slots += cn[0]; shadows += cn[1];
    }
    std::fclose(f);
    std::printf("%u regions of %u slots (every %u-th of the launch): %zu live slots, %zu with a shadow ray = %zu rays\n\n", head[1], head[2], head[3], slots, shadows, slots + shadows);

    Tally a;
    for (auto &rg : regions) model_replace(rg, max_inner, refill, a);
    a.print("(i)   lane replacement (today)");
    std::printf("      quanta per ray %.2f, lanes active per quantum %.1f\n", (double) a.quanta / a.rays, (double) a.lanes_q / std::max(a.quanta, 1ull));
    { Tally t; for (auto &rg : regions) model_replace(rg, 1, refill, t); t.print("(i')  lane replacement, quantum = 1 node step"); }
    { Tally t; for (auto &rg : regions) model_replace(rg, max_inner, 1, t); t.print("(i'') lane replacement, refill at 1 idle lane"); }
    for (size_t per : {(size_t) 1, (size_t) 4, regions.size()}) {
        for (int split = 0; split < 2; ++split) {
            Tally t;
            for (size_t r0 = 0; r0 < regions.size(); r0 += per) {
                std::vector<const RayIn *> q;
                size_t cap = 0;
                for (size_t r = r0; r < std::min(regions.size(), r0 + per); ++r) { for (auto &x : regions[r]) q.push_back(&x); cap += regions[r].size(); }
                model_rebin(q, cap, max_inner, split != 0, t);
            }
            char name[128];
            std::snprintf(name, sizeof name, "(ii)  re-binned by node, pool = %zu region(s)%s", per, split ? ", node / leaf waves apart" : "");
            t.print(name);
        }
    }
    { Tally t; std::vector<const RayIn *> q; for (auto &rg : regions) for (auto &x : rg) q.push_back(&x); model_rebin(q, q.size(), 1, true, t); t.print("(ii') everything, re-binned after EVERY node step"); }
    { Tally t; for (auto &rg : regions) model_pairs(rg, max_inner, refill, t); t.print("(iii) shadow + extension ray in one lane"); }
    { Tally t; for (auto &rg : regions) model_jobs(rg, max_inner, refill, t); t.print("(iv)  shadow and extension rays as separate jobs");
      std::printf("      quanta per region %.1f (today: %.1f) — the length of a launch is its slowest wave's quanta\n", (double) t.quanta / regions.size(), (double) a.quanta / regions.size()); }
    return 0;
}
