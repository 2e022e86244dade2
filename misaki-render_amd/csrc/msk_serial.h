// msk_serial.h — MSK_RNG_PCG_BLOCK on the device: the reference's sampler semantics as written (included by msk_gpu.hip).
//
// samplers/independent.cpp:9-35 gives a render job ONE PCG32 stream per image block (integrator.cpp:56-58 clones the
// sampler per block; SURVEY F7 / oracle D2 fix its seed): a block's samples draw from it in the order the scalar loops
// issue them — y, x, s (integrator.cpp:89-98), and inside a sample: film position (2), wavelength (1), aperture (2),
// then per bounce NEE (2), BSDF lobe (1), BSDF direction (2) and, from rr_depth on, Russian roulette (1)
// (integrator.cpp:103-107, path.cpp:56-73,116-122).  Where a sample's draws start depends on how long every earlier path
// of the block was, so a block is sequential by construction: here ONE LANE renders ONE BLOCK, sample after sample, path
// after path, and splats into the block's bordered buffer with ImageBlock::put's own loops (imageblock.cpp:55-114), in
// order.  Blocks are independent, so a film of B blocks is B lanes wide — a fidelity mode (BASELINE config 1, cbox 256^2
// @ 16 spp, is 64 lanes for three seconds), not a fast one; the wavefront path (MSK_RNG_COUNTER) is the product's hot path.
// The arithmetic of a bounce is the wavefront kernel's (shade_region, msk_kernels.h), call for call — make_interaction,
// bsdf_eval_pdf, bsdf_sample, the emitter sampling of scene.cpp:68-103 — only the control flow is the scalar loop's
// (oracle.cpp: path_sample): a path whose throughput is zero keeps drawing until the loop ends it, as the reference's does,
// because the stream position of the next sample depends on it.  The tree is walked as a binary tree in HBM (traverse<>,
// the arrays every scene has), whatever the scene's trace mode: any tree gives the same hits.
#pragma once

namespace msk {

struct Pcg32 {                                  // core/mathutils.h:85-121
    uint64_t state, inc;
    MSK_DEV uint32_t next_u32() {
        const uint64_t old = state;
        state = old * 0x5851f42d4c957f2dULL + inc;
        const uint32_t xs = (uint32_t) (((old >> 18u) ^ old) >> 27u), rot = (uint32_t) (old >> 59u);
        return (xs >> rot) | (xs << ((~rot + 1u) & 31u));
    }
    MSK_DEV void seed(uint64_t initstate, uint64_t initseq) {
        state = 0u; inc = (initseq << 1u) | 1u;
        next_u32(); state += initstate; next_u32();
    }
    MSK_DEV float next_float() { return u32_to_float01(next_u32()); }
};

struct SerialParams {
    uint64_t seed;
    uint32_t spp, sample_first, sample_stride;
    int32_t rr_depth, max_depth, hide_emitters;
    const BlockInfo *blocks; uint32_t n_blocks;
    float *block_buf; uint32_t buf_stride;
    uint32_t *stack_ovf;
    uint32_t per_wave;          // 1: image block = wave index, only lane 0 works (see k_path_serial)
    unsigned long long *counters;               // [0] samples [1] segments [2] shadow rays [3] invalid samples (imageblock.cpp:57-81)
};

// ImageBlock::put(pos, value) (imageblock.cpp:55-114) into a bordered block buffer of sx x sy pixels, 5 channels
MSK_DEV void serial_put(const DeviceScene &sc, float *data, int sx, int sy, int org_x, int org_y, float pos_x, float pos_y, const float (&value)[5]) {
    const float radius = sc.filter_radius;
    const float px = pos_x - 0.5f - (float) org_x, py = pos_y - 0.5f - (float) org_y;       // org = offset - border
    const int lo_x = max((int) ceilf(px - radius), 0), lo_y = max((int) ceilf(py - radius), 0);
    const int hi_x = min((int) floorf(px + radius), sx - 1), hi_y = min((int) floorf(py + radius), sy - 1);
    for (int y = lo_y; y <= hi_y; ++y) {
        const float wy = sc.lut[min((int) fabsf(((float) y - py) * sc.filter_scale), MSK_FILTER_RESOLUTION)];      // rfilter.h:13-16
        for (int x = lo_x; x <= hi_x; ++x) {
            const float wx = sc.lut[min((int) fabsf(((float) x - px) * sc.filter_scale), MSK_FILTER_RESOLUTION)];
            const float weight = wx * wy;
            float *dest = data + ((size_t) y * sx + x) * 5;
#pragma unroll
            for (int k = 0; k < 5; ++k) dest[k] += weight * value[k];
        }
    }
}

__global__ void __launch_bounds__(MSK_BLOCK)
k_path_serial(DeviceScene sc, SerialParams prm) {
    extern __shared__ float4 lds_dyn[];
    uint32_t *stack_base = (uint32_t *) lds_dyn;
    const LaneStack<true> stack{stack_base + threadIdx.x, prm.stack_ovf + (size_t) blockIdx.x * MSK_BLOCK + threadIdx.x,
                                (int) sc.stack_entries, (size_t) gridDim.x * MSK_BLOCK, nullptr};
    // one image block per lane — or, while there are fewer blocks than the GPU has room for waves, per WAVE (lane 0): 64 scalar
    // loops packed into one wave execute the union of their branches, one loop per wave executes its own (prm.per_wave)
    const uint32_t tid = blockIdx.x * MSK_BLOCK + threadIdx.x;
    if (prm.per_wave && (threadIdx.x & (MSK_WAVE - 1u))) return;
    const uint32_t bi = prm.per_wave ? tid / MSK_WAVE : tid;
    if (bi >= prm.n_blocks) return;
    SceneTablesR tb;                        // (the fidelity mode carries the table forms of tabulated spectra always: one instantiation)
    static_cast<SceneTables &>(tb) = stage_tables<false>(sc, nullptr);
    const BlockInfo b = prm.blocks[bi];
    const int border = sc.filter_border;
    const int sx = b.size_x + 2 * border, sy = b.size_y + 2 * border;
    float *data = prm.block_buf + (size_t) b.slot * prm.buf_stride;
    Pcg32 rng;
    rng.seed(0x853c49e6748fea9bULL + prm.seed, 0xda3e39cb94b95bdbULL);      // oracle D2; independent.cpp:20-26
    const uint32_t n_em = sc.n_emitters;
    const uint32_t sstride = prm.sample_stride ? prm.sample_stride : 1u;
    unsigned long long n_samples = 0, n_segments = 0, n_shadow = 0, n_invalid = 0;
    auto closest = [&](f3 o, f3 d, float tmin, float tmax) {
        float t, u, v; uint32_t prim;
        traverse<false, true>(sc.nodes, sc.tris, sc.tri_pad, sc.root_ref, sc.n_tris, o, d, tmin, tmax, stack, &t, &u, &v, &prim);
        const bool valid = (prim != MSK_NO_PRIM) && (t != tmax);                // scene.cpp:234
        return make_float4(valid ? t : MSK_INF_F, u, v, __uint_as_float(valid ? (prim & MSK_PRIM_ID) : MSK_PRIM_ID));
    };
    for (int y = 0; y < b.size_y; ++y)
        for (int x = 0; x < b.size_x; ++x)
            for (uint32_t s = 0; s < prm.spp; ++s) {
                if (s < prm.sample_first || (s - prm.sample_first) % sstride != 0u) continue;      // msk_gpu.h: s = first + k stride
                // ---- render_sample (integrator.cpp:103-126)
                const float jx = rng.next_float(), jy = rng.next_float();
                const float wsample = rng.next_float();
                (void) rng.next_float(); (void) rng.next_float();                   // the aperture sample (perspective.cpp:22: unused)
                const float px = (float) (x + b.off_x) + jx, py = (float) (y + b.off_y) + jy;
                spec wl;
#pragma unroll
                for (int q = 0; q < 4; ++q) wl.v[q] = wavelength_of(wsample, q);
                float r4[4];
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    r4[q] = ((sc.s2c[q * 4 + 0] * px + sc.s2c[q * 4 + 1] * py) + sc.s2c[q * 4 + 2] * 0.f) + sc.s2c[q * 4 + 3] * 1.f;
                const f3 near_p = mk3(r4[0] / r4[3], r4[1] / r4[3], r4[2] / r4[3]);
                const f3 dl = normalized(near_p);
                const float inv_z = 1.f / dl.z;
                float o4[4];
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    o4[q] = ((sc.to_world[q * 4 + 0] * 0.f + sc.to_world[q * 4 + 1] * 0.f) + sc.to_world[q * 4 + 2] * 0.f) + sc.to_world[q * 4 + 3] * 1.f;
                f3 ro = mk3(o4[0] / o4[3], o4[1] / o4[3], o4[2] / o4[3]);
                const float *m = sc.to_world;
                f3 rd = mk3(m[0] * dl.x + (m[1] * dl.y + m[2] * dl.z), m[4] * dl.x + (m[5] * dl.y + m[6] * dl.z), m[8] * dl.x + (m[9] * dl.y + m[10] * dl.z));
                ++n_samples;
                // ---- PathTracer::sample (path.cpp:23-125), the scalar loop
                spec thr = splat(1.f), res = splat(0.f);
                float eta = 1.f;
                ++n_segments;
                float4 hit = closest(ro, rd, sc.near_clip * inv_z, sc.far_clip * inv_z);
                for (int depth = 1; depth <= prm.max_depth || prm.max_depth < 0; ++depth) {
                    if (hit.x == MSK_INF_F) {                                          // path.cpp:34-41
                        if (depth == 1 && !prm.hide_emitters && sc.env_emitter >= 0) res = res + thr * emitter_radiance(tb, sc.env_emitter, wl);
                        break;
                    }
                    const Interaction si = make_interaction(tb, hit, rd);
                    BsdfRec bs = load_bsdf(tb, si.bsdf_id);
                    f3 wi_s = si.wi;
                    bool flipped = false;
                    {
                        const int back = __float_as_int(bs.a.y);
                        if (back >= 0 && wi_s.z < 0.f) { wi_s.z = -wi_s.z; flipped = true; if (back != si.bsdf_id) bs = load_bsdf(tb, back); }
                    }
                    if (si.emitter_id >= 0 && depth == 1 && !prm.hide_emitters && si.wi.z > 0.f)      // path.cpp:42-47, area.cpp:51-54
                        res = res + thr * emitter_radiance(tb, si.emitter_id, wl);
                    if (depth >= prm.max_depth && prm.max_depth > 0) break;             // path.cpp:48-49
                    spec refl = splat(0.f);
                    if (__float_as_int(bs.a.x) == 0) {
                        f3 c = mk3(bs.a.z, bs.a.w, bs.b.x);
                        float scale = bs.ior.w;
                        const uint32_t tex = __float_as_uint(bs.ior.z);
                        if (tex) { c = checkerboard_coeffs(tb, tex, hit); scale = 1.f; }
                        refl = spectrum_eval(tb, make_float4(c.x, c.y, c.z, scale), wl);
                    }
                    const float tmin = (1.f + max_abs(si.p)) * MSK_RAY_EPS_F;          // interaction.h:40-44
                    // ---- next-event estimation (path.cpp:56-67, scene.cpp:68-103); the draw is made whatever the scene holds
                    float nee_pdf = 0.f;
                    {
                        f2 u; u.x = rng.next_float(); u.y = rng.next_float();
                        if (n_em > 0) {
                            uint32_t e = 0;
                            float light_sel_pdf = 1.f;
                            if (n_em > 1) {
                                light_sel_pdf = 1.f / n_em;
                                uint32_t index = (uint32_t) (u.x * (float) n_em);
                                index = index < n_em - 1 ? index : n_em - 1;
                                u.x = (u.x - index * light_sel_pdf) * n_em;
                                e = index;
                            }
                            const float4 e0 = tb.emitters[2 * e], e1 = tb.emitters[2 * e + 1];
                            f3 d; float dist, pdf; spec emitter_val;
                            if ((int) e == sc.env_emitter) {                           // constant.cpp:53-72
                                d = square_to_uniform_sphere(u);
                                dist = 2.f * sc.env_radius;
                                pdf = MSK_INV_FOUR_PI_F;
                                emitter_val = emitter_radiance(tb, (int) e, wl) / pdf;
                                nee_pdf = pdf;
                            } else {
                                const uint32_t first_face = __float_as_uint(e1.y), n_faces = __float_as_uint(e1.z);
                                const float *cdf = tb.cdf + __float_as_uint(e1.w);
                                uint32_t lo = 0, hi = n_faces + 1;                     // Distribution1D::sample_reuse (core/distribution.h:106-116)
                                while (lo < hi) { uint32_t mid = (lo + hi) >> 1; if (!(u.y < cdf[mid])) lo = mid + 1; else hi = mid; }
                                int fidx = (int) lo - 1;
                                fidx = fidx < 0 ? 0 : fidx; fidx = fidx > (int) n_faces - 1 ? (int) n_faces - 1 : fidx;
                                u.y = (u.y - cdf[fidx]) / (cdf[fidx + 1] - cdf[fidx]);
                                const uint32_t lprim = first_face + (uint32_t) fidx;
                                const float4 la = tb.tri_verts[(size_t) lprim * 3], lb = tb.tri_verts[(size_t) lprim * 3 + 1], lc = tb.tri_verts[(size_t) lprim * 3 + 2];
                                const f3 p0 = mk3(la.x, la.y, la.z), p1 = mk3(lb.x, lb.y, lb.z), p2 = mk3(lc.x, lc.y, lc.z);
                                const f3 ed0 = p1 - p0, ed1 = p2 - p0;                 // mesh.cpp:103-133
                                const f2 bc = square_to_uniform_triangle(u);
                                const f3 lp = p0 + ed0 * bc.x + ed1 * bc.y;
                                const float4 lf = tb.tri_frames[(size_t) lprim * 3];
                                f3 ln = mk3(lf.x, lf.y, lf.z);
                                const int4 lmi = tb.mesh_info[__float_as_uint(la.w)];
                                if (lmi.z & 1) {
                                    const float4 na = tb.tri_normals[(size_t) lprim * 3], nb = tb.tri_normals[(size_t) lprim * 3 + 1], nc = tb.tri_normals[(size_t) lprim * 3 + 2];
                                    ln = normalized(mk3(na.x, na.y, na.z) * (1.f - bc.x - bc.y) + mk3(nb.x, nb.y, nb.z) * bc.x + mk3(nc.x, nc.y, nc.z) * bc.y);
                                }
                                pdf = e0.w;
                                d = lp - si.p;                                         // shape.cpp:64-78
                                const float dist2 = dot(d, d);
                                dist = __builtin_sqrtf(dist2);
                                d = d / dist;
                                const float dp = fabsf(dot(d, ln));
                                pdf *= (dp != 0.f) ? dist2 / dp : 0.f;
                                nee_pdf = e0.w * ((dp != 0.f) ? (dist * dist) / dp : 0.f);   // shape.cpp:80-86
                                if (dot(d, ln) < 0.f && pdf != 0.f) emitter_val = emitter_radiance(tb, (int) e, wl) / pdf;   // area.cpp:39-44
                                else { pdf = 0.f; emitter_val = splat(0.f); }
                            }
                            if (n_em > 1) { pdf *= light_sel_pdf; emitter_val = emitter_val * (float) n_em; nee_pdf = nee_pdf * (1.f / n_em); }
                            if (pdf != 0.f) {
                                f3 wo = si.sh.to_local(d);
                                if (flipped) wo.z = -wo.z;
                                spec bsdf_val; float bsdf_pdf;
                                bsdf_eval_pdf<false>(tb, bs, wi_s, wo, wl, refl, &bsdf_val, &bsdf_pdf);
                                const float w = mis_weight(pdf, bsdf_pdf);
                                const spec contrib = thr * emitter_val * bsdf_val * w;
                                if (any_nonzero(contrib)) {                            // scene.cpp:91-95: an occluded sample adds nothing
                                    ++n_shadow;
                                    float t, uu, vv; uint32_t pp_;
                                    const bool occluded = traverse<true, true>(sc.nodes, sc.tris, sc.tri_pad, sc.root_ref, sc.n_tris, si.p, d, tmin,
                                                                               dist * (1.f - MSK_SHADOW_EPS_F), stack, &t, &uu, &vv, &pp_);
                                    if (!occluded) res = res + contrib;
                                }
                            }
                        }
                    }
                    // ---- BSDF sampling (path.cpp:71-80)
                    const float sample1 = rng.next_float();
                    f2 u2; u2.x = rng.next_float(); u2.y = rng.next_float();
                    f3 wo_l; bool ok; float bs_pdf, bs_eta;
                    const spec bsdf_val = bsdf_sample<false>(tb, bs, wi_s, sample1, u2, wl, refl, &wo_l, &bs_pdf, &bs_eta, &ok);
                    f3 wo = mk3(0.f, 0.f, 0.f);
                    if (ok) { if (flipped) wo_l.z = -wo_l.z; wo = si.sh.to_world(wo_l); ++n_segments; }
                    // the sampled ray (a failed sample's zero direction finds nothing: path.cpp:89-97)
                    ro = si.p; rd = wo;
                    const float4 hit_b = ok ? closest(ro, rd, tmin, MSK_INF_F) : make_float4(MSK_INF_F, 0.f, 0.f, __uint_as_float(MSK_PRIM_ID));
                    spec value = splat(0.f);
                    bool hit_emitter = false;
                    float emitter_pdf = 0.f;
                    if (hit_b.x != MSK_INF_F) {
                        const Interaction sb = make_interaction(tb, hit_b, rd);
                        if (sb.emitter_id >= 0) {                                      // path.cpp:82-88; records.cpp:7-14; scene.cpp:105-112
                            if (sb.wi.z > 0.f) value = emitter_radiance(tb, sb.emitter_id, wl);
                            float pdf = tb.emitters[2 * sb.emitter_id].w;
                            const float dp = fabsf(dot(rd, sb.sh.n));
                            pdf *= (dp != 0.f) ? (sb.t * sb.t) / dp : 0.f;
                            if (n_em != 1) pdf = pdf * (1.f / n_em);
                            emitter_pdf = pdf;
                            hit_emitter = true;
                        }
                    } else if (sc.env_emitter >= 0) {                                   // path.cpp:90-95: `ds` is the NEE sample's record
                        value = emitter_radiance(tb, sc.env_emitter, wl);
                        emitter_pdf = nee_pdf;
                        hit_emitter = true;
                    } else {
                        break;                                                         // path.cpp:96-97
                    }
                    thr = thr * bsdf_val;                                              // path.cpp:99 (a failed sample: weight 0, pdf 0, eta 1)
                    eta *= bs_eta;                                                     // path.cpp:100
                    if (hit_emitter) res = res + thr * value * mis_weight(bs_pdf, emitter_pdf);
                    hit = hit_b;
                    if (depth + 1 >= prm.rr_depth) {                                   // path.cpp:116-122
                        const float q = fmin_std(max4(thr) * eta * eta, 0.95f);
                        if (rng.next_float() >= q) break;
                        thr = thr / q;
                    }
                }
                // ---- integrator.cpp:115-125: ray weight, XYZ, splat
                spec wgt;
#pragma unroll
                for (int k = 0; k < 4; ++k) wgt.v[k] = wavelength_weight(wl.v[k]);
                const spec result = res * wgt;
                float v5[5];
                spectrum_to_xyz(tb.cie, result, wl, &v5[0], &v5[1], &v5[2]);
                v5[3] = 1.f; v5[4] = 1.f;
                if (invalid_value(v5[0], v5[1], v5[2], true)) n_invalid += 1;                    // imageblock.cpp:57-81: warned about, splatted all the same
                serial_put(sc, data, sx, sy, b.off_x - border, b.off_y - border, px, py, v5);
            }
    atomicAdd(&prm.counters[0], n_samples); atomicAdd(&prm.counters[1], n_segments); atomicAdd(&prm.counters[2], n_shadow);
    if (n_invalid) atomicAdd(&prm.counters[3], n_invalid);
}

}  // namespace msk
