// msk_kernels.h — device data layout + the wavefront kernels (gfx950, wave64).
//
// Pipeline.  Path state lives in HBM as structure-of-arrays, split into
// per-wave REGIONS of `region_size` slots.  One wavefront iteration = two launches:
//
//   k_shade_gen  wave w sweeps its region 64 slots at a time: finishes the previous bounce of
//                every path (emitter hit + MIS, Russian roulette), runs the next bounce's
//                NEE + BSDF sampling (PathTracer::sample body, path.cpp:33-123), parks finished
//                paths in a wave-local LDS queue and writes their sample records 64 at a time,
//                compacts the survivors into the region's other half with a wave ballot + prefix
//                popcount (no block barrier, no global queue, no atomic) — those with a pending shadow
//                ray from the front, the others from the end (RegionView) — then fills what is free
//                with new camera samples (render_sample, integrator.cpp:103-116) from the region's own
//                static share of the pass (RegionCtl).
//   k_trace      per live slot: the pending shadow ray (Scene::ray_test, scene.cpp:255-273) and
//                the extension ray (Scene::ray_intersect, scene.cpp:216-253) against the
//                flattened BVH: k_trace<0> with nodes + triangles staged in LDS when the scene is
//                small, k_trace_r<2> (4-wide nodes in HBM/L2, lane replacement) when it is not.
//
// After the last iteration k_resolve_blocks / k_film_put rebuild the film from the per-sample
// records in exactly the order ImageBlock::put / Film::put accumulate them
// (imageblock.cpp:55-114, 133-173), so the film is bit-identical to the scalar CPU order.
#pragma once
#include "msk_device.h"
#include <type_traits>
#include "../../include/msk_gpu.h"

namespace msk {

#define MSK_WAVE 64
#define MSK_BSDF_F4 7   /* float4 per BSDF record (built from msk_bsdf_desc by msk_gpu_scene_create) */
#define MSK_BLOCK 256
#define MSK_LEAF_BIT 0x80000000u  /* child ref: leaf = BIT | first_tri << 5 | count ; inner = node index */
#define MSK_NO_PRIM 0xffffffffu
// The prim word of a triangle record / hit record: bits 0..25 the scene-global triangle index (msk_gpu_scene_create rejects
// scenes of 2^26 triangles and more), bits 27..28 the MATERIAL CLASS of the triangle's BSDF (0 diffuse, 1 rough conductor,
// 2 rough dielectric; a miss reads 3), written into the leaf records at scene creation so that the traversal kernels deliver
// it with the hit for free.  k_shade_gen sorts a region's paths by it (material-sorted shading, shade_region).
#define MSK_PRIM_ID 0x03ffffffu
#define MSK_CLASS_SHIFT 27
#define MSK_N_CLASSES 4
#define MSK_DEPTH_SHIFT 20                 /* id.y: 20 bits of sample index (msk_gpu_render rejects more than 2^20 samples per pixel and pass), */
#define MSK_SI_MASK 0xfffffu               /* 12 bits of depth: a path that reaches bounce 4094 is cut (Russian roulette makes that a 1e-89 event) */
#define MSK_MAX_DEPTH 4094u

struct DeviceScene {
    const float4 *nodes;        // 4 x float4 per node (msk_bvh.h)
    const float4 *nodes4;       // 8 x float4 per 4-wide node, or nullptr (scenes whose BVH is staged in LDS use `nodes`)
    uint32_t root_ref4, n_nodes4;
    const float4 *nodes4q;      // 4 x float4 per 4-wide node with quantised child boxes (msk_bvh.h: Built::nodes4q; same numbering as
                                // nodes4), or nullptr: what the traversal of a tree in HBM/L2 reads
    const float4 *tris3;        // 3 x float4 per triangle, leaf order: [v0 | prim] [e1 | e2.x] [e2.y e2.z - -] — `tris` without the
                                // normal (recomputed per test: e2 x e1, the builder's operations): three loads per test instead of four
    const float4 *nodes8;       // 8 x float4 per 8-wide node with quantised child boxes (msk_bvh.h: collapse8), or nullptr
    uint32_t root_ref8, n_nodes8;
    const float4 *tris;         // 4 x float4 per triangle, leaf order (msk_bvh.h)
    const float4 *tri_bounds;   // 2 x float4 per triangle, leaf order: padded bounds of the D10 predicate (staged next to the
                                // triangles for LDS-resident scenes; trees in HBM recompute them from the record instead)
    float tri_pad;
    const float4 *tri_verts;    // 3 x float4 per triangle, scene-global order: p0|mesh p1|- p2|-
    const float4 *tri_frames;   // 3 x float4 per triangle, written by k_tri_frames at scene creation: what make_interaction would
                                // compute for every hit but is the same for every hit of a triangle — geometric normal | frame s
                                // (dp_du for a mesh with vertex normals, whose frame follows the interpolated normal) | frame t
    const float4 *tri_normals;  // 3 x float4 per triangle (n0 n1 n2) or nullptr
    const float4 *tri_uvs;      // 2 x float4 per triangle (uv0 uv1 | uv2 -) or nullptr
    const int4 *mesh_info;      // {bsdf_id, emitter_id, flags(1=normals,2=texcoords), first_face}
    const float4 *bsdfs;        // MSK_BSDF_F4 x float4 per bsdf (BsdfRec; spectra as spectrum records, reflectance_texture as the
                                // float4 offset of the texture record), then 3 x float4 per texture: {color0, m02} {color1, m12}
                                // {m00 m01 m10 m11}
    uint32_t n_bsdf_f4;         // float4 count of `bsdfs` (records + textures)
    const float4 *emitters;     // 2 x float4 per emitter: {c0,c1,c2,inv_area} {mesh,first_face,face_count,cdf_off (uint bits)}
    const float *emitter_d65;   // 95 floats per emitter: d65 * d65_scale, or the values of a `regular` radiance (ABI v7)
    const float4 *emitter_grid; // per emitter {lambda_min, inv_interval, last segment (uint bits), 1 = the table IS the radiance (no
                                // sigmoid factor)} of that table: {360, 0.2, 93, 0} for the D65 form.  Read by the general shading
                                // variant only: a scene with a tabulated emitter never runs the diffuse-only one
    const float *spectra;       // values of the scene's other `regular` spectra (BSDF parameters), n_spectra floats; a spectrum
                                // record {lambda_min, inv_interval, first | last << 24 (uint bits), -1} names its part
    uint32_t n_spectra;
    const float *cdf;           // concatenated area CDFs (face_count+1 each)
    const float *cie;           // 285 floats
    uint32_t n_nodes, n_tris, n_emitters, n_meshes, n_bsdfs, cdf_len;
    uint32_t root_ref;          // packed child ref of the root
    uint32_t stack_entries;     // per-lane traversal stack entries kept in LDS
    uint32_t stack_total;       // entries a traversal can need (BVH depth + 2, or 3 per level of the 4-wide tree); the rest overflows to HBM
    float s2c[16], to_world[16];
    float near_clip, far_clip;
    int32_t width, height;
    int32_t crop_x, crop_y, crop_w, crop_h;     // the window of the film the render calls write (msk_film_desc: the whole film by default)
    float filter_radius, filter_scale;
    int32_t filter_border;
    float lut[33];
    int32_t env_emitter;        // index of the constant environment emitter in `emitters`, or -1 (scene.cpp:35-41)
    float env_radius;           // ConstantBackgroundEmitter::m_bsphere.radius after set_scene (constant.cpp:21-28)
};

struct PathState {
    uint2 *id;          // {film pixel y*W+x, owned-sample index si | depth << MSK_DEPTH_SHIFT}: the RNG key needs the film pixel and
                        // the sample index at every bounce; the pass pixel (the record's address) follows from the film pixel
                        // through PassParams::pix_to_j when the path is finished
    float4 *wl, *thr, *res;
    float4 *ray_o;      // o.xyz, tmin
    float4 *ray_d;      // d.xyz, then: camera ray: tmax (> 0); bounce ray (tmax = inf): minus the pdf of the BSDF sample that made
                        // it (BSDFSample::pdf > 0, needed by the MIS weight at the next hit).  d = 0 marks a path whose
                        // throughput is zero and which only waits for its shadow ray: such a ray hits nothing
    float4 *sh;         // shadow d.xyz, tmax                                        } written / read only for the slots
    float4 *contrib;    // NEE contribution added when the shadow ray is unoccluded  } c < ns of a region (RegionView)
    float4 *hit;        // t,u,v,prim
    float2 *aux;        // {eta (path.cpp:29), pdf_emitter_direct of the last NEE record (path.cpp:103-106 on an environment hit)}
};

// Per-region bookkeeping, one record per wave-region, touched only by its owner wave: no atomics
// anywhere on the wavefront path (a single hot cache line of global counters serialises at
// ~12 ns per atomic, which at 4 k waves x 5 counters was the whole launch time).
struct RegionCtl {
    // This region's static share of the pass's samples: the 64-sample chunks c with
    // c % n_regions == region, in order.  next/end count samples within that share; sample q of the
    // share is global sample ((q >> 6) * n_regions + region) * 64 + (q & 63).  Interleaving (rather
    // than contiguous slices) keeps every region's mix of cheap and expensive pixels the same.
    unsigned long long next_sample, end_sample;
    unsigned long long segments, shadow_rays, samples_done;
    uint32_t count;                               // live slots
    uint32_t half_ns;                             // bit 0: which half of the region holds them; bits 1..: how many of them carry a shadow ray
    uint32_t n_new;                               // the last n_new live slots are camera samples the last sweep started: their throughput is 1
                                                  // and their radiance 0 by definition — neither written nor read
    uint32_t invalid;                             // records ImageBlock::put would have warned about (imageblock.cpp:57-81; msk_stats::invalid_samples)
    uint32_t pad[2];
};
// A region is two halves of region_size slots.  A shading sweep reads the live paths from one half and writes the survivors
// (and the new camera samples) to the other, so nothing it writes can land on a slot it has not read yet, in whatever order
// it reads and wherever it writes.  That freedom is used to GROUP the survivors: those with a pending shadow ray are packed
// upwards from the front of the half, the others (and the new samples) downwards from its end.  Live slot c (0 <= c < count)
// is slot c for c < ns and slot region_size - 1 - (c - ns) otherwise: "has a shadow ray" is c < ns — no flag to carry, the
// shadow direction / contribution arrays are only written and read for the slots that need them, and k_trace's chunks are
// shadow-and-extension or extension-only as a whole instead of every chunk walking the tree twice with half of its lanes.
// (Slot indices are 32-bit: a pool of 2^32 slots would be 650 GB of state.)
struct RegionView {
    uint32_t base, n, ns, last;                   // last = region_size - 1
    MSK_DEV uint32_t slot(uint32_t c) const { return base + (c < ns ? c : last - (c - ns)); }
};
MSK_DEV RegionView region_view(uint32_t region, uint32_t region_size, uint32_t count, uint32_t half_ns) {
    RegionView v;
    v.base = (region * 2u + (half_ns & 1u)) * region_size; v.n = count; v.ns = half_ns >> 1; v.last = region_size - 1u;
    return v;
}
struct Ctrl {                                     // written by k_reduce_ctl, read by the host
    unsigned long long live, remaining, segments, shadow_rays, samples_done, invalid;
};

struct PassParams {
    uint64_t seed;
    uint32_t spp_owned, sample_first, sample_stride;
    int32_t rr_depth, max_depth, hide_emitters;
    const uint4 *pix_table;       // pass pixel j -> {film index y*W+x, x | y << 16 inside its block, block off_x - border, off_y - border (int bits)}
    const uint32_t *pix_to_j;     // film index -> pass pixel j (pixels of this pass only)
    float4 *rec_a;                // per sample {X,Y,Z,pos.x}; the record of (j, si) is j * spp_owned + si: [block][pixel][sample], a pixel's
    float *rec_b;                 // per sample pos.y           samples contiguous (the film replay streams them in that order)
    uint32_t packed;              // 1: the records carry the sample's filter weights instead of its position (SampleWeights below):
                                  //    rec_a.w = x word, rec_b = y word
    uint32_t region_size, n_regions;      // n_regions: all regions of the pass (the samples' static partition is over all of them)
    uint32_t region_first, region_count;  // the regions this launch covers (the pool's halves run on two streams)
    uint32_t sort_scratch;        // 1: the shading launch has LDS for the material sort (3 bytes per slot of a region and wave)
    uint32_t trace_split;         // waves per region in k_trace (each takes every trace_split-th chunk); shading is one wave per region
    RegionCtl *regions;
    uint32_t *stack_ovf;          // traversal-stack overflow (LaneStack), (stack_total - stack_entries) x lanes words, or nullptr
    float4 *aov_rgb;              // nullptr, or per sample {R,G,B,pos.x} of the nested path integrator (aov.cpp:124-141)
    // an "aov" render's primary-hit record groups (AovParams::rec, written by k_aov_primary while the sample's path is at depth 1):
    // the record of a finished sample reads them back for ImageBlock::put's validity test, which covers EVERY channel of the block
    uint32_t aov_groups;
    const float4 *aov_rec[8 /* MSK_MAX_AOV_GROUPS */];
};

// AOV channels read off the primary hit (integrators/aov.cpp:95-122), three per record group
#define MSK_MAX_AOV_GROUPS 8
struct AovParams {
    uint32_t n_groups;
    float4 *rec[MSK_MAX_AOV_GROUPS];       // per sample {a, b, c, pos.x or the x weight word}, same indexing as PassParams::rec_a
    uint32_t code[MSK_MAX_AOV_GROUPS];     // three 8-bit selectors: 0 none, 1 t, 2-4 p, 5-6 uv, 7-9 n, 10-12 sh.n
};
struct FilmOut { float *film; int32_t stride; int32_t ch[5]; };   // block channel c -> film channel ch[c] (or -1)

// Streaming accesses (cache policy).  The path state is streamed: a line is written by one kernel and read once, by the next
// kernel (ray_o, ray_d, sh, hit) or a whole iteration — 600 MB of other state — later (id, wl, thr, res, contrib).  The second
// kind, the sweep's last read of ray_d / hit and the sample records (read by the film replay at the end of the pass) use
// non-temporal loads and stores: they do not displace, in L2 and the memory-side cache, the lines the next kernel is about
// to read.  MSK_NT is how far that goes: 1 the iteration-distance arrays, 2 (default) + the sweep's loads of ray_d / hit and
// the record stores, 3 + the traversal's loads of ray_o / sh, 4 + the stores of ray_o / ray_d / sh / hit.  Measured (bench
// step, one box): 41.0 / 39.7 / 38.6 / 40.3 / 42.4 ms for 0 … 4 — what the next kernel reads must stay cacheable.
#ifndef MSK_NT
#define MSK_NT 2
#endif
typedef float msk_v4f __attribute__((ext_vector_type(4)));
typedef uint32_t msk_v4u __attribute__((ext_vector_type(4)));
template <int LEVEL> MSK_DEV void st4(float4 *p, float4 v) {
    if (MSK_NT >= LEVEL) { msk_v4f x = {v.x, v.y, v.z, v.w}; __builtin_nontemporal_store(x, (msk_v4f *) p); } else *p = v;
}
template <int LEVEL> MSK_DEV void st4(uint4 *p, uint4 v) {
    if (MSK_NT >= LEVEL) { msk_v4u x = {v.x, v.y, v.z, v.w}; __builtin_nontemporal_store(x, (msk_v4u *) p); } else *p = v;
}
template <int LEVEL> MSK_DEV float4 ld4(const float4 *p) {
    if (MSK_NT >= LEVEL) { const msk_v4f x = __builtin_nontemporal_load((const msk_v4f *) p); return make_float4(x.x, x.y, x.z, x.w); }
    return *p;
}
template <int LEVEL> MSK_DEV uint4 ld4(const uint4 *p) {
    if (MSK_NT >= LEVEL) { const msk_v4u x = __builtin_nontemporal_load((const msk_v4u *) p); return make_uint4(x.x, x.y, x.z, x.w); }
    return *p;
}
typedef uint32_t msk_v2u __attribute__((ext_vector_type(2)));
template <int LEVEL> MSK_DEV void st2(uint2 *p, uint2 v) {
    if (MSK_NT >= LEVEL) { msk_v2u x = {v.x, v.y}; __builtin_nontemporal_store(x, (msk_v2u *) p); } else *p = v;
}
template <int LEVEL> MSK_DEV uint2 ld2(const uint2 *p) {
    if (MSK_NT >= LEVEL) { const msk_v2u x = __builtin_nontemporal_load((const msk_v2u *) p); return make_uint2(x.x, x.y); }
    return *p;
}

// ------------------------------------------------------------------------------------------
// traversal
// ------------------------------------------------------------------------------------------
// Lanes of one wave exchanging data through LDS: the compiler may otherwise run one side of a divergent region past the
// other side's LDS accesses (measured, see the k_trace note).
MSK_DEV void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
MSK_DEV float xor_sign(float a, uint32_t s) { return __uint_as_float(__float_as_uint(a) ^ s); }
MSK_DEV float max_raw(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
MSK_DEV float min_raw(float a, float b) { float r; asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }

// Embree 3 Moeller-Trumbore, restated (see oracle/oracle.cpp header for the derivation);
// tmax is the ray's ORIGINAL far bound.
// + oracle D10: the hit point o + t d of an accepted hit must lie in the triangle's own padded bounding box (bounds[0..1],
// precomputed with the oracle's arithmetic): acceptance is then a property of the ray and the triangle, not of the tree —
// Moeller-Trumbore alone accepts points well outside sliver triangles and, for near-parallel rays, at meaningless t.
// The bounds are read only for hits the test above accepted.
// The test in two halves (tri_mt runs one after the other).
// tri_candidate: everything up to the distance-range test — what almost every test of a leaf fails in; T, U, V still carry |den|
struct TriCand { float T, U, V, den; };
MSK_DEV bool tri_candidate(float4 q0, float4 q1, float4 q2, float4 q3, f3 o, f3 d, float tmin, float tmax, TriCand &c) {
    const f3 v0 = mk3(q0.x, q0.y, q0.z), e1 = mk3(q1.x, q1.y, q1.z), e2 = mk3(q2.x, q2.y, q2.z),
             ng = mk3(q3.x, q3.y, q3.z);
    const f3 C = v0 - o;
    const f3 R = cross(C, d);
    const float den = dot(ng, d);
    const float abs_den = fabsf(den);
    const uint32_t sgn = __float_as_uint(den) & 0x80000000u;
    const float U = xor_sign(dot(R, e2), sgn);
    const float V = xor_sign(dot(R, e1), sgn);
    if (!(den != 0.f && U >= 0.f && V >= 0.f && U + V <= abs_den)) return false;
    const float T = xor_sign(dot(ng, C), sgn);
    if (!(abs_den * tmin < T && T <= abs_den * tmax)) return false;
    c.T = T; c.U = U; c.V = V; c.den = den;
    return true;
}
// tri_hit_point: the division; *p = the hit point o + t d the bounds predicate looks at
MSK_DEV void tri_hit_point(const TriCand &c, f3 o, f3 d, float *t, float *u, float *v, f3 *p) {
    const float rcp = 1.f / fabsf(c.den);
    *t = c.T * rcp;
    *u = fmin_std(c.U * rcp, 1.f);
    *v = fmin_std(c.V * rcp, 1.f);
    *p = mk3(o.x + *t * d.x, o.y + *t * d.y, o.z + *t * d.z);
}
// tri_mt: the Moeller-Trumbore part; *p = the hit point o + t d the bounds predicate looks at
MSK_DEV bool tri_mt(float4 q0, float4 q1, float4 q2, float4 q3, f3 o, f3 d, float tmin, float tmax, float *t, float *u, float *v, f3 *p) {
    TriCand c;
    if (!tri_candidate(q0, q1, q2, q3, o, d, tmin, tmax, c)) return false;
    tri_hit_point(c, o, d, t, u, v, p);
    return true;
}
// the D10 predicate of a tree in HBM/L2: the hit point inside the triangle's padded bounds, recomputed from the record (see tri_test)
MSK_DEV bool tri_in_bounds(float4 q0, float4 q1, float4 q2, f3 p, float pad) {
    const f3 v0 = mk3(q0.x, q0.y, q0.z), e1 = mk3(q1.x, q1.y, q1.z), e2 = mk3(q2.x, q2.y, q2.z);
    const float px = p.x, py = p.y, pz = p.z;
    const f3 w1 = v0 - e1, w2 = v0 + e2;
    const float lox = fminf(fminf(v0.x, w1.x), w2.x) - pad, hix = fmaxf(fmaxf(v0.x, w1.x), w2.x) + pad;
    const float loy = fminf(fminf(v0.y, w1.y), w2.y) - pad, hiy = fmaxf(fmaxf(v0.y, w1.y), w2.y) + pad;
    const float loz = fminf(fminf(v0.z, w1.z), w2.z) - pad, hiz = fmaxf(fmaxf(v0.z, w1.z), w2.z) + pad;
    return (px >= lox) & (px <= hix) & (py >= loy) & (py <= hiy) & (pz >= loz) & (pz <= hiz);
}
MSK_DEV bool tri_test(float4 q0, float4 q1, float4 q2, float4 q3, f3 o, f3 d, float tmin, float tmax,
                      float *t, float *u, float *v, const float4 *bounds, float pad) {
    f3 p;
    if (!tri_mt(q0, q1, q2, q3, o, d, tmin, tmax, t, u, v, &p)) return false;
    const float px = p.x, py = p.y, pz = p.z;
    if (bounds) {                           // LDS-resident scene: precomputed, two LDS reads
        const float4 lo = bounds[0], hi = bounds[1];
        return px >= lo.x && px <= hi.x && py >= lo.y && py <= hi.y && pz >= lo.z && pz <= hi.z;
    }
    // tree in HBM/L2: the same bounds from the record, no extra fetch.  v_min3 / v_max3 instead of the host's std::min / std::max
    // (a compare + select pair each, 6.3 against 4.2 SIMD cycles): the two differ only in the sign of a zero result, which the
    // -/+ pad that follows erases (pad > 0; with pad == 0 the comparison against -0 is the one against +0).  The six comparisons
    // are combined without branches (as `&&` the compiler nests six exec regions: 19 VALU and 11 SALU instructions more per step).
    return tri_in_bounds(q0, q1, q2, p, pad);
}

// Reciprocal direction of the slab test: v_rcp_f32 (1 ulp) instead of an IEEE division (11 instructions each).  The slab
// test only culls, behind boxes padded by 1e-5 of the scene's scale — two orders of magnitude more than this error and the
// other roundings of the slab arithmetic together (DESIGN.md §3 rule 4).
MSK_DEV f3 slab_idir(f3 d) {
    return mk3(fminf(fmaxf(__builtin_amdgcn_rcpf(d.x), -1e25f), 1e25f), fminf(fmaxf(__builtin_amdgcn_rcpf(d.y), -1e25f), 1e25f),
               fminf(fmaxf(__builtin_amdgcn_rcpf(d.z), -1e25f), 1e25f));
}
// conservative slab test, fma form t = b*idir - o*idir with idir clamped to +-1e25 by the caller
// (no infinities -> no NaNs; an axis-parallel ray sees +-huge instead of +-inf)
MSK_DEV bool box_test(float lox, float loy, float loz, float hix, float hiy, float hiz, f3 idir, f3 oi,
                      float tmin, float tcur, float *tnear) {
    float ax = __fmaf_rn(lox, idir.x, -oi.x), bx = __fmaf_rn(hix, idir.x, -oi.x);
    float ay = __fmaf_rn(loy, idir.y, -oi.y), by = __fmaf_rn(hiy, idir.y, -oi.y);
    float az = __fmaf_rn(loz, idir.z, -oi.z), bz = __fmaf_rn(hiz, idir.z, -oi.z);
    float t0 = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fmaxf(fminf(az, bz), tmin));
    float t1 = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fminf(fmaxf(az, bz), tcur));
    *tnear = t0;
    return t0 <= t1 * 1.0000004f;
}

// Measured and rejected: the two children's slabs as v_pk_fma_f32 pairs (cbox trace +6 %, 70 k-triangle scene +6 %: the
// packed op saves no issue cycles and costs register shuffles and 6 VGPRs = one wave of occupancy in k_trace<0>).

// A lane's traversal stack: the first `cap` entries in LDS (stride MSK_BLOCK), deeper ones in an HBM overflow array
// (stride = number of lanes of the launch).  Scenes staged in LDS (OVF = false) never overflow: their whole stack is LDS.
// For trees in HBM this bounds the LDS a block needs whatever the tree depth; the overflow is touched by few rays.
template <bool OVF>
struct LaneStack {
    uint32_t *lds; uint32_t *ovf; int cap; size_t stride;
    uint32_t *scratch = nullptr;        // OVF kernels: 4 words of LDS per lane (node4_step's child references), else unused
    MSK_DEV void push(int &sp, uint32_t v) const {
        if (!OVF || sp < cap) lds[sp * MSK_BLOCK] = v; else ovf[(size_t) (sp - cap) * stride] = v;
        sp += 1;
    }
    MSK_DEV uint32_t pop(int &sp) const {
        sp -= 1;
        return (!OVF || sp < cap) ? lds[sp * MSK_BLOCK] : ovf[(size_t) (sp - cap) * stride];
    }
};

// ANY: returns true on the first accepted triangle.  Closest: keeps (t, prim)-minimal hit; the first accepted hit is kept
// whatever its t (it can exceed tmax by an ulp: the test is T <= |den| tmax, then t = T / |den| — as in Embree, and
// scene.cpp:234 only asks tfar != maxt).
// nodes/tris may point into LDS or HBM.  stack: this lane's LDS stack, stride MSK_BLOCK.
// "while-while" form: the inner loop walks inner nodes until the lane holds a leaf (or runs out of
// work), then the wave tests leaf triangles together — lanes at inner nodes do not sit through
// other lanes' triangle tests one node at a time.
template <bool ANY, bool OVF>
MSK_DEV bool traverse(const float4 *__restrict__ nodes, const float4 *__restrict__ tris, float tri_pad, uint32_t root_ref,
                      uint32_t n_tris, f3 o, f3 d, float tmin, float tmax, const LaneStack<OVF> &stack, float *best_t, float *best_u,
                      float *best_v, uint32_t *best_prim) {
    float bt = tmax, bu = 0.f, bv = 0.f;
    uint32_t bp = MSK_NO_PRIM;
    *best_t = bt; *best_u = bu; *best_v = bv; *best_prim = bp;
    if (n_tris == 0) return false;
    const f3 idir = slab_idir(d);
    const f3 oi = mk3(o.x * idir.x, o.y * idir.y, o.z * idir.z);
    int sp = 0;
    uint32_t cur = root_ref;
    const uint32_t DONE = 0xffffffffu;      // not a valid ref: a leaf ref never has all of its count bits and index bits set
    while (cur != DONE) {
        // ---- inner nodes
        while (!(cur & MSK_LEAF_BIT)) {
            const float4 *n = nodes + (size_t) cur * 4;
            const float4 a = n[0], b = n[1], c = n[2], m = n[3];
            float t0, t1;
            const bool h0 = box_test(a.x, a.z, b.x, b.z, c.x, c.z, idir, oi, tmin, bt, &t0);
            const bool h1 = box_test(a.y, a.w, b.y, b.w, c.y, c.w, idir, oi, tmin, bt, &t1);
            const uint32_t c0 = __float_as_uint(m.x), c1 = __float_as_uint(m.y);
            if (h0 && h1) {
                const bool swap = t1 < t0;            // nearer child first, the other one on the stack
                cur = swap ? c1 : c0;
                stack.push(sp, swap ? c0 : c1);
            } else if (h0) { cur = c0; }
            else if (h1) { cur = c1; }
            else if (sp > 0) { cur = stack.pop(sp); }
            else { cur = DONE; break; }
        }
        if (cur == DONE) break;
        // ---- leaf
        const uint32_t first = (cur & 0x7fffffffu) >> 5, cnt = cur & 31u;
        for (uint32_t i = 0; i < cnt; ++i) {
            const float4 *q = tris + (size_t) (first + i) * (OVF ? 4 : 6);      // LDS copies carry their bounds: 6 float4
            const float4 q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3];
            float t, u, v;
            if (tri_test(q0, q1, q2, q3, o, d, tmin, tmax, &t, &u, &v, OVF ? nullptr : q + 4, tri_pad)) {
                if (ANY) return true;
                const uint32_t prim = __float_as_uint(q0.w);
                if (bp == MSK_NO_PRIM || t < bt || (t == bt && (prim & MSK_PRIM_ID) < (bp & MSK_PRIM_ID))) { bt = t; bu = u; bv = v; bp = prim; }
            }
        }
        if (sp > 0) { cur = stack.pop(sp); } else cur = DONE;
    }
    *best_t = bt; *best_u = bu; *best_v = bv; *best_prim = bp;
    return false;
}


// ------------------------------------------------------------------------------------------
// One visit of a 4-wide node in HBM/L2 (msk_bvh.h: collapse4), written against the measured issue costs of gfx950's VALU
// (tools/micro/valu_ops.hip: v_fma / v_mul / v_add / v_sub / v_and / v_or / v_xor / v_add_u32 / v_ashrrev take ~2.3-3 cycles of
// a SIMD per wave64 instruction; v_min / v_max / v_max3 / v_cmp / v_cndmask_e64 / v_lshl / v_and_or / v_min_u32 ~4.1) — the
// traversal kernels are bound by VALU issue, so the step is built from few and cheap instructions:
//   * near / far planes are picked by ADDRESS: the ray holds, per axis, the byte offset of its near plane quadruple inside
//     the node (0 or 48, by the sign of the reciprocal direction: Sel4); the far one is that offset ^ 48.  24 v_min / v_max of
//     the order-free slab test are gone, six loads go through a buffer resource with 32-bit per-lane offsets;
//   * an unused slot holds an inverted box (lo = +3e38, hi = -3e38: collapse4), so it misses like any other box;
//   * a miss becomes an all-ones key through the sign of t1 - t0 (v_sub, v_ashrrev, v_or), a hit the bits of its entry
//     distance with the slot number in the two low bits; four keys are sorted by five v_min_u32 / v_max_u32 pairs;
//   * the children's references go to four words of LDS per lane and come back by index (ds_read_b32 at the key's low bits).
// Only culling and visiting ORDER are involved (hit selection is by (t, prim)): same hits as every other traversal.
// ------------------------------------------------------------------------------------------
struct Sel4 { uint32_t kx, ky, kz; };          // byte offsets of the near-plane quadruples: {0|48, 16 + (0|48), 32 + (0|48)}
MSK_DEV Sel4 make_sel4(f3 idir) {
    Sel4 s;
    s.kx = idir.x < 0.f ? 48u : 0u; s.ky = idir.y < 0.f ? 64u : 16u; s.kz = idir.z < 0.f ? 80u : 32u;
    return s;
}
typedef uint32_t msk_u4 __attribute__((ext_vector_type(4)));
MSK_DEV __amdgpu_buffer_rsrc_t nodes4_rsrc(const DeviceScene &sc) {
    return __builtin_amdgcn_make_buffer_rsrc((void *) sc.nodes4, 0, sc.n_nodes4 * 128u, 0x00020000);
}
// returns the next node / leaf reference (0xffffffff: nothing left), pushes the other hit children farthest first
template <bool OVF>
MSK_DEV uint32_t node4_step(__amdgpu_buffer_rsrc_t rsrc, uint32_t node, const Sel4 &sel, f3 idir, f3 oi, float tmin, float tcur,
                            const LaneStack<OVF> &stack, int &sp) {
    const uint32_t base = node << 7;
    const uint32_t ax = base + sel.kx, ay = base + sel.ky, az = base + sel.kz;
    const msk_u4 nx = __builtin_amdgcn_raw_buffer_load_b128(rsrc, ax, 0, 0), fx = __builtin_amdgcn_raw_buffer_load_b128(rsrc, ax ^ 48u, 0, 0);
    const msk_u4 ny = __builtin_amdgcn_raw_buffer_load_b128(rsrc, ay, 0, 0), fy = __builtin_amdgcn_raw_buffer_load_b128(rsrc, ay ^ 80u, 0, 0);   // 16 <-> 64
    const msk_u4 nz = __builtin_amdgcn_raw_buffer_load_b128(rsrc, az, 0, 0), fz = __builtin_amdgcn_raw_buffer_load_b128(rsrc, az ^ 112u, 0, 0);  // 32 <-> 80
    const msk_u4 rf = __builtin_amdgcn_raw_buffer_load_b128(rsrc, base + 96u, 0, 0);
    *(msk_u4 *) stack.scratch = rf;
    uint32_t key[4];
#define MSK_CHILD(I, C) {                                                                                                        \
        const float t0 = fmaxf(fmaxf(fmaxf(__fmaf_rn(__uint_as_float(nx.C), idir.x, -oi.x), __fmaf_rn(__uint_as_float(ny.C), idir.y, -oi.y)), \
                                     __fmaf_rn(__uint_as_float(nz.C), idir.z, -oi.z)), tmin);                                   \
        const float t1 = fminf(fminf(fminf(__fmaf_rn(__uint_as_float(fx.C), idir.x, -oi.x), __fmaf_rn(__uint_as_float(fy.C), idir.y, -oi.y)), \
                                     __fmaf_rn(__uint_as_float(fz.C), idir.z, -oi.z)), tcur);                                   \
        const uint32_t miss = (uint32_t) ((int32_t) __float_as_uint(t1 * 1.0000004f - t0) >> 31);                              \
        key[I] = ((__float_as_uint(t0) & 0x7ffffff0u) | (uint32_t) ((I) << 2)) | miss; }
    MSK_CHILD(0, x) MSK_CHILD(1, y) MSK_CHILD(2, z) MSK_CHILD(3, w)
#undef MSK_CHILD
#ifdef MSK_EXP_LOAD     /* what-if builds (tools/build_variant.sh): one more 16-byte load of the node's own line per visit (its zero padding) */
    { const msk_u4 ex_ = __builtin_amdgcn_raw_buffer_load_b128(rsrc, base + 112u, 0, 0); key[0] |= ex_.x; }
#endif
#ifdef MSK_EXP_VALU     /* what-if builds: MSK_EXP_VALU more dependent VALU instructions per visit */
    { float z_ = tmin; for (int e_ = 0; e_ < MSK_EXP_VALU; ++e_) z_ = __fmaf_rn(z_, idir.x, oi.x); key[0] |= z_ == 12345.678f ? 16u : 0u; }
#endif
#define MSK_CSWAPU(a, b) { const uint32_t lo_ = a < b ? a : b; b = a < b ? b : a; a = lo_; }
    MSK_CSWAPU(key[0], key[1]) MSK_CSWAPU(key[2], key[3]) MSK_CSWAPU(key[0], key[2]) MSK_CSWAPU(key[1], key[3]) MSK_CSWAPU(key[1], key[2])
#undef MSK_CSWAPU
    const uint32_t NONE = 0xffffffffu;
    const char *sc4 = (const char *) stack.scratch;
    if (key[0] == NONE) return sp > 0 ? stack.pop(sp) : NONE;
    if (key[1] != NONE) {
        if (key[2] != NONE) {
            if (key[3] != NONE) stack.push(sp, *(const uint32_t *) (sc4 + (key[3] & 12u)));
            stack.push(sp, *(const uint32_t *) (sc4 + (key[2] & 12u)));
        }
        stack.push(sp, *(const uint32_t *) (sc4 + (key[1] & 12u)));
    }
    return *(const uint32_t *) (sc4 + (key[0] & 12u));
}


// The visit above over the QUANTISED twin of the node (Built::nodes4q, 64 bytes): four 16-byte loads instead of seven.  The
// traversal of a tree in HBM/L2 is bound by the texture addresser — about one cycle per lane and load instruction whatever its
// width (measured: one more 16-byte load of the node's own line per visit +7 % trace time, 32 more dependent VALU instructions
// +1.4 %) — so bytes per visit are what counts, and the VALU has room for the decoding: t = q * (scale * idir) + (origin * idir -
// o * idir) for the plane byte q (v_cvt_f32_ubyte, v_fma).  Near / far plane words are picked by the ray's sign masks (v_bfi).
// The decoded boxes contain the padded full-precision ones (checked when they are built), so this, too, only culls.
MSK_DEV uint32_t bfi_b32(uint32_t mask, uint32_t a, uint32_t b) {          // (mask & a) | (~mask & b)
    uint32_t r;
    asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(r) : "v"(mask), "v"(a), "v"(b));
    return r;
}
struct Sel4q { uint32_t mx, my, mz; };          // all ones where the reciprocal direction is negative (near plane = hi)
MSK_DEV Sel4q make_sel4q(f3 idir) {
    Sel4q s;
    s.mx = idir.x < 0.f ? 0xffffffffu : 0u; s.my = idir.y < 0.f ? 0xffffffffu : 0u; s.mz = idir.z < 0.f ? 0xffffffffu : 0u;
    return s;
}
MSK_DEV __amdgpu_buffer_rsrc_t nodes4q_rsrc(const DeviceScene &sc) {
    return __builtin_amdgcn_make_buffer_rsrc((void *) sc.nodes4q, 0, sc.n_nodes4 * 64u, 0x00020000);
}
template <bool OVF>
MSK_DEV uint32_t node4q_step(__amdgpu_buffer_rsrc_t rsrc, uint32_t node, const Sel4q &sel, f3 idir, f3 oi, float tmin, float tcur,
                             const LaneStack<OVF> &stack, int &sp) {
    const uint32_t base = node << 6;
    const msk_u4 h0 = __builtin_amdgcn_raw_buffer_load_b128(rsrc, base, 0, 0), h1 = __builtin_amdgcn_raw_buffer_load_b128(rsrc, base + 16u, 0, 0);
    const msk_u4 h2 = __builtin_amdgcn_raw_buffer_load_b128(rsrc, base + 32u, 0, 0), rf = __builtin_amdgcn_raw_buffer_load_b128(rsrc, base + 48u, 0, 0);
    *(msk_u4 *) stack.scratch = rf;
    // h0 = origin.xyz, scale.x   h1 = scale.y, scale.z, lo.x[4], lo.y[4]   h2 = lo.z[4], hi.x[4], hi.y[4], hi.z[4]
    const float ax = __uint_as_float(h0.w) * idir.x, ay = __uint_as_float(h1.x) * idir.y, az = __uint_as_float(h1.y) * idir.z;
    const float bx = __fmaf_rn(__uint_as_float(h0.x), idir.x, -oi.x), by = __fmaf_rn(__uint_as_float(h0.y), idir.y, -oi.y),
                bz = __fmaf_rn(__uint_as_float(h0.z), idir.z, -oi.z);
    // (v_bfi_b32 by hand: the compiler's v_and + v_and_or pair costs 6.5 SIMD cycles against 4.2, tools/micro/valu_ops.hip)
    const uint32_t nx = bfi_b32(sel.mx, h2.y, h1.z), fx = bfi_b32(sel.mx, h1.z, h2.y);
    const uint32_t ny = bfi_b32(sel.my, h2.z, h1.w), fy = bfi_b32(sel.my, h1.w, h2.z);
    const uint32_t nz = bfi_b32(sel.mz, h2.w, h2.x), fz = bfi_b32(sel.mz, h2.x, h2.w);
    uint32_t key[4];
#define MSK_QB(w, k) ((float) (((w) >> (8 * (k))) & 0xffu))
#define MSK_CHILD(I) {                                                                                                            \
        const float t0 = fmaxf(fmaxf(fmaxf(__fmaf_rn(MSK_QB(nx, I), ax, bx), __fmaf_rn(MSK_QB(ny, I), ay, by)), __fmaf_rn(MSK_QB(nz, I), az, bz)), tmin); \
        const float t1 = fminf(fminf(fminf(__fmaf_rn(MSK_QB(fx, I), ax, bx), __fmaf_rn(MSK_QB(fy, I), ay, by)), __fmaf_rn(MSK_QB(fz, I), az, bz)), tcur); \
        const uint32_t miss = (uint32_t) ((int32_t) __float_as_uint(__fmaf_rn(t1, 1.0000004f, -t0)) >> 31);                      \
        key[I] = ((__float_as_uint(t0) & 0x7ffffff0u) | (uint32_t) ((I) << 2)) | miss; }
    MSK_CHILD(0) MSK_CHILD(1) MSK_CHILD(2) MSK_CHILD(3)
#undef MSK_CHILD
#undef MSK_QB
#define MSK_CSWAPU(a, b) { const uint32_t lo_ = a < b ? a : b; b = a < b ? b : a; a = lo_; }
    MSK_CSWAPU(key[0], key[1]) MSK_CSWAPU(key[2], key[3]) MSK_CSWAPU(key[0], key[2]) MSK_CSWAPU(key[1], key[3]) MSK_CSWAPU(key[1], key[2])
#undef MSK_CSWAPU
    const uint32_t NONE = 0xffffffffu;
    const char *sc4 = (const char *) stack.scratch;
    if (key[0] == NONE) return sp > 0 ? stack.pop(sp) : NONE;
    if (key[1] != NONE) {
        if (key[2] != NONE) {
            if (key[3] != NONE) stack.push(sp, *(const uint32_t *) (sc4 + (key[3] & 12u)));
            stack.push(sp, *(const uint32_t *) (sc4 + (key[2] & 12u)));
        }
        stack.push(sp, *(const uint32_t *) (sc4 + (key[1] & 12u)));
    }
    return *(const uint32_t *) (sc4 + (key[0] & 12u));
}

// The same visit over the HALF-FLOAT twin of the node (Built::nodes4h, 80 bytes, trace mode 6; round 5): the child boxes' planes
// as fp16 offsets from the node's origin in units of one power-of-two scale, read straight into the slab test's fma by
// v_fma_mix_f32 (an fp16 operand costs nothing extra: 4.1 SIMD cycles against v_cvt_f32_ubyte + v_fma = 7.0 for a byte plane)
// — at the price of a fifth 16-byte load per visit and twelve v_bfi instead of six (a plane quadruple is two dwords).  The
// planes are rounded outwards (lo down, hi up; a non-zero hi never below the smallest normal fp16, so that a flushed
// subnormal cannot shrink a box) around the padded boxes and checked in exact arithmetic when they are built; eleven bits of
// mantissa make the boxes tighter than the byte grid's.  Measured: DESIGN.md section 9 row 3, round 5.
//   q0 = origin.xyz, scale   q1 = lo.x[4], lo.y[4] (two fp16 per dword)   q2 = lo.z[4], hi.x[4]   q3 = hi.y[4], hi.z[4]   q4 = 4 refs
typedef _Float16 msk_h2 __attribute__((ext_vector_type(2)));
MSK_DEV float half_lo(uint32_t w) { return (float) __builtin_bit_cast(msk_h2, w)[0]; }
MSK_DEV float half_hi(uint32_t w) { return (float) __builtin_bit_cast(msk_h2, w)[1]; }
MSK_DEV __amdgpu_buffer_rsrc_t nodes4h_rsrc(const DeviceScene &sc) {
    return __builtin_amdgcn_make_buffer_rsrc((void *) sc.nodes4q, 0, sc.n_nodes4 * 80u, 0x00020000);
}
template <bool OVF>
MSK_DEV uint32_t node4h_step(__amdgpu_buffer_rsrc_t rsrc, uint32_t node, const Sel4q &sel, f3 idir, f3 oi, float tmin, float tcur,
                             const LaneStack<OVF> &stack, int &sp) {
    const uint32_t base = node;             // an inner node's reference IS its byte offset in this form (msk_bvh.h: quantise_h)
    const msk_u4 h0 = __builtin_amdgcn_raw_buffer_load_b128(rsrc, base, 0, 0), h1 = __builtin_amdgcn_raw_buffer_load_b128(rsrc, base + 16u, 0, 0);
    const msk_u4 h2 = __builtin_amdgcn_raw_buffer_load_b128(rsrc, base + 32u, 0, 0), h3 = __builtin_amdgcn_raw_buffer_load_b128(rsrc, base + 48u, 0, 0);
    const msk_u4 rf = __builtin_amdgcn_raw_buffer_load_b128(rsrc, base + 64u, 0, 0);
    *(msk_u4 *) stack.scratch = rf;
    const float sc_ = __uint_as_float(h0.w);
    const float ax = sc_ * idir.x, ay = sc_ * idir.y, az = sc_ * idir.z;
    const float bx = __fmaf_rn(__uint_as_float(h0.x), idir.x, -oi.x), by = __fmaf_rn(__uint_as_float(h0.y), idir.y, -oi.y),
                bz = __fmaf_rn(__uint_as_float(h0.z), idir.z, -oi.z);
    // lo.x = h1.xy  lo.y = h1.zw  lo.z = h2.xy  hi.x = h2.zw  hi.y = h3.xy  hi.z = h3.zw
    const uint32_t nxa = bfi_b32(sel.mx, h2.z, h1.x), nxb = bfi_b32(sel.mx, h2.w, h1.y), fxa = bfi_b32(sel.mx, h1.x, h2.z), fxb = bfi_b32(sel.mx, h1.y, h2.w);
    const uint32_t nya = bfi_b32(sel.my, h3.x, h1.z), nyb = bfi_b32(sel.my, h3.y, h1.w), fya = bfi_b32(sel.my, h1.z, h3.x), fyb = bfi_b32(sel.my, h1.w, h3.y);
    const uint32_t nza = bfi_b32(sel.mz, h3.z, h2.x), nzb = bfi_b32(sel.mz, h3.w, h2.y), fza = bfi_b32(sel.mz, h2.x, h3.z), fzb = bfi_b32(sel.mz, h2.y, h3.w);
    uint32_t key[4];
#define MSK_CHILD(I, H, NX, NY, NZ, FX, FY, FZ) {                                                                               \
        const float t0 = fmaxf(fmaxf(fmaxf(__fmaf_rn(H(NX), ax, bx), __fmaf_rn(H(NY), ay, by)), __fmaf_rn(H(NZ), az, bz)), tmin); \
        const float t1 = fminf(fminf(fminf(__fmaf_rn(H(FX), ax, bx), __fmaf_rn(H(FY), ay, by)), __fmaf_rn(H(FZ), az, bz)), tcur); \
        const uint32_t miss = (uint32_t) ((int32_t) __float_as_uint(__fmaf_rn(t1, 1.0000004f, -t0)) >> 31);                      \
        key[I] = ((__float_as_uint(t0) & 0x7ffffff0u) | (uint32_t) ((I) << 2)) | miss; }
    MSK_CHILD(0, half_lo, nxa, nya, nza, fxa, fya, fza) MSK_CHILD(1, half_hi, nxa, nya, nza, fxa, fya, fza)
    MSK_CHILD(2, half_lo, nxb, nyb, nzb, fxb, fyb, fzb) MSK_CHILD(3, half_hi, nxb, nyb, nzb, fxb, fyb, fzb)
#undef MSK_CHILD
#define MSK_CSWAPU(a, b) { const uint32_t lo_ = a < b ? a : b; b = a < b ? b : a; a = lo_; }
    MSK_CSWAPU(key[0], key[1]) MSK_CSWAPU(key[2], key[3]) MSK_CSWAPU(key[0], key[2]) MSK_CSWAPU(key[1], key[3]) MSK_CSWAPU(key[1], key[2])
#undef MSK_CSWAPU
    const uint32_t NONE = 0xffffffffu;
    const char *sc4 = (const char *) stack.scratch;
    if (key[0] == NONE) return sp > 0 ? stack.pop(sp) : NONE;
    if (key[1] != NONE) {
        if (key[2] != NONE) {
            if (key[3] != NONE) stack.push(sp, *(const uint32_t *) (sc4 + (key[3] & 12u)));
            stack.push(sp, *(const uint32_t *) (sc4 + (key[2] & 12u)));
        }
        stack.push(sp, *(const uint32_t *) (sc4 + (key[1] & 12u)));
    }
    return *(const uint32_t *) (sc4 + (key[0] & 12u));
}

// a triangle of a tree in HBM/L2: three loads (DeviceScene::tris3), the normal recomputed with the builder's operations
// MSK_TRI_REC: what a tree in HBM/L2 reads per triangle test.  0: 48 bytes {v0 | prim, e1 | e2.x, e2.yz}, the normal and the D10
// bounds recomputed (three loads);  1: 64 bytes with the normal (DeviceScene::tris as the builder wrote it), bounds recomputed;
// 2: 64 bytes {v0 | prim, e1 | e2.x, e2.yz | lo.xy, lo.z | hi}: the D10 bounds as the builder padded them, the normal recomputed.
// Measured (round 4, config-5 / config-3 class renders, three runs each): 0: 128.3 / 165.0 ms;  1: 127.5 / 165.1 (8 VALU instructions
// fewer per test, one load more: nothing);  2: 131.0 / 169.5 (17 fewer, one load more: +2 %) — the fourth load costs what the
// arithmetic saves or more, although the kernel's VALU is busy all the time: the default stays 0.
#ifndef MSK_TRI_REC
#define MSK_TRI_REC 0
#endif
#define MSK_TRI_REC_BYTES (MSK_TRI_REC == 0 ? 48u : 64u)
MSK_DEV __amdgpu_buffer_rsrc_t tris3_rsrc(const DeviceScene &sc) {
    return __builtin_amdgcn_make_buffer_rsrc((void *) (MSK_TRI_REC == 1 ? sc.tris : sc.tris3), 0, sc.n_tris * MSK_TRI_REC_BYTES, 0x00020000);
}
struct TriRec { float4 q0, q1, q2, q3; f3 lo, hi; };
MSK_DEV void load_tri_rec(__amdgpu_buffer_rsrc_t rsrc, uint32_t k, TriRec &r) {
    const uint32_t off = k * MSK_TRI_REC_BYTES;
    const msk_u4 a = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 0), b = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off + 16u, 0, 0),
                 c = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off + 32u, 0, 0);
    r.q0 = make_float4(__uint_as_float(a.x), __uint_as_float(a.y), __uint_as_float(a.z), __uint_as_float(a.w));
    if constexpr (MSK_TRI_REC == 1) {
        const msk_u4 e = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off + 48u, 0, 0);
        r.q1 = make_float4(__uint_as_float(b.x), __uint_as_float(b.y), __uint_as_float(b.z), 0.f);
        r.q2 = make_float4(__uint_as_float(c.x), __uint_as_float(c.y), __uint_as_float(c.z), 0.f);
        r.q3 = make_float4(__uint_as_float(e.x), __uint_as_float(e.y), __uint_as_float(e.z), 0.f);
        return;
    }
    r.q1 = make_float4(__uint_as_float(b.x), __uint_as_float(b.y), __uint_as_float(b.z), 0.f);
    r.q2 = make_float4(__uint_as_float(b.w), __uint_as_float(c.x), __uint_as_float(c.y), 0.f);
    r.q3 = make_float4(r.q2.y * r.q1.z - r.q2.z * r.q1.y, r.q2.z * r.q1.x - r.q2.x * r.q1.z, r.q2.x * r.q1.y - r.q2.y * r.q1.x, 0.f);     // Ng = e2 x e1 (msk_bvh.h: build)
    if constexpr (MSK_TRI_REC == 2) {
        const msk_u4 e = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off + 48u, 0, 0);
        r.lo = mk3(__uint_as_float(c.z), __uint_as_float(c.w), __uint_as_float(e.x));
        r.hi = mk3(__uint_as_float(e.y), __uint_as_float(e.z), __uint_as_float(e.w));
    }
}
MSK_DEV bool tri_test_rec(const TriRec &r, f3 o, f3 d, float tmin, float tmax, float *t, float *u, float *v, float pad) {
    if constexpr (MSK_TRI_REC == 2) {
        f3 p;
        if (!tri_mt(r.q0, r.q1, r.q2, r.q3, o, d, tmin, tmax, t, u, v, &p)) return false;
        return (p.x >= r.lo.x) & (p.x <= r.hi.x) & (p.y >= r.lo.y) & (p.y <= r.hi.y) & (p.z >= r.lo.z) & (p.z <= r.hi.z);
    } else {
        return tri_test(r.q0, r.q1, r.q2, r.q3, o, d, tmin, tmax, t, u, v, nullptr, pad);
    }
}
MSK_DEV void load_tri3(__amdgpu_buffer_rsrc_t rsrc, uint32_t k, float4 &q0, float4 &q1, float4 &q2, float4 &q3) {
    const uint32_t off = k * 48u;
    const msk_u4 a = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 0), b = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off + 16u, 0, 0),
                 c = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off + 32u, 0, 0);
    q0 = make_float4(__uint_as_float(a.x), __uint_as_float(a.y), __uint_as_float(a.z), __uint_as_float(a.w));
    q1 = make_float4(__uint_as_float(b.x), __uint_as_float(b.y), __uint_as_float(b.z), 0.f);
    q2 = make_float4(__uint_as_float(b.w), __uint_as_float(c.x), __uint_as_float(c.y), 0.f);
    q3 = make_float4(q2.y * q1.z - q2.z * q1.y, q2.z * q1.x - q2.x * q1.z, q2.x * q1.y - q2.y * q1.x, 0.f);     // Ng = e2 x e1 (msk_bvh.h: build)
}

// The same traversal over the 4-wide nodes of msk_bvh.h (one 128-byte line per visit, half the dependent round trips
// of the binary tree); used when the tree lives in HBM/L2.  Hit selection is by (t, prim), so the result is the binary
// tree's, bit for bit.
#define MSK_EMPTY4 0xfffffffeu
template <bool ANY, bool OVF>
MSK_DEV bool traverse4(const float4 *__restrict__ nodes, const float4 *__restrict__ tris, float tri_pad, uint32_t root_ref,
                       uint32_t n_tris, f3 o, f3 d, float tmin, float tmax, const LaneStack<OVF> &stack, float *best_t, float *best_u,
                       float *best_v, uint32_t *best_prim) {
    float bt = tmax, bu = 0.f, bv = 0.f;
    uint32_t bp = MSK_NO_PRIM;
    *best_t = bt; *best_u = bu; *best_v = bv; *best_prim = bp;
    if (n_tris == 0) return false;
    const f3 idir = slab_idir(d);
    const f3 oi = mk3(o.x * idir.x, o.y * idir.y, o.z * idir.z);
    int sp = 0;
    uint32_t cur = root_ref;
    const uint32_t DONE = 0xffffffffu;
    while (cur != DONE) {
        while (!(cur & MSK_LEAF_BIT)) {
            const float4 *n = nodes + (size_t) cur * 8;
            const float4 lx = n[0], ly = n[1], lz = n[2], hx = n[3], hy = n[4], hz = n[5], rf = n[6];
            float t0, t1, t2, t3;
            const bool h0 = box_test(lx.x, ly.x, lz.x, hx.x, hy.x, hz.x, idir, oi, tmin, bt, &t0);
            const bool h1 = box_test(lx.y, ly.y, lz.y, hx.y, hy.y, hz.y, idir, oi, tmin, bt, &t1);
            const bool h2 = box_test(lx.z, ly.z, lz.z, hx.z, hy.z, hz.z, idir, oi, tmin, bt, &t2);
            const bool h3 = box_test(lx.w, ly.w, lz.w, hx.w, hy.w, hz.w, idir, oi, tmin, bt, &t3);
            uint32_t r0 = __float_as_uint(rf.x), r1 = __float_as_uint(rf.y), r2 = __float_as_uint(rf.z), r3 = __float_as_uint(rf.w);
            // misses and empty slots sort to the end
            t0 = (h0 && r0 != MSK_EMPTY4) ? t0 : MSK_INF_F; t1 = (h1 && r1 != MSK_EMPTY4) ? t1 : MSK_INF_F;
            t2 = (h2 && r2 != MSK_EMPTY4) ? t2 : MSK_INF_F; t3 = (h3 && r3 != MSK_EMPTY4) ? t3 : MSK_INF_F;
#define MSK_CSWAP(ta, ra, tb, rb) { const bool s_ = tb < ta; const float tt_ = s_ ? tb : ta; const uint32_t rr_ = s_ ? rb : ra; \
                                    tb = s_ ? ta : tb; rb = s_ ? ra : rb; ta = tt_; ra = rr_; }
            MSK_CSWAP(t0, r0, t1, r1) MSK_CSWAP(t2, r2, t3, r3) MSK_CSWAP(t0, r0, t2, r2) MSK_CSWAP(t1, r1, t3, r3) MSK_CSWAP(t1, r1, t2, r2)
#undef MSK_CSWAP
            if (t0 != MSK_INF_F) {
                // nearest child next, the others on the stack, farthest first
                if (t3 != MSK_INF_F) { stack.push(sp, r3); }
                if (t2 != MSK_INF_F) { stack.push(sp, r2); }
                if (t1 != MSK_INF_F) { stack.push(sp, r1); }
                cur = r0;
            } else if (sp > 0) { cur = stack.pop(sp); }
            else { cur = DONE; break; }
        }
        if (cur == DONE) break;
        const uint32_t first = (cur & 0x7fffffffu) >> 5, cnt = cur & 31u;
        for (uint32_t i = 0; i < cnt; ++i) {
            const float4 *q = tris + (size_t) (first + i) * (OVF ? 4 : 6);      // LDS copies carry their bounds: 6 float4
            const float4 q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3];
            float t, u, v;
            if (tri_test(q0, q1, q2, q3, o, d, tmin, tmax, &t, &u, &v, OVF ? nullptr : q + 4, tri_pad)) {
                if (ANY) return true;
                const uint32_t prim = __float_as_uint(q0.w);
                if (bp == MSK_NO_PRIM || t < bt || (t == bt && (prim & MSK_PRIM_ID) < (bp & MSK_PRIM_ID))) { bt = t; bu = u; bv = v; bp = prim; }
            }
        }
        if (sp > 0) { cur = stack.pop(sp); } else cur = DONE;
    }
    *best_t = bt; *best_u = bu; *best_v = bv; *best_prim = bp;
    return false;
}

// The 4-wide tree in HBM/L2 (trace modes 2, 5 and 6): node4_step / node4q_step / node4h_step visits, three-load triangles.
template <bool ANY, int MODE>
MSK_DEV bool traverse4h(const DeviceScene &sc, f3 o, f3 d, float tmin, float tmax, const LaneStack<true> &stack, float *best_t, float *best_u,
                        float *best_v, uint32_t *best_prim) {
    float bt = tmax, bu = 0.f, bv = 0.f;
    uint32_t bp = MSK_NO_PRIM;
    *best_t = bt; *best_u = bu; *best_v = bv; *best_prim = bp;
    if (sc.n_tris == 0) return false;
    const f3 idir = slab_idir(d);
    const f3 oi = mk3(o.x * idir.x, o.y * idir.y, o.z * idir.z);
    const __amdgpu_buffer_rsrc_t rn = MODE == 5 ? nodes4q_rsrc(sc) : MODE == 6 ? nodes4h_rsrc(sc) : nodes4_rsrc(sc), rt = tris3_rsrc(sc);
    const Sel4 sel = make_sel4(idir);
    const Sel4q selq = make_sel4q(idir);
    int sp = 0;
    uint32_t cur = sc.root_ref4;
    const uint32_t DONE = 0xffffffffu;
    while (cur != DONE) {
        while (!(cur & MSK_LEAF_BIT)) {
            if constexpr (MODE == 5) cur = node4q_step<true>(rn, cur, selq, idir, oi, tmin, bt, stack, sp);
            else if constexpr (MODE == 6) cur = node4h_step<true>(rn, cur, selq, idir, oi, tmin, bt, stack, sp);
            else cur = node4_step<true>(rn, cur, sel, idir, oi, tmin, bt, stack, sp);
        }
        if (cur == DONE) break;
        const uint32_t first = (cur & 0x7fffffffu) >> 5, cnt = cur & 31u;
        for (uint32_t i = 0; i < cnt; ++i) {
            TriRec r;
            load_tri_rec(rt, first + i, r);
            float t, u, v;
            if (tri_test_rec(r, o, d, tmin, tmax, &t, &u, &v, sc.tri_pad)) {
                if (ANY) return true;
                const uint32_t prim = __float_as_uint(r.q0.w);
                if (bp == MSK_NO_PRIM || t < bt || (t == bt && (prim & MSK_PRIM_ID) < (bp & MSK_PRIM_ID))) { bt = t; bu = u; bv = v; bp = prim; }
            }
        }
        if (sp > 0) { cur = stack.pop(sp); } else cur = DONE;
    }
    *best_t = bt; *best_u = bu; *best_v = bv; *best_prim = bp;
    return false;
}

// ------------------------------------------------------------------------------------------
// 8-wide nodes with quantised child boxes (msk_bvh.h: Built::nodes8), for trees that live in HBM/L2: one 128-byte line and
// one dependent round trip per eight boxes, a tree a quarter of the binary one's size.  A visit moves the ray into the
// node's frame — t = q * (scale * idir) + (origin - o) * idir for a box plane stored as the byte q — picks each axis's near
// and far planes by the sign of the direction, tests the eight boxes and hands back the nearest hit child IN SLOT ORDER
// (the builder sorted the slots along the node's ordering axis; the ray walks them up or down by the sign of its direction
// there), the others go on the stack farthest first.  The boxes are conservative (rounded outwards around boxes that are
// already padded by 1e-4 of the scene diagonal), so this, too, can only cull what no triangle test would accept: the hit is
// the binary tree's, bit for bit.
// ------------------------------------------------------------------------------------------
#define MSK_NONE_REF 0xffffffffu
template <bool OVF>
MSK_DEV uint32_t node8_step(const float4 *__restrict__ nodes8, uint32_t node, f3 o, f3 d, f3 idir, float tmin, float tcur,
                            const LaneStack<OVF> &stack, int &sp) {
    const uint4 *n = (const uint4 *) (nodes8 + (size_t) node * 8);
    const uint4 h0 = n[0], h1 = n[1], q0 = n[2], q1 = n[3], q2 = n[4], r0 = n[5], r1 = n[6];
    const float ax = __uint_as_float(h1.x) * idir.x, ay = __uint_as_float(h1.y) * idir.y, az = __uint_as_float(h1.z) * idir.z;
    const float bx = (__uint_as_float(h0.x) - o.x) * idir.x, by = (__uint_as_float(h0.y) - o.y) * idir.y, bz = (__uint_as_float(h0.z) - o.z) * idir.z;
    // q0 = lo.x[0..7] lo.y[0..7]; q1 = lo.z, hi.x; q2 = hi.y, hi.z (two dwords of four bytes each)
    const bool nx = idir.x < 0.f, ny = idir.y < 0.f, nz = idir.z < 0.f;
    const uint32_t nxa = nx ? q1.z : q0.x, nxb = nx ? q1.w : q0.y, fxa = nx ? q0.x : q1.z, fxb = nx ? q0.y : q1.w;
    const uint32_t nya = ny ? q2.x : q0.z, nyb = ny ? q2.y : q0.w, fya = ny ? q0.z : q2.x, fyb = ny ? q0.w : q2.y;
    const uint32_t nza = nz ? q2.z : q1.x, nzb = nz ? q2.w : q1.y, fza = nz ? q1.x : q2.z, fzb = nz ? q1.y : q2.w;
    uint32_t mask = 0;
#define MSK_BYTE(w, k) ((float) (((w) >> (8 * (k))) & 0xffu))
#define MSK_BOX8(S, NXW, NYW, NZW, FXW, FYW, FZW, K) {                                                                     \
        const float t0 = fmaxf(fmaxf(__fmaf_rn(MSK_BYTE(NXW, K), ax, bx), __fmaf_rn(MSK_BYTE(NYW, K), ay, by)),            \
                               fmaxf(__fmaf_rn(MSK_BYTE(NZW, K), az, bz), tmin));                                          \
        const float t1 = fminf(fminf(__fmaf_rn(MSK_BYTE(FXW, K), ax, bx), __fmaf_rn(MSK_BYTE(FYW, K), ay, by)),            \
                               fminf(__fmaf_rn(MSK_BYTE(FZW, K), az, bz), tcur));                                          \
        mask |= (t0 <= t1 * 1.0000004f) ? (1u << (S)) : 0u; }
    MSK_BOX8(0, nxa, nya, nza, fxa, fya, fza, 0) MSK_BOX8(1, nxa, nya, nza, fxa, fya, fza, 1)
    MSK_BOX8(2, nxa, nya, nza, fxa, fya, fza, 2) MSK_BOX8(3, nxa, nya, nza, fxa, fya, fza, 3)
    MSK_BOX8(4, nxb, nyb, nzb, fxb, fyb, fzb, 0) MSK_BOX8(5, nxb, nyb, nzb, fxb, fyb, fzb, 1)
    MSK_BOX8(6, nxb, nyb, nzb, fxb, fyb, fzb, 2) MSK_BOX8(7, nxb, nyb, nzb, fxb, fyb, fzb, 3)
#undef MSK_BOX8
#undef MSK_BYTE
    mask &= (h0.w >> 8) & 0xffu;                       // used slots only (an unused one holds no ref)
    if (mask == 0u) return MSK_NONE_REF;
    const uint32_t axis = h0.w & 3u;
    const float da = axis == 0u ? d.x : axis == 1u ? d.y : d.z;
    // the nearest hit child in slot order is returned, the others are pushed farthest first; `pend` delays every push by one
    // so that the nearest never goes through the stack
    uint32_t pend = MSK_NONE_REF;
#define MSK_VISIT(S, R) if (mask & (1u << (S))) { if (pend != MSK_NONE_REF) stack.push(sp, pend); pend = (R); }
    if (da >= 0.f) {
        MSK_VISIT(7, r1.w) MSK_VISIT(6, r1.z) MSK_VISIT(5, r1.y) MSK_VISIT(4, r1.x) MSK_VISIT(3, r0.w) MSK_VISIT(2, r0.z) MSK_VISIT(1, r0.y) MSK_VISIT(0, r0.x)
    } else {
        MSK_VISIT(0, r0.x) MSK_VISIT(1, r0.y) MSK_VISIT(2, r0.z) MSK_VISIT(3, r0.w) MSK_VISIT(4, r1.x) MSK_VISIT(5, r1.y) MSK_VISIT(6, r1.z) MSK_VISIT(7, r1.w)
    }
#undef MSK_VISIT
    return pend;
}

// The traversal over those nodes; leaves and triangles as in traverse4.
template <bool ANY, bool OVF>
MSK_DEV bool traverse8(const float4 *__restrict__ nodes8, const float4 *__restrict__ tris, float tri_pad, uint32_t root_ref,
                       uint32_t n_tris, f3 o, f3 d, float tmin, float tmax, const LaneStack<OVF> &stack, float *best_t, float *best_u,
                       float *best_v, uint32_t *best_prim) {
    float bt = tmax, bu = 0.f, bv = 0.f;
    uint32_t bp = MSK_NO_PRIM;
    *best_t = bt; *best_u = bu; *best_v = bv; *best_prim = bp;
    if (n_tris == 0) return false;
    const f3 idir = slab_idir(d);
    int sp = 0;
    uint32_t cur = root_ref;
    const uint32_t DONE = 0xffffffffu;
    while (cur != DONE) {
        while (!(cur & MSK_LEAF_BIT)) {
            cur = node8_step<OVF>(nodes8, cur, o, d, idir, tmin, bt, stack, sp);
            if (cur == MSK_NONE_REF) { if (sp > 0) cur = stack.pop(sp); else break; }      // MSK_NONE_REF == DONE
        }
        if (cur == DONE) break;
        const uint32_t first = (cur & 0x7fffffffu) >> 5, cnt = cur & 31u;
        for (uint32_t i = 0; i < cnt; ++i) {
            const float4 *q = tris + (size_t) (first + i) * 4;
            const float4 q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3];
            float t, u, v;
            if (tri_test(q0, q1, q2, q3, o, d, tmin, tmax, &t, &u, &v, nullptr, tri_pad)) {
                if (ANY) return true;
                const uint32_t prim = __float_as_uint(q0.w);
                if (bp == MSK_NO_PRIM || t < bt || (t == bt && (prim & MSK_PRIM_ID) < (bp & MSK_PRIM_ID))) { bt = t; bu = u; bv = v; bp = prim; }
            }
        }
        if (sp > 0) { cur = stack.pop(sp); } else cur = DONE;
    }
    *best_t = bt; *best_u = bu; *best_v = bv; *best_prim = bp;
    return false;
}

struct TraceLds {
    const float4 *nodes, *tris;        // staged triangles are 6 float4 each (record + bounds), global ones 4
};

// stage nodes + triangles into dynamic LDS (all threads of the block)
MSK_DEV TraceLds stage_scene(const DeviceScene &sc, float4 *lds, bool use_lds, bool wide = false) {
    TraceLds r;
    if (!use_lds) { r.nodes = sc.nodes; r.tris = sc.tris; return r; }
    const uint32_t nn = wide ? sc.n_nodes4 * 8 : sc.n_nodes * 4, nt = sc.n_tris * 4, nb = sc.n_tris * 2;
    const float4 *src = wide ? sc.nodes4 : sc.nodes;
    for (uint32_t i = threadIdx.x; i < nn; i += blockDim.x) lds[i] = src[i];
    for (uint32_t i = threadIdx.x; i < nt; i += blockDim.x) lds[nn + (i >> 2) * 6 + (i & 3)] = sc.tris[i];
    for (uint32_t i = threadIdx.x; i < nb; i += blockDim.x) lds[nn + (i >> 1) * 6 + 4 + (i & 1)] = sc.tri_bounds[i];
    __syncthreads();
    r.nodes = lds; r.tris = lds + nn;
    return r;
}

// ------------------------------------------------------------------------------------------
// k_trace
// ------------------------------------------------------------------------------------------
// The shadow result travels in bit 31 of the hit record's prim word (1 = unoccluded); k_shade_gen
// adds the NEE contribution, so this kernel never touches the radiance arrays.
// Measured here and rejected (round 1, cbox): (i) a per-lane state machine with dynamic ray fetch —
// with 64 lanes some lane finishes a ray in almost every iteration, so the fetch/switch path runs every
// iteration (trace +60 %); (ii) a wave-local LDS counting sort of each region's rays by (origin octant,
// direction octant) plus packing of the shadow rays — 256 rays over 64 keys leave a wave incoherent, the
// pre-pass costs more than it saves (trace +5..20 %).  Note for any future wave-level LDS exchange:
// lanes communicating through LDS need a convergent `__builtin_amdgcn_wave_barrier()` + wavefront fence
// between the phases — the compiler otherwise runs one side of a divergent region past the other's writes.
// tmax of the extension ray in slot state (see PathState::ray_d)
MSK_DEV float slot_tmax(float rd_w) {
    const int m = __float_as_int(rd_w) >> 31;                 // all ones for a bounce ray (sign bit set)
    return __int_as_float((__float_as_int(rd_w) & ~m) | (0x7f800000 & m));
}
#define MSK_HIT_UNOCCLUDED 0x80000000u
#define MSK_PRIM_MASK 0x7fffffffu
// MODE 0: binary tree staged in LDS; 1: binary tree in HBM/L2; 2: 4-wide tree in HBM/L2; 3: 4-wide tree staged in LDS;
// 4: 8-wide tree with quantised boxes in HBM/L2; 5: 4-wide tree with quantised boxes in HBM/L2 (64-byte nodes: the default);
// 6: 4-wide tree with half-float boxes in HBM/L2 (80-byte nodes, MSK_QUANT_BVH=2)
#define MSK_WIDE4H(MODE) ((MODE) == 2 || (MODE) == 5 || (MODE) == 6)      /* the 4-wide trees in HBM: node4*_step + three-load triangles */
#define MSK_OVF(MODE) ((MODE) == 1 || (MODE) == 4 || MSK_WIDE4H(MODE))      /* the stack can overflow to HBM only when the tree lives there */
template <int MODE, bool ANY>
MSK_DEV bool traverse_scene(const DeviceScene &sc, const TraceLds &g, f3 o, f3 d, float tmin, float tmax, const LaneStack<MSK_OVF(MODE)> &stack,
                            float *bt, float *bu, float *bv, uint32_t *bp) {
    if constexpr (MSK_WIDE4H(MODE)) return traverse4h<ANY, MODE>(sc, o, d, tmin, tmax, stack, bt, bu, bv, bp);
    else if constexpr (MODE == 4) return traverse8<ANY, true>(sc.nodes8, g.tris, sc.tri_pad, sc.root_ref8, sc.n_tris, o, d, tmin, tmax, stack, bt, bu, bv, bp);
    else if constexpr (MODE == 3) return traverse4<ANY, false>(g.nodes, g.tris, sc.tri_pad, sc.root_ref4, sc.n_tris, o, d, tmin, tmax, stack, bt, bu, bv, bp);
    else return traverse<ANY, MSK_OVF(MODE)>(g.nodes, g.tris, sc.tri_pad, sc.root_ref, sc.n_tris, o, d, tmin, tmax, stack, bt, bu, bv, bp);
}

template <int MODE>
MSK_DEV void trace_chunks(const DeviceScene &sc, const PathState &st, const PassParams &pp) {
    constexpr bool LDS_SCENE = MODE == 0 || MODE == 3;
    extern __shared__ float4 lds_dyn[];
    uint32_t *stack_base = (uint32_t *) lds_dyn;                         // stack_entries * MSK_BLOCK words
    float4 *scene_lds = lds_dyn + (sc.stack_entries * MSK_BLOCK) / 4;
    TraceLds g = stage_scene(sc, scene_lds, LDS_SCENE, MODE == 3);
    const LaneStack<MSK_OVF(MODE)> stack{stack_base + threadIdx.x, pp.stack_ovf + (size_t) blockIdx.x * MSK_BLOCK + threadIdx.x,
                                     (int) sc.stack_entries, (size_t) gridDim.x * MSK_BLOCK,
                                     MSK_OVF(MODE) ? stack_base + sc.stack_entries * MSK_BLOCK + threadIdx.x * 4 : nullptr};
    const uint32_t gwave = (blockIdx.x * MSK_BLOCK + threadIdx.x) / MSK_WAVE;
    const uint32_t lwave = gwave / pp.trace_split, sub = gwave % pp.trace_split;     // region of this launch, and which of its chunks
    const uint32_t lane = threadIdx.x & (MSK_WAVE - 1);
    if (lwave >= pp.region_count) return;
    const uint32_t wave = pp.region_first + lwave;
    const RegionView rv = region_view(wave, pp.region_size, pp.regions[wave].count, pp.regions[wave].half_ns);
    for (uint32_t c = sub * MSK_WAVE + lane; c < rv.n; c += MSK_WAVE * pp.trace_split) {
        const uint32_t i = rv.slot(c);
        const float4 ro = ld4<3>(st.ray_o + i);
        float4 rd = st.ray_d[i];
        const bool has_shadow = c < rv.ns;
        rd.w = slot_tmax(rd.w);
        const f3 o = mk3(ro.x, ro.y, ro.z);
        float bt, bu, bv; uint32_t bp;
        uint32_t unocc = 0;
        if (has_shadow) {
            const float4 s = ld4<3>(st.sh + i);
            const bool occ = traverse_scene<MODE, true>(sc, g, o, mk3(s.x, s.y, s.z), ro.w, s.w, stack, &bt, &bu, &bv, &bp);
            unocc = occ ? 0u : MSK_HIT_UNOCCLUDED;
        }
        traverse_scene<MODE, false>(sc, g, o, mk3(rd.x, rd.y, rd.z), ro.w, rd.w, stack, &bt, &bu, &bv, &bp);
        const bool valid = (bp != MSK_NO_PRIM) && (bt != rd.w);           // scene.cpp:234 tfar != maxt
        st4<4>(st.hit + i, make_float4(valid ? bt : MSK_INF_F, bu, bv, __uint_as_float((valid ? bp : MSK_PRIM_MASK) | unocc)));
    }
}

template <int MODE>
__global__ void __launch_bounds__(MSK_BLOCK)
k_trace(DeviceScene sc, PathState st, PassParams pp) { trace_chunks<MODE>(sc, st, pp); }

// ------------------------------------------------------------------------------------------
// k_trace_q: the LDS-resident binary tree walked as two JOB QUEUES per wave — first the region's shadow rays (slots c < ns,
// any-hit), then its extension rays (closest hit) — with the while-while loop of traverse() and lane replacement hoisted into
// its outer iteration: when at least `refill` lanes have no ray, they take the wave's next jobs.  In k_trace<0> a chunk's 64
// rays start together and the wave steps until the longest of them is done (36 % of the lanes active in an inner-node step);
// here a finished lane idles only until enough of its neighbours have finished too.  Both queues run specialised code (the
// any-hit loop returns at the first accepted triangle, the closest-hit loop keeps (t, prim) minima), unlike k_trace_r's one
// stream for both.  The shadow results wait in an LDS bit per slot until the slot's extension ray writes the hit record.
// Same per-ray arithmetic as traverse(): same hits, bit for bit.
// (Round 5, measured and rejected: a wave's jobs ORDERED by where their rays start — a counting sort over 64 keys with LDS atomics
// before each queue, 4 x 4 x 4 cells of the origin, or 2 x 2 x 2 cells + the direction's octant; perm[q] instead of q —, so that the
// rays that walk the tree together are neighbours.  Same films; the kernel alone 18.6 / 19.7 ms per bench step against 13.2, the step
// 35.1 / 35.5 against 30.6 ms: 4.3 KB more LDS per wave (five instead of eight waves per SIMD), 28 bytes of scratch at the 64-VGPR
// cap, every ray's origin read twice and gathered; whatever the walk gains from coherent neighbours is a fraction of that.)
// ------------------------------------------------------------------------------------------
// MSK_TQ_SEL: the LDS-resident tree's slabs ordered by address instead of by v_min / v_max (see the inner-node loop)
#ifndef MSK_TQ_SEL
#define MSK_TQ_SEL 1
#endif

template <bool ANY>
MSK_DEV void trace_queue(const DeviceScene &sc, const TraceLds &g, const PathState &st, const RegionView &rv, const LaneStack<false> &stack,
                         uint32_t n_jobs, uint32_t sub, uint32_t split, uint32_t *unocc_bits, uint32_t lane, uint32_t refill) {
    const uint32_t DONE = 0xffffffffu;
    // this wave's jobs: the slots of the chunks k with k % split == sub, in order
    const uint32_t n_chunks = (n_jobs + MSK_WAVE - 1) / MSK_WAVE;
    const uint32_t my_chunks = n_chunks > sub ? (n_chunks - sub + split - 1) / split : 0u;
    uint32_t my_total = my_chunks * MSK_WAVE;
    if (my_chunks && ((my_chunks - 1) * split + sub) == n_chunks - 1) my_total -= n_chunks * MSK_WAVE - n_jobs;      // the partial last chunk
    uint32_t next = 0;                                   // jobs handed out so far (wave-uniform)
    bool active = false;
    uint32_t c = 0;
    f3 o = mk3(0, 0, 0), d = o, idir = o, oi = o;
    float tmin = 0.f, tfar = 0.f, bt = 0.f, bu = 0.f, bv = 0.f;
    uint32_t bp = MSK_NO_PRIM, cur = DONE;
    int sp = 0;
    bool occluded = false;
    struct { uint32_t nx, ny, nz, fx, fy, fz; } sel = {0u, 8u, 16u, 24u, 32u, 40u};
    (void) sel;
    for (;;) {
        const unsigned long long idle = __ballot(!active);
        if (next < my_total && ((uint32_t) __popcll(idle) >= refill || idle == ~0ull)) {
            if (!active) {
                const uint32_t q = next + (uint32_t) __popcll(idle & ((1ull << lane) - 1ull));
                if (q < my_total) {
                    c = ((q >> 6) * split + sub) * MSK_WAVE + (q & 63u);
                    const uint32_t slot = rv.slot(c);
                    const float4 ro = st.ray_o[slot];
                    const float4 rd = ANY ? st.sh[slot] : st.ray_d[slot];
                    o = mk3(ro.x, ro.y, ro.z); d = mk3(rd.x, rd.y, rd.z);
                    tmin = ro.w; tfar = ANY ? rd.w : slot_tmax(rd.w);
                    idir = slab_idir(d); oi = mk3(o.x * idir.x, o.y * idir.y, o.z * idir.z);
#if MSK_TQ_SEL
                    // node = [lo.x pair, lo.y pair | lo.z pair, hi.x pair | hi.y pair, hi.z pair | refs]: byte offsets 0 8 16 24 32 40
                    sel.nx = idir.x < 0.f ? 24u : 0u; sel.fx = sel.nx ^ 24u;
                    sel.ny = idir.y < 0.f ? 32u : 8u; sel.fy = sel.ny ^ 40u;
                    sel.nz = idir.z < 0.f ? 40u : 16u; sel.fz = sel.nz ^ 56u;
#endif
                    bt = tfar; bu = 0.f; bv = 0.f; bp = MSK_NO_PRIM; sp = 0; occluded = false;
                    cur = sc.n_tris ? sc.root_ref : DONE;
                    active = true;
                }
            }
            next += (uint32_t) __popcll(idle);
        }
        if (__ballot(active) == 0ull) break;             // (an all-idle wave refills while jobs remain: nothing is left)
        // ---- inner nodes, until every lane with a ray holds a leaf or has run out of nodes
        while (active && !(cur & MSK_LEAF_BIT)) {
            // (Measured and rejected, round 3: near / far planes picked by LDS address from per-ray sign offsets instead of by
            // v_min / v_max, as node4_step does for trees in HBM — 9 % fewer VALU instructions per segment, but three more live
            // registers in a kernel pinned at 64: 28 instead of 12 bytes of scratch, +26 % HBM-side bytes per launch, trace time
            // -1.5 % alone and unchanged beside the shading kernel.  Round 4, with registers to spare (58 VGPRs since the build
            // stopped pairing scalar arithmetic, __graft_entry__.HIPCC_FLAGS): the stack's top entry in a register, refilled
            // from LDS behind the pop — 59 VGPRs, trace alone 16.4 vs 15.9 ms, step unchanged.)
            float t0, t1; bool h0, h1; uint32_t c0, c1;
#if MSK_TQ_SEL
            // near / far plane PAIRS by LDS address (round 5): the node's 12 planes are six float pairs {child 0, child 1}; the ray
            // holds, per axis, the byte offset of its near pair (lo or hi, by the sign of its direction) and of its far pair: six
            // 8-byte reads instead of three 16-byte ones, and no v_min / v_max to order the slabs (12 of a visit's ~56 VALU instructions)
            {
                const char *nb = (const char *) g.nodes + ((size_t) cur << 6);
                const float2 nx = *(const float2 *) (nb + sel.nx), ny = *(const float2 *) (nb + sel.ny), nz = *(const float2 *) (nb + sel.nz);
                const float2 fx = *(const float2 *) (nb + sel.fx), fy = *(const float2 *) (nb + sel.fy), fz = *(const float2 *) (nb + sel.fz);
                const uint2 m = *(const uint2 *) (nb + 48);
                // (max_raw / min_raw: v_max_f32 / v_min_f32 as they are — fmaxf against a value that came from memory makes the compiler
                // quiet a possible signalling NaN first, one more v_max per use; tmin and bt are never NaN)
                t0 = fmaxf(fmaxf(__fmaf_rn(nx.x, idir.x, -oi.x), __fmaf_rn(ny.x, idir.y, -oi.y)), max_raw(__fmaf_rn(nz.x, idir.z, -oi.z), tmin));
                t1 = fmaxf(fmaxf(__fmaf_rn(nx.y, idir.x, -oi.x), __fmaf_rn(ny.y, idir.y, -oi.y)), max_raw(__fmaf_rn(nz.y, idir.z, -oi.z), tmin));
                const float e0 = fminf(fminf(__fmaf_rn(fx.x, idir.x, -oi.x), __fmaf_rn(fy.x, idir.y, -oi.y)), min_raw(__fmaf_rn(fz.x, idir.z, -oi.z), bt));
                const float e1 = fminf(fminf(__fmaf_rn(fx.y, idir.x, -oi.x), __fmaf_rn(fy.y, idir.y, -oi.y)), min_raw(__fmaf_rn(fz.y, idir.z, -oi.z), bt));
                h0 = t0 <= e0 * 1.0000004f; h1 = t1 <= e1 * 1.0000004f;
                c0 = m.x; c1 = m.y;
            }
#else
            {
                const float4 *n = g.nodes + (size_t) cur * 4;
                const float4 a = n[0], b = n[1], cc = n[2], m = n[3];
                h0 = box_test(a.x, a.z, b.x, b.z, cc.x, cc.z, idir, oi, tmin, bt, &t0);
                h1 = box_test(a.y, a.w, b.y, b.w, cc.y, cc.w, idir, oi, tmin, bt, &t1);
                c0 = __float_as_uint(m.x); c1 = __float_as_uint(m.y);
            }
#endif
            if (h0 && h1) {
                const bool swap = t1 < t0;            // (nearest first for the any-hit queue too: node order instead measured +-0, round 5)
                cur = swap ? c1 : c0;
                stack.push(sp, swap ? c0 : c1);
            } else if (h0) { cur = c0; }
            else if (h1) { cur = c1; }
            else if (sp > 0) { cur = stack.pop(sp); }
            else { cur = DONE; }
        }
        // ---- the leaf
        if (active && cur != DONE) {
            const uint32_t first = (cur & 0x7fffffffu) >> 5, cnt = cur & 31u;
            for (uint32_t i = 0; i < cnt; ++i) {
                const float4 *q = g.tris + (size_t) (first + i) * 6;
                const float4 q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3];
                float t, u, v;
                if (tri_test(q0, q1, q2, q3, o, d, tmin, tfar, &t, &u, &v, q + 4, sc.tri_pad)) {
                    if (ANY) { occluded = true; break; }
                    const uint32_t prim = __float_as_uint(q0.w);
                    if (bp == MSK_NO_PRIM || t < bt || (t == bt && (prim & MSK_PRIM_ID) < (bp & MSK_PRIM_ID))) { bt = t; bu = u; bv = v; bp = prim; }
                }
            }
            if (sp > 0 && !(ANY && occluded)) cur = stack.pop(sp); else cur = DONE;
        }
        // ---- finished rays
        if (active && cur == DONE) {
            if (ANY) { if (!occluded) atomicOr(&unocc_bits[c >> 5], 1u << (c & 31u)); }
            else {
                const bool valid = (bp != MSK_NO_PRIM) && (bt != tfar);           // scene.cpp:234 tfar != maxt
                const uint32_t unocc = ((unocc_bits[c >> 5] >> (c & 31u)) & 1u) ? MSK_HIT_UNOCCLUDED : 0u;
                st.hit[rv.slot(c)] = make_float4(valid ? bt : MSK_INF_F, bu, bv, __uint_as_float((valid ? bp : MSK_PRIM_MASK) | unocc));
            }
            active = false;
        }
    }
}

__global__ void __launch_bounds__(MSK_BLOCK) __attribute__((amdgpu_waves_per_eu(8, 8)))
k_trace_q(DeviceScene sc, PathState st, PassParams pp, uint32_t refill, uint32_t bits_f4) {
    extern __shared__ float4 lds_dyn[];
    uint32_t *stack_base = (uint32_t *) lds_dyn;                         // stack_entries * MSK_BLOCK words
    float4 *scene_lds = lds_dyn + (sc.stack_entries * MSK_BLOCK) / 4;
    const TraceLds g = stage_scene(sc, scene_lds, true, false);
    // behind the staged scene: one bit per slot of a region and wave (the shadow rays' results)
    uint32_t *bits = (uint32_t *) (lds_dyn + bits_f4) + (threadIdx.x / MSK_WAVE) * (pp.region_size / 32u);
    const LaneStack<false> stack{stack_base + threadIdx.x, nullptr, (int) sc.stack_entries, 0};
    const uint32_t gwave = (blockIdx.x * MSK_BLOCK + threadIdx.x) / MSK_WAVE;
    const uint32_t lwave = gwave / pp.trace_split, sub = gwave % pp.trace_split;
    const uint32_t lane = threadIdx.x & (MSK_WAVE - 1);
    if (lwave >= pp.region_count) return;
    const uint32_t wave = pp.region_first + lwave;
    const RegionView rv = region_view(wave, pp.region_size, pp.regions[wave].count, pp.regions[wave].half_ns);
    for (uint32_t i = lane; i < pp.region_size / 32u; i += MSK_WAVE) bits[i] = 0u;
    wave_sync();
    trace_queue<true>(sc, g, st, rv, stack, rv.ns, sub, pp.trace_split, bits, lane, refill);
    wave_sync();
    trace_queue<false>(sc, g, st, rv, stack, rv.n, sub, pp.trace_split, bits, lane, refill);
}

// ------------------------------------------------------------------------------------------
// k_trace_r: the same rays with lane replacement.  Every lane is a small state machine {slot, phase (shadow / closest),
// traversal cursor}; a lane whose ray is finished takes the region's next slot once at least MSK_REFILL lanes are idle
// (or nothing else is running), so a long ray no longer idles the 63 lanes that shared its chunk.  Each lane's
// arithmetic is exactly traverse()'s: same hit, bit for bit.
// ------------------------------------------------------------------------------------------
#ifdef MSK_COUNT
// instrumented builds only (tools/build_variant.sh count -DMSK_COUNT; read through msk_gpu_debug_counts): traversal work of
// k_trace_r.  [0] rays, [1] quanta (wave-level), [2] active lanes summed over quanta, [3] inner-node steps (wave-level),
// [4] inner-node visits (lanes), [5] triangle steps (wave-level), [6] triangle tests (lanes), [7] leaf visits (lanes)
__device__ unsigned long long msk_counts[16];
MSK_DEV bool first_active_lane() {
    const unsigned long long e = __builtin_amdgcn_read_exec();
    return __builtin_amdgcn_mbcnt_hi((uint32_t) (e >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t) e, 0u)) == 0u;
}
#define MSK_CNT(i, v) atomicAdd(&msk_counts[i], (unsigned long long) (v))
#define MSK_CNT_WAVE(i) do { if (first_active_lane()) atomicAdd(&msk_counts[i], 1ull); } while (0)
#else
#define MSK_CNT(i, v) do {} while (0)
#define MSK_CNT_WAVE(i) do {} while (0)
#endif
struct TravState {
    Sel4 sel;           // trace mode 2
    Sel4q selq;         // trace mode 5 (the unused one of the two is dead code in either instantiation)
    f3 o, d, idir, oi;
    float tmin, tmax, bt, bu, bv;
    uint32_t bp, cur;
    int sp;
};
MSK_DEV void trav_begin(TravState &t, uint32_t root_ref, uint32_t n_tris, f3 o, f3 d, float tmin, float tmax) {
    t.o = o; t.d = d; t.tmin = tmin; t.tmax = tmax;
    t.idir = slab_idir(d);
    t.sel = make_sel4(t.idir); t.selq = make_sel4q(t.idir);
    t.oi = mk3(o.x * t.idir.x, o.y * t.idir.y, o.z * t.idir.z);
    t.bt = tmax; t.bu = 0.f; t.bv = 0.f; t.bp = MSK_NO_PRIM; t.sp = 0;
    t.cur = n_tris ? root_ref : 0xffffffffu;
}
// one work quantum: up to `max_inner` inner nodes, then (if the lane holds one) a leaf.  Returns true when an any-hit
// query found its hit.  t.cur == 0xffffffff afterwards means the traversal is complete.
// `any` is a per-lane run-time flag on purpose: lanes in the shadow phase and lanes in the closest-hit phase share one
// instruction stream instead of executing two instantiations one after the other.
template <int MODE>
MSK_DEV bool trav_quantum(const DeviceScene &sc, const TraceLds &g, TravState &t, const LaneStack<MSK_OVF(MODE)> &stack, int max_inner, bool any) {
    const uint32_t DONE = 0xffffffffu;
    int steps = 0;
    bool found = false;
    __amdgpu_buffer_rsrc_t rsrc4, rsrc_t3;
    if constexpr (MODE == 2) rsrc4 = nodes4_rsrc(sc);
    if constexpr (MODE == 5) rsrc4 = nodes4q_rsrc(sc);
    if constexpr (MODE == 6) rsrc4 = nodes4h_rsrc(sc);
    if constexpr (MSK_WIDE4H(MODE)) rsrc_t3 = tris3_rsrc(sc);
    while (!(t.cur & MSK_LEAF_BIT) && steps < max_inner) {
        ++steps;
        MSK_CNT_WAVE(3); MSK_CNT(4, 1);
        if constexpr (MODE == 4) {
            t.cur = node8_step<true>(sc.nodes8, t.cur, t.o, t.d, t.idir, t.tmin, t.bt, stack, t.sp);
            if (t.cur == MSK_NONE_REF && t.sp > 0) t.cur = stack.pop(t.sp);
        } else if constexpr (MODE == 2) {
            t.cur = node4_step<true>(rsrc4, t.cur, t.sel, t.idir, t.oi, t.tmin, t.bt, stack, t.sp);
        } else if constexpr (MODE == 5) {
            t.cur = node4q_step<true>(rsrc4, t.cur, t.selq, t.idir, t.oi, t.tmin, t.bt, stack, t.sp);
        } else if constexpr (MODE == 6) {
            t.cur = node4h_step<true>(rsrc4, t.cur, t.selq, t.idir, t.oi, t.tmin, t.bt, stack, t.sp);
        } else {
            const float4 *n = g.nodes + (size_t) t.cur * 4;
            const float4 a = n[0], b = n[1], c = n[2], m = n[3];
            float t0, t1;
            const bool h0 = box_test(a.x, a.z, b.x, b.z, c.x, c.z, t.idir, t.oi, t.tmin, t.bt, &t0);
            const bool h1 = box_test(a.y, a.w, b.y, b.w, c.y, c.w, t.idir, t.oi, t.tmin, t.bt, &t1);
            const uint32_t c0 = __float_as_uint(m.x), c1 = __float_as_uint(m.y);
            if (h0 && h1) {
                const bool swap = t1 < t0;
                t.cur = swap ? c1 : c0;
                stack.push(t.sp, swap ? c0 : c1);
            } else if (h0) { t.cur = c0; }
            else if (h1) { t.cur = c1; }
            else if (t.sp > 0) { t.cur = stack.pop(t.sp); }
            else { t.cur = DONE; }
        }
    }
    // Measured and rejected on the 64-byte-node kernel (round 3, config-5-class render, 78.3 ms): the next triangle's loads issued
    // before the current one is tested, a leaf of two triangles as one memory round trip (12 more live registers = 56 bytes of
    // scratch at six waves: 85.6 ms; five waves without scratch: 87.7); postponed leaves — a lane that reaches a leaf parks it, goes
    // on with its stack and tests the parked leaf at the end of the quantum, so that inner-node steps run fuller: 78.1 ms (config-3
    // class 94.0 vs 96.2): the steps it fills are paid back by later culling.
    if (t.cur != DONE && (t.cur & MSK_LEAF_BIT)) {
        const uint32_t first = (t.cur & 0x7fffffffu) >> 5, cnt = t.cur & 31u;
        MSK_CNT(7, 1);
        // (Round 5, built, measured and taken out again: the two triangles of a leaf in two
        // phases, tri_candidate for both and then ONE tail — division, hit point, D10 bounds, (t, prim) update: half of a triangle
        // step's instructions, which a wave runs whenever one of its ~27 lanes gets that far — on whichever was a candidate, its
        // record fetched again.  Same hits; 22 % fewer VALU instructions per leaf visit, three more loads: config-5 / config-3
        // class renders +3.8 % / +1.8 %.)
        for (uint32_t i = 0; i < cnt; ++i) {
            MSK_CNT_WAVE(5); MSK_CNT(6, 1);
            const float4 *q = g.tris + (size_t) (first + i) * (MSK_OVF(MODE) ? 4 : 6);
            TriRec r;
            bool acc;
            float tt, u, v;
            if constexpr (MSK_WIDE4H(MODE)) {
                load_tri_rec(rsrc_t3, first + i, r);
                acc = tri_test_rec(r, t.o, t.d, t.tmin, t.tmax, &tt, &u, &v, sc.tri_pad);
            } else {
                r.q0 = q[0]; r.q1 = q[1]; r.q2 = q[2]; r.q3 = q[3];
                acc = tri_test(r.q0, r.q1, r.q2, r.q3, t.o, t.d, t.tmin, t.tmax, &tt, &u, &v, MSK_OVF(MODE) ? nullptr : q + 4, sc.tri_pad);
            }
            if (acc) {
                if (any) { found = true; break; }
                const uint32_t prim = __float_as_uint(r.q0.w);
                if (t.bp == MSK_NO_PRIM || tt < t.bt || (tt == t.bt && (prim & MSK_PRIM_ID) < (t.bp & MSK_PRIM_ID))) { t.bt = tt; t.bu = u; t.bv = v; t.bp = prim; }
            }
        }
        if (t.sp > 0 && !found) { t.cur = stack.pop(t.sp); } else t.cur = DONE;
    }
    return found;
}

#ifndef MSK_THIN_RAYS
#define MSK_THIN_RAYS 1          /* 0 (A/B builds): thin regions walked like any other */
#endif
#ifndef MSK_THIN_RAYS_LDS
#define MSK_THIN_RAYS_LDS 0      /* the same inside k_wavefront (LDS-resident scenes): built, bit-identical, bench step 33.0 vs 33.0 ms: off */
#endif
// A THIN region — its live slots and their shadow rays together fit one wave (n + ns <= 64: the last third of a pass, when Russian
// roulette's tail drains) — is walked one RAY per lane instead of one slot per lane: lanes [0, ns) take the shadow rays, lanes
// [ns, ns + n) the extension rays, all of them through trav_quantum's one instruction stream (its `any` flag is per lane), and
// the extension lane of a slot fetches the shadow lane's verdict with one cross-lane read at the end.  A slot's two rays are
// then walked side by side instead of one after the other: the launch — as long as its slowest wave — is about half as long
// (round 6; measured in profiles/r06_ab_thin.txt).  Each ray's arithmetic is trav_quantum's as everywhere: same hits.
template <int MODE>
MSK_DEV void trace_thin(const DeviceScene &sc, const PathState &st, const TraceLds &g, const RegionView &rv, const LaneStack<MSK_OVF(MODE)> &stack,
                        uint32_t lane, int max_inner) {
    const bool is_shadow = lane < rv.ns;
    const uint32_t c = is_shadow ? lane : lane - rv.ns;
    const bool work = lane < rv.ns + rv.n;
    uint32_t slot = 0;
    float tmax_ext = 0.f;
    TravState t;
    t.cur = 0xffffffffu; t.sp = 0; t.bp = MSK_NO_PRIM; t.bt = 0.f; t.bu = 0.f; t.bv = 0.f;
    if (work) {
        slot = rv.slot(c);
        const float4 ro = st.ray_o[slot];
        const uint32_t root = MODE == 4 ? sc.root_ref8 : MSK_WIDE4H(MODE) ? sc.root_ref4 : sc.root_ref;
        if (is_shadow) {
            const float4 s = st.sh[slot];
            trav_begin(t, root, sc.n_tris, mk3(ro.x, ro.y, ro.z), mk3(s.x, s.y, s.z), ro.w, s.w);
        } else {
            float4 rd = st.ray_d[slot];
            rd.w = slot_tmax(rd.w);
            tmax_ext = rd.w;
            trav_begin(t, root, sc.n_tris, mk3(ro.x, ro.y, ro.z), mk3(rd.x, rd.y, rd.z), ro.w, rd.w);
        }
    }
    bool occ = false;
    while (__ballot(t.cur != 0xffffffffu) != 0ull) {
        if (t.cur != 0xffffffffu) occ = trav_quantum<MODE>(sc, g, t, stack, max_inner, is_shadow) || occ;
    }
    // the shadow lane of slot c is lane c; its extension lane is lane ns + c
    const int verdict = __shfl((int) occ, (int) (lane - rv.ns), MSK_WAVE);
    if (work && !is_shadow) {
        const uint32_t unocc = (c < rv.ns && !verdict) ? MSK_HIT_UNOCCLUDED : 0u;
        const bool valid = (t.bp != MSK_NO_PRIM) && (t.bt != tmax_ext);
        st.hit[slot] = make_float4(valid ? t.bt : MSK_INF_F, t.bu, t.bv, __uint_as_float((valid ? t.bp : MSK_PRIM_MASK) | unocc));
    }
}

// Six waves per SIMD (80 VGPRs, no scratch in the default instantiation <5>; the unconstrained build takes 90 = five waves).
// Alone the kernel gains nothing from the sixth wave (65.3 vs 66.2 ms on the config-5-class scene), beside the shading kernel's
// 168-VGPR waves in the four-loop mode the smaller footprint is worth 7 % of the render (77.7 vs 84.0 ms); seven waves (72 VGPRs,
// 44 bytes of scratch) are slower in both (71.6 / 88.4 ms).
#ifndef MSK_TRACE_R_WAVES
#define MSK_TRACE_R_WAVES 6, 6
#endif
template <int MODE>
MSK_DEV void trace_replace(const DeviceScene &sc, const PathState &st, const PassParams &pp, int refill, int max_inner) {
    constexpr bool LDS_SCENE = MODE == 0 || MODE == 3;
    extern __shared__ float4 lds_dyn[];
    uint32_t *stack_base = (uint32_t *) lds_dyn;
    float4 *scene_lds = lds_dyn + (sc.stack_entries * MSK_BLOCK) / 4;
    TraceLds g = stage_scene(sc, scene_lds, LDS_SCENE, MODE == 3);
    const LaneStack<MSK_OVF(MODE)> stack{stack_base + threadIdx.x, pp.stack_ovf + (size_t) blockIdx.x * MSK_BLOCK + threadIdx.x,
                                     (int) sc.stack_entries, (size_t) gridDim.x * MSK_BLOCK,
                                     MSK_OVF(MODE) ? stack_base + sc.stack_entries * MSK_BLOCK + threadIdx.x * 4 : nullptr};
    const uint32_t lwave = (blockIdx.x * MSK_BLOCK + threadIdx.x) / MSK_WAVE;
    const uint32_t lane = threadIdx.x & (MSK_WAVE - 1);
    if (lwave >= pp.region_count) return;
    const uint32_t wave = pp.region_first + lwave;
    const RegionView rv = region_view(wave, pp.region_size, pp.regions[wave].count, pp.regions[wave].half_ns);
    if (MSK_THIN_RAYS && rv.n + rv.ns <= MSK_WAVE) { trace_thin<MODE>(sc, st, g, rv, stack, lane, max_inner); return; }
    const uint32_t n = rv.n;
    uint32_t next = 0;                       // wave-uniform: first slot nobody has taken yet
    bool active = false, shadow_phase = false;
    uint32_t slot = 0;
    float4 ro = make_float4(0, 0, 0, 0), rd = make_float4(0, 0, 0, 0);      // rd.w = the extension ray's tmax (slot_tmax)
    uint32_t unocc = 0;
    TravState t;
    t.cur = 0xffffffffu; t.sp = 0;
    for (;;) {
        const unsigned long long idle = __ballot(!active);
        if (next < n && idle != 0ull && ((int) __popcll(idle) >= refill || idle == ~0ull)) {
            if (!active) {
                const uint32_t c = next + (uint32_t) __popcll(idle & ((1ull << lane) - 1ull));
                if (c < n) {
                    slot = rv.slot(c);
                    ro = st.ray_o[slot]; rd = st.ray_d[slot];
                    shadow_phase = c < rv.ns;
                    rd.w = slot_tmax(rd.w);
                    unocc = 0; active = true;
                    if (shadow_phase) {
                        const float4 s = st.sh[slot];
                        trav_begin(t, MODE == 4 ? sc.root_ref8 : MSK_WIDE4H(MODE) ? sc.root_ref4 : sc.root_ref, sc.n_tris, mk3(ro.x, ro.y, ro.z), mk3(s.x, s.y, s.z), ro.w, s.w);
                    } else {
                        trav_begin(t, MODE == 4 ? sc.root_ref8 : MSK_WIDE4H(MODE) ? sc.root_ref4 : sc.root_ref, sc.n_tris, mk3(ro.x, ro.y, ro.z), mk3(rd.x, rd.y, rd.z), ro.w, rd.w);
                    }
                }
            }
            next += (uint32_t) __popcll(idle);
        }
        if (__ballot(active) == 0ull) break;          // next >= n here: an all-idle wave always refills while slots remain
        MSK_CNT_WAVE(1);
        if (active) {
            MSK_CNT(2, 1);
            const bool occ = trav_quantum<MODE>(sc, g, t, stack, max_inner, shadow_phase);
            if (t.cur == 0xffffffffu) {
                MSK_CNT(0, 1);
                if (shadow_phase) {
                    unocc = occ ? 0u : MSK_HIT_UNOCCLUDED;
                    shadow_phase = false;
                    trav_begin(t, MODE == 4 ? sc.root_ref8 : MSK_WIDE4H(MODE) ? sc.root_ref4 : sc.root_ref, sc.n_tris, mk3(ro.x, ro.y, ro.z), mk3(rd.x, rd.y, rd.z), ro.w, rd.w);
                } else {
                    const bool valid = (t.bp != MSK_NO_PRIM) && (t.bt != rd.w);
                    st.hit[slot] = make_float4(valid ? t.bt : MSK_INF_F, t.bu, t.bv, __uint_as_float((valid ? t.bp : MSK_PRIM_MASK) | unocc));
                    active = false;
                }
            }
        }
    }
}
// The register cap is the measured optimum of the instantiation that is a default (<5>: 80 VGPRs, no scratch); the others are
// left to the compiler (the cap would cost <2> eight bytes of scratch).
template <int MODE>
__global__ void __launch_bounds__(MSK_BLOCK)
k_trace_r(DeviceScene sc, PathState st, PassParams pp, int refill, int max_inner) { trace_replace<MODE>(sc, st, pp, refill, max_inner); }
template <>
__global__ void __launch_bounds__(MSK_BLOCK) __attribute__((amdgpu_waves_per_eu(MSK_TRACE_R_WAVES)))
k_trace_r<5>(DeviceScene sc, PathState st, PassParams pp, int refill, int max_inner) { trace_replace<5>(sc, st, pp, refill, max_inner); }
template <>
__global__ void __launch_bounds__(MSK_BLOCK) __attribute__((amdgpu_waves_per_eu(MSK_TRACE_R_WAVES)))
k_trace_r<6>(DeviceScene sc, PathState st, PassParams pp, int refill, int max_inner) { trace_replace<6>(sc, st, pp, refill, max_inner); }

// tris (4 x float4: v0|prim, e1, e2, Ng) -> tris3 (3 x float4: v0|prim, e1|e2.x, e2.y e2.z - -), at scene creation
__global__ void __launch_bounds__(MSK_BLOCK) k_pack_tris3(const float4 *tris, const float4 *bounds, uint32_t n, float4 *out) {
    const uint32_t k = blockIdx.x * MSK_BLOCK + threadIdx.x;
    if (k >= n) return;
    const float4 a = tris[(size_t) k * 4], e1 = tris[(size_t) k * 4 + 1], e2 = tris[(size_t) k * 4 + 2];
    if (MSK_TRI_REC == 2) {                  // + the padded D10 bounds (msk_bvh.h: Built::bounds)
        const float4 lo = bounds[(size_t) k * 2], hi = bounds[(size_t) k * 2 + 1];
        out[(size_t) k * 4] = a; out[(size_t) k * 4 + 1] = make_float4(e1.x, e1.y, e1.z, e2.x);
        out[(size_t) k * 4 + 2] = make_float4(e2.y, e2.z, lo.x, lo.y); out[(size_t) k * 4 + 3] = make_float4(lo.z, hi.x, hi.y, hi.z);
        return;
    }
    out[(size_t) k * 3] = a; out[(size_t) k * 3 + 1] = make_float4(e1.x, e1.y, e1.z, e2.x); out[(size_t) k * 3 + 2] = make_float4(e2.y, e2.z, 0.f, 0.f);
}

// batch entry points for the sub-stage parity tests (msk_gpu_trace_closest / _any)
template <int MODE>
__global__ void __launch_bounds__(MSK_BLOCK)
k_trace_batch(DeviceScene sc, const float4 *rays, uint64_t n, float4 *out_hit, uint8_t *out_any, uint32_t *stack_ovf) {
    constexpr bool LDS_SCENE = MODE == 0 || MODE == 3;
    extern __shared__ float4 lds_dyn[];
    uint32_t *stack_base = (uint32_t *) lds_dyn;
    float4 *scene_lds = lds_dyn + (sc.stack_entries * MSK_BLOCK) / 4;
    TraceLds g = stage_scene(sc, scene_lds, LDS_SCENE, MODE == 3);
    const LaneStack<MSK_OVF(MODE)> stack{stack_base + threadIdx.x, stack_ovf + (size_t) blockIdx.x * MSK_BLOCK + threadIdx.x,
                                     (int) sc.stack_entries, (size_t) gridDim.x * MSK_BLOCK,
                                     MSK_OVF(MODE) ? stack_base + sc.stack_entries * MSK_BLOCK + threadIdx.x * 4 : nullptr};
    for (uint64_t i = (uint64_t) blockIdx.x * MSK_BLOCK + threadIdx.x; i < n; i += (uint64_t) gridDim.x * MSK_BLOCK) {
        const float4 ro = rays[2 * i], rd = rays[2 * i + 1];
        float bt, bu, bv; uint32_t bp;
        if (out_any) {
            out_any[i] = traverse_scene<MODE, true>(sc, g, mk3(ro.x, ro.y, ro.z), mk3(rd.x, rd.y, rd.z), ro.w, rd.w, stack,
                                                    &bt, &bu, &bv, &bp) ? 1 : 0;
        } else {
            traverse_scene<MODE, false>(sc, g, mk3(ro.x, ro.y, ro.z), mk3(rd.x, rd.y, rd.z), ro.w, rd.w, stack, &bt, &bu, &bv, &bp);
            const bool valid = (bp != MSK_NO_PRIM) && (bt != rd.w);
            out_hit[i] = make_float4(valid ? bt : MSK_INF_F, valid ? bu : 0.f, valid ? bv : 0.f,
                                     __uint_as_float(valid ? (bp & MSK_PRIM_ID) : MSK_NO_PRIM));
        }
    }
}

// ------------------------------------------------------------------------------------------
// shading helpers
// ------------------------------------------------------------------------------------------
struct Interaction {
    f3 p, wi;
    frame3 sh;
    float t;
    int bsdf_id, emitter_id;
};

// Small read-only scene tables of the shading kernel; staged in LDS when they fit (every hit
// looks them up through a chain of dependent indices, which from HBM/L2 costs a round trip each).
struct SceneTables {
    const float4 *tri_verts, *tri_frames, *tri_normals, *tri_uvs;
    const int4 *mesh_info;
    const float4 *bsdfs, *emitters;
    const float *emitter_d65, *cdf, *cie;
    const float4 *emitter_grid;
    const float *spectra;
};
// The same tables for a scene that holds tabulated (`regular`) spectra: the TYPE selects the code that can evaluate them (spectrum
// records of the table form, emitters' tables on their own grids).  Scenes without any — every BASELINE config — run the
// instantiations that do not carry those branches (measured: 3.2 % of the config-3-class render, 1.4 % of the config-5-class one).
struct SceneTablesR : SceneTables {};
template <class TB> struct tb_traits { static constexpr bool regular = false; };
template <> struct tb_traits<SceneTablesR> { static constexpr bool regular = true; };
MSK_DEV uint32_t tables_lds_float4s(const DeviceScene &sc) {
    return sc.n_tris * 6 + sc.n_meshes + sc.n_bsdf_f4 + sc.n_emitters * 3 + (sc.n_emitters * 95 + 3) / 4 + (sc.cdf_len + 3) / 4 + 72 + (sc.n_spectra + 3) / 4;
}
// A scene whose per-triangle tables do not fit LDS can still keep the SMALL ones there (mesh / BSDF / texture / emitter records, the
// emitters' D65 tables and area CDFs, the CIE table: every bounce and every finished sample looks several of them up through
// dependent indices): MSK_SMALL_TABLES_KB (compile time, default 16; 0 = all tables from HBM/L2 as in rounds 1-4).  0: not staged.
#ifndef MSK_SMALL_TABLES_KB
#define MSK_SMALL_TABLES_KB 16
#endif
MSK_DEV uint32_t small_tables_float4s(const DeviceScene &sc) {
    const uint32_t n = sc.n_meshes + sc.n_bsdf_f4 + sc.n_emitters * 3 + (sc.n_emitters * 95 + 3) / 4 + (sc.cdf_len + 3) / 4 + 72 + (sc.n_spectra + 3) / 4;
    return n * 16u <= MSK_SMALL_TABLES_KB * 1024u ? n : 0u;
}
template <bool LDS_TABLES>
MSK_DEV SceneTables stage_tables(const DeviceScene &sc, float4 *lds) {
    SceneTables t;
    t.tri_normals = sc.tri_normals; t.tri_uvs = sc.tri_uvs;
    float4 *p = lds;
    auto copy4 = [&](const float4 *src, uint32_t n) { for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) p[i] = src[i]; float4 *r = p; p += n; return r; };
    auto copy1 = [&](const float *src, uint32_t n) { float *d = (float *) p; for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) d[i] = src[i]; p += (n + 3) / 4; return d; };
    if (!LDS_TABLES) {
        t.tri_verts = sc.tri_verts; t.tri_frames = sc.tri_frames;
        if (lds == nullptr || small_tables_float4s(sc) == 0u) {
            t.mesh_info = sc.mesh_info; t.bsdfs = sc.bsdfs; t.emitters = sc.emitters;
            t.emitter_d65 = sc.emitter_d65; t.cdf = sc.cdf; t.cie = sc.cie;
            t.emitter_grid = sc.emitter_grid; t.spectra = sc.spectra;
            return t;
        }
    } else {
        t.tri_verts = copy4(sc.tri_verts, sc.n_tris * 3);
        t.tri_frames = copy4(sc.tri_frames, sc.n_tris * 3);
    }
    t.mesh_info = (const int4 *) copy4((const float4 *) sc.mesh_info, sc.n_meshes);
    t.bsdfs = copy4(sc.bsdfs, sc.n_bsdf_f4);
    t.emitters = copy4(sc.emitters, sc.n_emitters * 2);
    t.emitter_d65 = copy1(sc.emitter_d65, sc.n_emitters * 95);
    t.cdf = copy1(sc.cdf, sc.cdf_len);
    t.cie = copy1(sc.cie, 285);
    t.emitter_grid = copy4(sc.emitter_grid, sc.n_emitters);          // (after the older tables: their LDS offsets are those of ABI v6)
    t.spectra = copy1(sc.spectra, sc.n_spectra);
    // (Round 5, measured and taken out: the area emitters' triangles — what every NEE sample reads — staged beside them: +-0;
    // tri_verts and tri_frames of a scene whose tables stay in HBM interleaved into ONE 128-byte line per triangle, so that a hit
    // touches one line instead of pieces of two to four: config-5 / config-3 class renders 116.2 / 138.9 ms against 115.9 / 138.2,
    // shading alone 35.9 vs 36.1 ms — the gathers hit in L2 / the Infinity Cache either way.)
    __syncthreads();
    return t;
}

// The per-triangle part of mesh.cpp:50-101 / interaction.h:55-60: geometric normal, dp_du (coordinate_system of the normal,
// or from the texcoords, mesh.cpp:66-80) and — for a mesh without vertex normals, whose shading normal is the geometric one —
// the finished frame.  Runs once per scene with the arithmetic the per-hit code used to run, so hits see the same bits.
__global__ void k_tri_frames(DeviceScene sc, float4 *out) {
    const uint32_t prim = blockIdx.x * blockDim.x + threadIdx.x;
    if (prim >= sc.n_tris) return;
    const float4 a = sc.tri_verts[(size_t) prim * 3], b = sc.tri_verts[(size_t) prim * 3 + 1], c = sc.tri_verts[(size_t) prim * 3 + 2];
    const int4 mi = sc.mesh_info[__float_as_uint(a.w)];
    const f3 p0 = mk3(a.x, a.y, a.z), p1 = mk3(b.x, b.y, b.z), p2 = mk3(c.x, c.y, c.z);
    const f3 dp0 = p1 - p0, dp1 = p2 - p0;
    const f3 n = normalized(cross(dp0, dp1));
    f3 dp_du, dp_dv;
    coordinate_system(n, &dp_du, &dp_dv);
    if (mi.z & 2) {                                   // mesh.cpp:68-80
        const float4 ua = sc.tri_uvs[(size_t) prim * 2], ub = sc.tri_uvs[(size_t) prim * 2 + 1];
        const float d0x = ua.z - ua.x, d0y = ua.w - ua.y, d1x = ub.x - ua.x, d1y = ub.y - ua.y;
        const float det = d0x * d1y - d0y * d1x, inv_det = 1.f / det;
        if (det != 0.f) {
            dp_du = (dp0 * d1y - dp1 * d0y) * inv_det;
            dp_dv = (dp0 * (-d1x) + dp1 * d0x) * inv_det;
        }
    }
    f3 s = dp_du, t = mk3(0.f, 0.f, 0.f);
    if (!(mi.z & 1)) {                                // interaction.h:55-60 with sh_frame.n = n
        const f3 ff = (-n) * dot(n, dp_du) + dp_du;
        s = normalized(ff);
        t = cross(n, s);
    }
    out[(size_t) prim * 3] = make_float4(n.x, n.y, n.z, 0.f);
    out[(size_t) prim * 3 + 1] = make_float4(s.x, s.y, s.z, 0.f);
    out[(size_t) prim * 3 + 2] = make_float4(t.x, t.y, t.z, 0.f);
}

// mesh.cpp:50-101 + interaction.cpp:23-37 + interaction.h:55-60
MSK_DEV Interaction make_interaction(const SceneTables &sc, float4 hit, f3 ray_d) {
    Interaction si;
    const uint32_t prim = __float_as_uint(hit.w);
    const float4 a = sc.tri_verts[(size_t) prim * 3], b = sc.tri_verts[(size_t) prim * 3 + 1],
                 c = sc.tri_verts[(size_t) prim * 3 + 2];
    const float4 f1 = sc.tri_frames[(size_t) prim * 3 + 1];
    const int4 mi = sc.mesh_info[__float_as_uint(a.w)];
    const f3 p0 = mk3(a.x, a.y, a.z), p1 = mk3(b.x, b.y, b.z), p2 = mk3(c.x, c.y, c.z);
    const float b1 = hit.y, b2 = hit.z, b0 = 1.f - b1 - b2;
    si.t = hit.x;
    si.p = p0 * b0 + p1 * b1 + p2 * b2;
    if (mi.z & 1) {                                   // mesh.cpp:81-96: the frame follows the interpolated normal
        const float4 na = sc.tri_normals[(size_t) prim * 3], nb = sc.tri_normals[(size_t) prim * 3 + 1],
                     nc = sc.tri_normals[(size_t) prim * 3 + 2];
        si.sh.n = normalized(mk3(na.x, na.y, na.z) * b0 + mk3(nb.x, nb.y, nb.z) * b1 + mk3(nc.x, nc.y, nc.z) * b2);
        const f3 dp_du = mk3(f1.x, f1.y, f1.z);
        const f3 ff = (-si.sh.n) * dot(si.sh.n, dp_du) + dp_du;
        si.sh.s = normalized(ff);
        si.sh.t = cross(si.sh.n, si.sh.s);
    } else {                                          // per-triangle constants (k_tri_frames)
        const float4 f0 = sc.tri_frames[(size_t) prim * 3], f2 = sc.tri_frames[(size_t) prim * 3 + 2];
        si.sh.n = mk3(f0.x, f0.y, f0.z);
        si.sh.s = mk3(f1.x, f1.y, f1.z);
        si.sh.t = mk3(f2.x, f2.y, f2.z);
    }
    si.wi = si.sh.to_local(-ray_d);
    si.bsdf_id = mi.x; si.emitter_id = mi.y;
    return si;
}

// spectra/srgb_d65.cpp:34-36.  With SceneTablesR: the emitter's table on its own grid — the D65 grid again for the srgb_d65 form
// (same constants, same bits), or a `regular` radiance as it stands (area.cpp:51-54 with RegularSpectrum::eval, regular.cpp:148).
template <class TB>
MSK_DEV spec emitter_radiance(const TB &sc, int e, spec wl) {
    const float4 c = sc.emitters[2 * e];
    if (tb_traits<TB>::regular) {
        const float4 g = sc.emitter_grid[e];
        const spec t = regular_eval_grid(sc.emitter_d65 + 95 * e, g.x, g.y, __float_as_uint(g.z), wl);
        if (g.w != 0.f) return t;
        return t * srgb_model_eval(c.x, c.y, c.z, wl);
    }
    return regular_eval(sc.emitter_d65 + 95 * e, wl) * srgb_model_eval(c.x, c.y, c.z, wl);
}

// ------------------------------------------------------------------------------------------
// BSDF layer (bsdfs/diffuse.cpp:18-57, bsdfs/roughconductor.cpp:52-120, bsdfs/twosided.cpp:38-101).
// A bsdf record is msk_bsdf_desc as 7 float4: {type, back, r0, r1} {r2, au, av, sample_visible} eta k spec trans
// {ior_eta, ior_inv_eta, texture record offset, reflectance_scale}.
// ------------------------------------------------------------------------------------------
struct BsdfRec { float4 a, b, eta, k, spec, trans, ior; };
MSK_DEV BsdfRec load_bsdf(const SceneTables &tb, int id) {
    BsdfRec r; const float4 *p = tb.bsdfs + (size_t) id * MSK_BSDF_F4;
    r.a = p[0]; r.b = p[1]; r.eta = p[2]; r.k = p[3]; r.spec = p[4]; r.trans = p[5]; r.ior = p[6];
    return r;
}
// textures/checkerboard.cpp:24-33 at the hit's uv (mesh.cpp:66,68-72): the coefficients SmoothDiffuse::m_reflectance->eval(si)
// evaluates.  `rec` = float4 offset of the texture record in tb.bsdfs.
MSK_DEV f3 checkerboard_coeffs(const SceneTables &tb, uint32_t rec, float4 hit) {
    const uint32_t prim = __float_as_uint(hit.w);
    const int4 mi = tb.mesh_info[__float_as_uint(tb.tri_verts[(size_t) prim * 3].w)];
    float u = hit.y, v = hit.z;
    if (mi.z & 2) {
        const float4 ua = tb.tri_uvs[(size_t) prim * 2], ub = tb.tri_uvs[(size_t) prim * 2 + 1];
        const float b1 = hit.y, b2 = hit.z, b0 = 1.f - b1 - b2;
        u = ua.x * b0 + ua.z * b1 + ub.x * b2;
        v = ua.y * b0 + ua.w * b1 + ub.y * b2;
    }
    const float4 t0 = tb.bsdfs[rec], t1 = tb.bsdfs[rec + 1], m = tb.bsdfs[rec + 2];
    const float x = m.x * u + (m.y * v + t0.w * 1.f), y = m.z * u + (m.w * v + t1.w * 1.f);
    const float fu = x - floorf(x), fv = y - floorf(y);
    const bool first = (fu > .5f) == (fv > .5f);
    return first ? mk3(t0.x, t0.y, t0.z) : mk3(t1.x, t1.y, t1.z);
}
// a spectrum record: {c0, c1, c2, scale >= 0} = scale * S(c, l), or {lambda_min, inv_interval, first | last << 24, -1} = the
// `regular` spectrum whose values start at tb.spectra[first] (msk_spectrum_desc::regular; RegularSpectrum::eval, regular.cpp:148)
template <class TB>
MSK_DEV spec spectrum_eval(const TB &tb, float4 s, spec wl) {
    if (tb_traits<TB>::regular && s.w < 0.f) {
        const uint32_t w = __float_as_uint(s.z);
        return regular_eval_grid(tb.spectra + (w & 0xffffffu), s.x, s.y, w >> 24, wl);
    }
    return srgb_model_eval(s.x, s.y, s.z, wl) * s.w;
}
MSK_DEV float clamp_alpha(float a) { return fmax_std(a, 1e-4f); }

// bsdfs/roughdielectric.cpp:118-190 eval + pdf (both lobes, TransportMode::Radiance)
template <class TB>
MSK_DEV void roughdielectric_eval_pdf(const TB &tb, const BsdfRec &b, f3 wi, f3 wo, spec wl, spec *val, float *pdf) {
    const float cos_i = wi.z, cos_o = wo.z;
    if (cos_i == 0.f) return;
    const float au = clamp_alpha(b.b.y), av = clamp_alpha(b.b.z);
    const bool reflect = cos_i * cos_o > 0.f;
    const float eta = cos_i > 0.f ? b.ior.x : b.ior.y, inv_eta = cos_i > 0.f ? b.ior.y : b.ior.x;
    f3 m = normalized(wi + wo * (reflect ? 1.f : eta));
    m = m * copysignf(1.f, m.z);
    float F, ct, e_it, e_ti;
    const float D = distr_eval(m, au, av);
    fresnel_dielectric(dot(wi, m), b.ior.x, &F, &ct, &e_it, &e_ti);
    const float G = smith_g1(wi, m, au, av) * smith_g1(wo, m, au, av);
    if (reflect) {
        *val = spectrum_eval(tb, b.spec, wl) * (F * D * G) / (4.f * fabsf(cos_i));
    } else {
        const float scale = inv_eta * inv_eta;
        const float denom = dot(wi, m) + eta * dot(wo, m);
        *val = spectrum_eval(tb, b.trans, wl) *
               fabsf((scale * (1.f - F) * D * G * eta * eta * dot(wi, m) * dot(wo, m)) / (cos_i * (denom * denom)));
    }
    if (dot(wi, m) * wi.z <= 0.f || dot(wo, m) * wo.z <= 0.f) return;
    const float denom = dot(wi, m) + eta * dot(wo, m);
    const float dwh_dwo = reflect ? 1.f / (4.f * dot(wo, m)) : (eta * eta * dot(wo, m)) / (denom * denom);
    float sau = au, sav = av;
    if (!__float_as_int(b.b.w)) { const float sc = 1.2f - .2f * __builtin_sqrtf(fabsf(wi.z)); sau *= sc; sav *= sc; }
    float prob = distr_eval(m, sau, sav) * m.z;
    prob *= reflect ? F : 1.f - F;
    *pdf = prob * fabsf(dwh_dwo);
}
// bsdfs/roughdielectric.cpp:57-116 sample; *eta_out = bs.eta
template <class TB>
MSK_DEV spec roughdielectric_sample(const TB &tb, const BsdfRec &b, f3 wi, float sample1, f2 sample, spec wl, f3 *wo, float *pdf,
                                    float *eta_out, bool *ok) {
    const float cos_i = wi.z;
    const float au = clamp_alpha(b.b.y), av = clamp_alpha(b.b.z);
    float sau = au, sav = av;
    if (!__float_as_int(b.b.w)) { const float sc = 1.2f - .2f * __builtin_sqrtf(fabsf(cos_i)); sau *= sc; sav *= sc; }
    const f3 m = sample_ggx(sample, sau, sav, pdf);
    if (*pdf == 0.f) return splat(0.f);
    *ok = true;
    float F, cos_t, eta_it, eta_ti;
    fresnel_dielectric(dot(wi, m), b.ior.x, &F, &cos_t, &eta_it, &eta_ti);
    const bool selected_r = sample1 <= F;
    spec weight = splat(1.f);
    *pdf *= selected_r ? F : (1.f - F);
    const float bs_eta = selected_r ? 1.f : eta_it;
    float dwh_dwo;
    if (selected_r) {
        *wo = m * 2.f * dot(wi, m) - wi;
        weight = weight * spectrum_eval(tb, b.spec, wl);
        dwh_dwo = 1.f / (4.f * dot(*wo, m));
    } else {
        *wo = m * (dot(wi, m) * eta_ti + cos_t) - wi * eta_ti;
        weight = weight * (eta_ti * eta_ti);
        const float denom = dot(wi, m) + bs_eta * dot(*wo, m);
        dwh_dwo = (bs_eta * bs_eta) * dot(*wo, m) / (denom * denom);
    }
    if (__float_as_int(b.b.w)) weight = weight * smith_g1(*wo, m, au, av);
    else weight = weight * (smith_g1(wi, m, au, av) * smith_g1(*wo, m, au, av) * dot(wi, m) / (cos_i * m.z));
    *pdf *= fabsf(dwh_dwo);
    *eta_out = bs_eta;
    return weight;
}

// eval + pdf with wi on the front side (roughconductor.cpp:82-117 / diffuse.cpp:35-57)
// `refl` = the diffuse reflectance spectrum at wl, evaluated once per bounce by the caller (used by eval and by sample)
template <bool DIFFUSE_ONLY, class TB>
MSK_DEV void bsdf_eval_pdf(const TB &tb, const BsdfRec &b, f3 wi, f3 wo, spec wl, spec refl, spec *val, float *pdf) {
    *val = splat(0.f); *pdf = 0.f;
    const float cos_i = wi.z, cos_o = wo.z;
    if (DIFFUSE_ONLY || __float_as_int(b.a.x) == 0) {
        if (cos_i > 0.f && cos_o > 0.f) {
            *val = refl * MSK_INV_PI_F * cos_o;
            *pdf = MSK_INV_PI_F * wo.z;
        }
        return;
    }
    if (__float_as_int(b.a.x) == MSK_BSDF_ROUGHDIELECTRIC) { roughdielectric_eval_pdf(tb, b, wi, wo, wl, val, pdf); return; }
    const float au = clamp_alpha(b.b.y), av = clamp_alpha(b.b.z);
    if (cos_i > 0.f && cos_o > 0.f) {
        const f3 H = normalized(wo + wi);
        const float D = distr_eval(H, au, av);
        if (D != 0) {
            const float G = smith_g1(wi, H, au, av) * smith_g1(wo, H, au, av);
            const float result = D * G / (4.f * wi.z);
            const spec eta = spectrum_eval(tb, b.eta, wl), kk = spectrum_eval(tb, b.k, wl);
            const float c = dot(wi, H);
            spec F;
#pragma unroll
            for (int i = 0; i < 4; ++i) F.v[i] = fresnel_conductor(c, eta.v[i], kk.v[i]);
            *val = F * spectrum_eval(tb, b.spec, wl) * result;
        }
    }
    const f3 m = normalized(wo + wi);
    if (cos_i > 0.f && cos_o > 0.f && dot(wi, m) > 0.f && dot(wo, m) > 0.f) {
        if (__float_as_int(b.b.w)) *pdf = distr_eval(m, au, av) * smith_g1(wi, m, au, av) / (4.f * cos_i);
        else *pdf = (distr_eval(m, au, av) * m.z) / (4.f * dot(wo, m));
    }
}
// sample with wi on the front side; returns the weight, fills wo / pdf / ok (= a direction was produced)
template <bool DIFFUSE_ONLY, class TB>
MSK_DEV spec bsdf_sample(const TB &tb, const BsdfRec &b, f3 wi, float sample1, f2 sample, spec wl, spec refl, f3 *wo, float *pdf, float *eta_out,
                         bool *ok) {
    *wo = mk3(0.f, 0.f, 0.f); *pdf = 0.f; *ok = false; *eta_out = 1.f;
    if (!DIFFUSE_ONLY && __float_as_int(b.a.x) == MSK_BSDF_ROUGHDIELECTRIC)
        return roughdielectric_sample(tb, b, wi, sample1, sample, wl, wo, pdf, eta_out, ok);
    const float cos_i = wi.z;
    if (cos_i <= 0.f) return splat(0.f);
    *ok = true;
    if (DIFFUSE_ONLY || __float_as_int(b.a.x) == 0) {
        *wo = square_to_cosine_hemisphere(sample);
        *pdf = MSK_INV_PI_F * wo->z;
        return *pdf > 0.f ? refl : splat(0.f);
    }
    const float au = clamp_alpha(b.b.y), av = clamp_alpha(b.b.z);
    const f3 m = sample_ggx(sample, au, av, pdf);
    *wo = m * 2.f * dot(wi, m) - wi;
    if (!(*pdf != 0.f && wo->z > 0.f)) return splat(0.f);
    float weight;
    if (__float_as_int(b.b.w)) weight = smith_g1(*wo, m, au, av);
    else weight = smith_g1(wi, m, au, av) * smith_g1(*wo, m, au, av) * dot(wi, m) / (cos_i * m.z);
    *pdf /= 4.f * dot(*wo, m);
    const spec eta = spectrum_eval(tb, b.eta, wl), kk = spectrum_eval(tb, b.k, wl);
    const float c = dot(wi, m);
    spec F;
#pragma unroll
    for (int i = 0; i < 4; ++i) F.v[i] = fresnel_conductor(c, eta.v[i], kk.v[i]);
    return F * weight;
}

// ------------------------------------------------------------------------------------------
// k_shade_gen
// ------------------------------------------------------------------------------------------

// Cold kernel arguments, read where they are used (MSK_COLD_KARGS, default on since round 5).  The shading kernels take
// DeviceScene, PathState and PassParams by value: ~180 dwords of kernel arguments which the compiler loads at the kernel's entry
// and — the chunk loop needing every SGPR it can get — parks in VGPR lanes: a v_writelane per dword up front and a v_readlane,
// a VALU instruction, at every use (the camera matrices alone: 34 lanes, read back per new camera sample chunk).  What only
// the record writer and the regeneration loop need (record addresses, pixel tables, filter constants, camera) is instead read
// from the kernel-argument segment by scalar loads at that point, through a pointer the optimiser cannot see through (an empty
// `asm volatile`), so that the loads cannot be hoisted back to the entry.  Requires (DeviceScene, PathState, PassParams) to be
// the kernel's first three arguments, in that order: k_shade_gen and k_wavefront, the only kernels that run shade_region.
#ifndef MSK_COLD_KARGS
#define MSK_COLD_KARGS 1
#endif
typedef const char __attribute__((address_space(4))) *karg_ptr;
MSK_DEV karg_ptr karg_base() {
    karg_ptr p = (karg_ptr) __builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(p));
    return p;
}
template <class T> MSK_DEV T karg(karg_ptr base, size_t off) { return *(const T __attribute__((address_space(4))) *) (base + off); }
#define MSK_KARG_SC(base, T, member) karg<T>(base, offsetof(DeviceScene, member))
#define MSK_KARG_PP(base, T, member) karg<T>(base, sizeof(DeviceScene) + sizeof(PathState) + offsetof(PassParams, member))
static_assert(sizeof(DeviceScene) % 8 == 0 && sizeof(PathState) % 8 == 0, "the three argument structs are laid out back to back");

// ImageBlock::put's per-sample part (imageblock.cpp:84-96), done once per sample instead of once per (sample, target pixel):
// a sample of block pixel (lx, ly) can only reach the five target columns lx .. lx + 4 and rows ly .. ly + 4 of the bordered
// block (border = 2: the default Gaussian, radius 2).  Field i (6 bits) of the x word is the index into the filter's
// discretisation (rfilter.h:13-16: min(int(|t - pos| * scale), 32)) of target column lx + i, or MSK_W_OUT when that column is
// outside [ceil(pos - r), floor(pos + r)]; the y word likewise.  Same fp32 expressions as the replay kernels evaluate per target.
#define MSK_W_OUT 33u                       /* lut[33] = 0 in the replay kernel's copy of the table */
// Field i sits at bits 2 + 6 i .. 7 + 6 i (round 5; bits 6 i .. before): `(word >> 6 i) & 0xfc` is then the BYTE offset of the
// table entry — a shift and a mask of the fast class (2.7 SIMD cycles each) where index extraction + address scaling were
// v_bfe_u32 + v_lshlrev_b32 (4.1 each) per lookup of the film replay, whose inner loop is a third lookups.  Bits 0-1 are free.
#define MSK_W_SHIFT 2
#define MSK_W_FIELDS 0xfffffffcu
#define MSK_W_NONFINITE 0x1u                /* bit 0 of the x word: a value of this record is inf / nan (see k_resolve_rows) */
#define MSK_W_ALL_OUT ((MSK_W_OUT * 0x1041041u) << MSK_W_SHIFT)      /* five fields of MSK_W_OUT */
struct SampleWeights { uint32_t x, y; };
MSK_DEV uint32_t weight_word(float pos, uint32_t l, float radius, float scale) {
    const float lo = pos - radius, hi = pos + radius;
    uint32_t w = 0;
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        const float ft = (float) (l + (uint32_t) i);
        const int idx = min((int) fabsf((ft - pos) * scale), 32);
        w |= ((ft >= lo && ft <= hi) ? (uint32_t) idx : MSK_W_OUT) << (6 * i + MSK_W_SHIFT);
    }
    return w;
}
MSK_DEV SampleWeights sample_weights(float filter_radius, float filter_scale, uint4 pt, float px, float py) {
    // imageblock.cpp:84-85: pos - 0.5 - (offset - border)
    const float bx = px - 0.5f - (float) (int) pt.z, by = py - 0.5f - (float) (int) pt.w;
    SampleWeights r;
    r.x = weight_word(bx, pt.y & 0xffffu, filter_radius, filter_scale);
    r.y = weight_word(by, pt.y >> 16, filter_radius, filter_scale);
    return r;
}
MSK_DEV SampleWeights sample_weights(const DeviceScene &sc, uint4 pt, float px, float py) { return sample_weights(sc.filter_radius, sc.filter_scale, pt, px, py); }
// film coordinates of a pass pixel from its table entry {film index, x | y << 16 inside its block, block offset - border}
// (= film index % width, film index / width, without the integer divisions)
MSK_DEV uint32_t pixel_x(int filter_border, uint4 pt) { return (uint32_t) ((int) pt.z + filter_border) + (pt.y & 0xffffu); }
MSK_DEV uint32_t pixel_y(int filter_border, uint4 pt) { return (uint32_t) ((int) pt.w + filter_border) + (pt.y >> 16); }
MSK_DEV uint32_t nonfinite_flag(float a, float b, float c) {
    const bool fin = fabsf(a) < MSK_INF_F && fabsf(b) < MSK_INF_F && fabsf(c) < MSK_INF_F;      // false for nan too
    return fin ? 0u : MSK_W_NONFINITE;
}

// render_sample's tail for one finished path (integrator.cpp:115-125): ray weight, XYZ, film position -> sample record
// ImageBlock::put's validity test (imageblock.cpp:57-81): every value finite and, unless the block carries AOV channels
// (integrator.cpp:59-60: warn_negative = !has_aovs), not below -1e-5
MSK_DEV bool invalid_value(float a, float b, float c, bool warn_negative) {
    const bool fin = fabsf(a) < MSK_INF_F && fabsf(b) < MSK_INF_F && fabsf(c) < MSK_INF_F;
    const bool neg = warn_negative && !(a >= -1e-5f && b >= -1e-5f && c >= -1e-5f);
    return !fin || neg;
}
// returns how many of the wave's records ImageBlock::put would have warned about
template <bool DIFFUSE_ONLY>
MSK_DEV uint32_t emit_record(const DeviceScene &sc, const SceneTables &tb, const PassParams &pp, spec wl, spec res, uint32_t pix, uint32_t si) {
    // what only this function needs of the kernel's arguments (MSK_COLD_KARGS above)
    struct Cold { const uint32_t *pix_to_j; const uint4 *pix_table; float4 *rec_a; float *rec_b; float4 *aov_rgb; uint32_t spp_owned, packed, aov_groups;
                  float filter_radius, filter_scale; int filter_border; } k;
    if (MSK_COLD_KARGS) {
        const karg_ptr ka = karg_base();
        k.pix_to_j = MSK_KARG_PP(ka, const uint32_t *, pix_to_j); k.pix_table = MSK_KARG_PP(ka, const uint4 *, pix_table);
        k.rec_a = MSK_KARG_PP(ka, float4 *, rec_a); k.rec_b = MSK_KARG_PP(ka, float *, rec_b); k.aov_rgb = MSK_KARG_PP(ka, float4 *, aov_rgb);
        k.spp_owned = MSK_KARG_PP(ka, uint32_t, spp_owned); k.packed = MSK_KARG_PP(ka, uint32_t, packed);
        k.aov_groups = DIFFUSE_ONLY ? 0u : MSK_KARG_PP(ka, uint32_t, aov_groups);
        k.filter_radius = MSK_KARG_SC(ka, float, filter_radius); k.filter_scale = MSK_KARG_SC(ka, float, filter_scale);
        k.filter_border = MSK_KARG_SC(ka, int32_t, filter_border);
    } else {
        k.pix_to_j = pp.pix_to_j; k.pix_table = pp.pix_table; k.rec_a = pp.rec_a; k.rec_b = pp.rec_b; k.aov_rgb = pp.aov_rgb;
        k.spp_owned = pp.spp_owned; k.packed = pp.packed; k.aov_groups = DIFFUSE_ONLY ? 0u : pp.aov_groups;
        k.filter_radius = sc.filter_radius; k.filter_scale = sc.filter_scale; k.filter_border = sc.filter_border;
    }
    const uint32_t j = k.pix_to_j[pix];
    spec wgt;
#pragma unroll
    for (int q = 0; q < 4; ++q) { wgt.v[q] = wavelength_weight(wl.v[q]); __builtin_amdgcn_sched_barrier(0); }
    const spec result = res * wgt;
    float X, Y, Z;
    spectrum_to_xyz(tb.cie, result, wl, &X, &Y, &Z);
    const uint64_t key = counter_key(pp.seed, pix, pp.sample_first + si * pp.sample_stride);
    const f2 jit = counter_pair(key, 0);
    const uint4 pt = k.pix_table[j];
    const float px = (float) pixel_x(k.filter_border, pt) + jit.x, py = (float) pixel_y(k.filter_border, pt) + jit.y;
    const size_t r = (size_t) j * k.spp_owned + si;
    float wx = px, wy = py;
    if (k.packed) {
        const SampleWeights sw = sample_weights(k.filter_radius, k.filter_scale, pt, px, py);
        wx = __uint_as_float(sw.x | nonfinite_flag(X, Y, Z)); wy = __uint_as_float(sw.y);
    }
    st4<2>(k.rec_a + r, make_float4(X, Y, Z, wx));
    if (MSK_NT >= 2) __builtin_nontemporal_store(wy, k.rec_b + r); else k.rec_b[r] = wy;
    // integrator.cpp:59-60: warn_negative = !has_aovs — any "aov" render with channels, nested integrator or not
    bool invalid = invalid_value(X, Y, Z, DIFFUSE_ONLY || !(k.aov_rgb || k.aov_groups));
    if (!DIFFUSE_ONLY)                                              // imageblock.cpp:57-81 tests every channel of the block: the AOV groups too
        for (uint32_t g = 0; g < k.aov_groups; ++g) {
            const float4 v = pp.aov_rec[g][r];
            invalid = invalid || invalid_value(v.x, v.y, v.z, false);
        }
    if (!DIFFUSE_ONLY && k.aov_rgb) {                              // aov.cpp:124-136: the sample before ray_weight
        float x0, y0, z0;
        spectrum_to_xyz(tb.cie, res, wl, &x0, &y0, &z0);
        const float R = 3.240479f * x0 + (-1.537150f * y0 + -0.498535f * z0), G = -0.969256f * x0 + (1.875991f * y0 + 0.041556f * z0),
                    B = 0.055648f * x0 + (-0.204043f * y0 + 1.057311f * z0);
        if (k.packed) wx = __uint_as_float((__float_as_uint(wx) & MSK_W_FIELDS) | nonfinite_flag(R, G, B));
        k.aov_rgb[r] = make_float4(R, G, B, wx);
        invalid = invalid || invalid_value(R, G, B, false);
    }
    return (uint32_t) __popcll(__ballot(invalid));
}

// Finished paths are parked in a wave-local LDS queue and turned into records 64 at a time: the record arithmetic (four
// fp64 cosh, twelve CIE lookups, one RNG draw — about a fifth of this kernel's instructions) then runs with all lanes
// busy instead of once per chunk for the ~25 % of its lanes that happened to finish.
#define MSK_DONE_Q 128                     /* entries per wave: up to 63 parked + 64 new */
#define MSK_DONE_Q_F4 (MSK_DONE_Q * 5 / 2)  /* float4 of LDS per wave: wl and res (16 B per entry), id (8 B) */
struct DoneQueue { float4 *wl, *res; uint2 *id; };      // id: {film pixel, sample index}

// Material-sorted shading (general variant).  Before a region is shaded its live paths are ordered by the material class of
// the surface their ray has just hit (bits 27..28 of the hit record's prim word, delivered by the traversal): a counting
// sort over the four classes with wave ballots and prefix popcounts, the permutation kept in LDS (two bytes per slot).
// The sweep then reads the region through that permutation, so a chunk's 64 lanes run ONE BSDF's code — diffuse chunks
// skip the microfacet branches altogether (no lane enters them), conductor chunks do not wait for the dielectric lobe
// selection — instead of every chunk paying for every BSDF some lane of it needs.  Results do not depend on the order:
// a path's arithmetic is its own and its record is addressed by (pixel, sample).  What makes reading in any order legal
// is the two-half region (RegionView): nothing this sweep writes is something it still has to read.
// (Round 5, measured and taken out: the sort inside consecutive WINDOWS of 128 / 256 / 512 live paths instead of over the whole
// region — so that a window's chunks read the same few KB of every state array one after the other instead of every second
// or third 16-byte piece of the region's 32 KB per array (the general variant fetches 2.4x the bytes it writes): config-3 /
// config-5 class renders 145.2 / 145.7 / 144.3 and 118.5 / 118.5 / 118.4 ms against 144.6 and 117.4 — the pieces a chunk leaves
// of a sector are L2 hits for the chunk that takes them, windows or not.  On these two scenes the sort itself is worth nothing any
// more since the small tables moved to LDS (MSK_SORT=0: 146.1 / 117.8 ms; shading alone 65.8 vs 66.6 ms); it stays for the
// scenes it was built for — several materials on surfaces of similar area: tools/divergence_probe.py)
struct SortScratch { uint16_t *perm; uint8_t *cls; };        // per wave: region_size entries each, or {nullptr, nullptr}
MSK_DEV bool sort_by_class(const PathState &st, const RegionView &in, const SortScratch &ss, uint32_t lane) {
    uint32_t cnt[MSK_N_CLASSES] = {0u, 0u, 0u, 0u};
    for (uint32_t c0 = 0; c0 < in.n; c0 += MSK_WAVE) {
        const uint32_t c = c0 + lane;
        uint32_t k = MSK_N_CLASSES;                           // no lane
        if (c < in.n) {
            k = (((const uint32_t *) &st.hit[in.slot(c)])[3] >> MSK_CLASS_SHIFT) & (MSK_N_CLASSES - 1u);
            ss.cls[c] = (uint8_t) k;
        }
#pragma unroll
        for (uint32_t q = 0; q < MSK_N_CLASSES; ++q) cnt[q] += (uint32_t) __popcll(__ballot(k == q));
    }
    uint32_t n_present = 0;
#pragma unroll
    for (uint32_t q = 0; q + 1 < MSK_N_CLASSES; ++q) n_present += cnt[q] ? 1u : 0u;
    if (n_present <= 1u) return false;                        // one material (misses aside): the region is read in slot order
    uint32_t start[MSK_N_CLASSES];
    start[0] = 0;
#pragma unroll
    for (uint32_t q = 1; q < MSK_N_CLASSES; ++q) start[q] = start[q - 1] + cnt[q - 1];
    wave_sync();
    for (uint32_t c0 = 0; c0 < in.n; c0 += MSK_WAVE) {
        const uint32_t c = c0 + lane;
        const uint32_t k = c < in.n ? (uint32_t) ss.cls[c] : MSK_N_CLASSES;
        const unsigned long long below = (1ull << lane) - 1ull;
#pragma unroll
        for (uint32_t q = 0; q < MSK_N_CLASSES; ++q) {
            const unsigned long long m = __ballot(k == q);
            if (k == q) ss.perm[start[q] + (uint32_t) __popcll(m & below)] = (uint16_t) c;
            start[q] += (uint32_t) __popcll(m);
        }
    }
    wave_sync();
    return true;
}

// One shading sweep of region `wave` by its owner wave (see the file header); returns the region as the sweep leaves it.
// (Round 5, measured and taken out again — commit 767b33c holds the code: the sweep as TWO kernels by material class, the sorted
// region's plain-diffuse range and its misses by the diffuse code at four waves per SIMD, the classes in between + the sweep's
// tail by the general variant.  Bit-identical, and slower: config-5 / config-3 class renders +4.5 % / +3.7 % — every region is
// visited, sorted and its counters read and written twice, and two thin launches drain twice.)
template <bool DIFFUSE_ONLY, class TB>
MSK_DEV RegionView shade_region(const DeviceScene &sc, const TB &tb, const DoneQueue &dq, const SortScratch &ss, const PathState &st,
                                const PassParams &pp, uint32_t wave, uint32_t lane) {
    uint32_t n_queued = 0;
    RegionCtl rc = pp.regions[wave];
    const uint32_t n_in = rc.count;
    const RegionView in = region_view(wave, pp.region_size, rc.count, rc.half_ns);
    const uint32_t base_out = (wave * 2u + ((rc.half_ns & 1u) ^ 1u)) * pp.region_size;      // the other half
    const uint32_t last = pp.region_size - 1u;
    uint32_t cur_s = 0, cur_n = 0;                // survivors written so far: with a shadow ray (upwards from slot 0), without (downwards from `last`)
    const uint32_t n_em = sc.n_emitters;
    const bool one_emitter = n_em == 1 && (DIFFUSE_ONLY || sc.env_emitter < 0);      // one AREA emitter
    uint32_t n_done = 0, n_invalid = 0;

    // A chunk's state as it is loaded.  (Measured and rejected: issuing the NEXT chunk's loads before this one is shaded — legal
    // with two-half regions — costs 30 VGPRs = one wave per SIMD and is slower, 21.6 vs 20.7 ms of shading per bench step.  Round 3:
    // the same through LDS-DMA (global_load_lds_dwordx4 of wl / thr / res into 3 KB of LDS per wave while the current chunk is
    // shaded, read back before the chunk's stores): the 12 registers that carry the values across the loop edge push the kernel
    // to 145 VGPRs, and held at 128 it spills 80 bytes: shading 21.8 vs 17.7 ms alone, step 39.4 vs 36.1 ms; films identical.
    // Round 5: the next chunk's lines only PULLED INTO L2 while this one is shaded — one discarded dword per lane and array, six
    // more load instructions and one register (128 VGPRs, four waves, no scratch): shading 22.8 vs 20.8 ms alone, step 34.2 vs
    // 32.5 ms, config-5 class 122.7 vs 119.8 ms.  More requests make it slower, earlier ones do not make it faster: what the
    // kernel waits for is the memory system's throughput on this read / write mix, not the latency of its loads.  The same for
    // the per-triangle TABLES of a scene that keeps them in HBM: the next chunk's prim words loaded early (the traversal has
    // written them) and its tri_verts / tri_frames / tri_normals records pulled towards L2 at the end of this chunk, so that the
    // dependent gathers overlap the next chunk's state loads — 163 VGPRs, three waves, same films: config-5 / config-3 class
    // 120.3 / 144.0 ms against 116.4 / 140.8, shading alone 40.0 vs 37.3 ms.)
    struct ChunkIn { uint2 id; float4 wl, thr, res, rd4, hit, contrib; float2 aux; };
    const uint32_t first_new = n_in - rc.n_new;      // live index of the first camera sample the last sweep started
    bool sorted = false;
    if (!DIFFUSE_ONLY && ss.perm) sorted = sort_by_class(st, in, ss, lane);
    // live index of the path at position p of the sweep (the slot order, or the material order)
    auto live_index = [&](uint32_t p) { return (!DIFFUSE_ONLY && sorted && p < n_in) ? (uint32_t) ss.perm[p] : p; };
    auto load_chunk = [&](uint32_t c) {
        const uint32_t i = in.slot(c < n_in ? c : 0u);
        ChunkIn k;
        k.id = ld2<1>(st.id + i); k.wl = ld4<1>(st.wl + i); k.rd4 = ld4<2>(st.ray_d + i); k.hit = ld4<2>(st.hit + i);
        k.thr = make_float4(1.f, 1.f, 1.f, 1.f); k.res = make_float4(0.f, 0.f, 0.f, 0.f);
        if (c < first_new) { k.thr = ld4<1>(st.thr + i); k.res = ld4<1>(st.res + i); }      // (integrator.cpp:104 / path.cpp:24-25 for the rest)
        k.contrib = make_float4(0.f, 0.f, 0.f, 0.f);
        if (c < in.ns) k.contrib = ld4<1>(st.contrib + i);                          // (whole chunks, but for the one the boundary falls in)
        k.aux = make_float2(1.f, 0.f);
        if (!DIFFUSE_ONLY) k.aux = st.aux[i];
        return k;
    };
    for (uint32_t c0 = 0; c0 < n_in; c0 += MSK_WAVE) {
        const uint32_t c = live_index(c0 + lane);
        const bool active = c < n_in;
        const bool shadow_in = c < in.ns;          // this path's last bounce sent a shadow ray (ns <= n_in)
        // ---- load
        const ChunkIn cur = load_chunk(c);
        const uint2 id = cur.id;
        spec wl = from4(cur.wl), thr = from4(cur.thr), res = from4(cur.res);
        const float4 rd4 = cur.rd4;
        float4 hit = cur.hit;
        // the previous bounce's NEE term (path.cpp:60-66), now that the shadow ray has been traced
        if (shadow_in && (__float_as_uint(hit.w) & MSK_HIT_UNOCCLUDED))
            res = res + from4(cur.contrib);
        hit.w = __uint_as_float(__float_as_uint(hit.w) & MSK_PRIM_ID);
        float bs_pdf = -rd4.w;                                             // meaningful for depth > 1 (PathState::ray_d)
        float eta = cur.aux.x, nee_pdf = cur.aux.y;                        // carried only by the general variant
        uint32_t depth = id.y >> MSK_DEPTH_SHIFT;
        const uint32_t s_own = id.y & MSK_SI_MASK;
        const f3 rd = mk3(rd4.x, rd4.y, rd4.z);
        const uint32_t pix = id.x;
        const uint32_t sidx = pp.sample_first + s_own * pp.sample_stride;
        const uint64_t key = counter_key(pp.seed, pix, sidx);

        bool alive = active;
        float4 new_o = make_float4(0, 0, 0, 0), new_d = make_float4(0, 0, 0, 0), new_sh = make_float4(0, 0, 0, 0);
        spec contrib = splat(0.f);
        bool has_shadow = false;

        if (alive && hit.x == MSK_INF_F) {                                 // path.cpp:34-41 / 89-97
            if (!DIFFUSE_ONLY && sc.env_emitter >= 0) {
                // depth 1: the camera ray left the scene.  depth > 1: the BSDF sample did; its MIS weight uses the NEE
                // sample's record, which the reference does not re-query on this branch (path.cpp:90-95,103-108).
                if (depth == 1) {
                    if (!pp.hide_emitters && pp.max_depth != 0) res = res + thr * emitter_radiance(tb, sc.env_emitter, wl);   // path.cpp:33
                }
                else res = res + thr * emitter_radiance(tb, sc.env_emitter, wl) * mis_weight(bs_pdf, nee_pdf);
            }
            alive = false;
        }
        if (alive) {
            Interaction si = make_interaction(tb, hit, rd);
            // twosided.cpp:38-101: the nested BSDF of the side wi is on, with z of wi / wo flipped on the back
            BsdfRec bs = load_bsdf(tb, si.bsdf_id);
            f3 wi_s = si.wi;
            bool flipped = false;
            if (!DIFFUSE_ONLY) {
                const int back = __float_as_int(bs.a.y);
                if (back >= 0 && wi_s.z < 0.f) { wi_s.z = -wi_s.z; flipped = true; if (back != si.bsdf_id) bs = load_bsdf(tb, back); }
            }
            // AreaLight::eval at this hit (area.cpp:51-54): used by the MIS term of the previous bounce or by the directly
            // visible emitter, never both
            // With a single emitter, its radiance at the path's wavelengths is what both this hit (area.cpp:51-54) and the
            // bounce's NEE sample (area.cpp:39-44) evaluate — same function, same arguments, same bits: once per chunk for
            // every lane instead of once for the few lanes on the emitter plus once for the NEE samples.
            spec le_one = splat(0.f);
            if (one_emitter) le_one = emitter_radiance(tb, 0, wl);
            spec le_hit = splat(0.f);
            if (one_emitter) { if (si.emitter_id >= 0 && si.wi.z > 0.f) le_hit = le_one; }
            else if (si.emitter_id >= 0 && si.wi.z > 0.f) le_hit = emitter_radiance(tb, si.emitter_id, wl);
            if (depth > 1) {
                // ---- tail of the previous bounce: emitter hit by the BSDF sample (path.cpp:82-88,103-108)
                if (si.emitter_id >= 0) {
                    const spec value = le_hit;
                    // set_query (records.cpp:7-14) + pdf_emitter_direct (scene.cpp:105-112, shape.cpp:80-86)
                    float pdf = tb.emitters[2 * si.emitter_id].w;
                    const float dp = fabsf(dot(rd, si.sh.n));
                    pdf *= (dp != 0.f) ? (si.t * si.t) / dp : 0.f;
                    if (n_em != 1) pdf = pdf * (1.f / n_em);
                    res = res + thr * value * mis_weight(bs_pdf, pdf);
                }
                // ---- Russian roulette (path.cpp:116-122)
                if ((int) depth >= pp.rr_depth) {
                    const float q = fmin_std(DIFFUSE_ONLY ? max4(thr) : max4(thr) * eta * eta, 0.95f);
                    const float u = counter_pair(key, 3 + 3 * (depth - 2) + 1).y;
                    if (u >= q) alive = false;
                    else thr = thr / q;
                }
            }
            // loop condition of the bounce that starts now (path.cpp:33)
            if (alive && !((int) depth <= pp.max_depth || pp.max_depth < 0)) alive = false;
            if (alive && depth == 1 && si.emitter_id >= 0 && !pp.hide_emitters) {   // path.cpp:42-47
                res = res + thr * le_hit;
            }
            if (alive && (int) depth >= pp.max_depth && pp.max_depth > 0) alive = false;   // path.cpp:48-49
            if (alive) {
                const uint32_t pb = 3 + 3 * (depth - 1);
                spec refl = splat(0.f);                  // SmoothDiffuse::m_reflectance->eval(si) (diffuse.cpp:31,44): once per bounce
                if (DIFFUSE_ONLY) refl = srgb_model_eval(bs.a.z, bs.a.w, bs.b.x, wl) * bs.ior.w;      // * reflectance_scale
                else if (__float_as_int(bs.a.x) == 0) {
                    f3 c = mk3(bs.a.z, bs.a.w, bs.b.x);
                    float scale = bs.ior.w;
                    const uint32_t tex = __float_as_uint(bs.ior.z);
                    if (tex) { c = checkerboard_coeffs(tb, tex, hit); scale = 1.f; }
                    refl = spectrum_eval(tb, make_float4(c.x, c.y, c.z, scale), wl);      // (scale -1: a `regular` reflectance)
                }
                // ---- next-event estimation (path.cpp:56-67, scene.cpp:68-103)
                if (n_em > 0) {
                    f2 u = counter_pair(key, pb + 0);
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
                    if (!DIFFUSE_ONLY && (int) e == sc.env_emitter) {
                        // constant.cpp:53-72 (radiance at the path's wavelengths, oracle D8)
                        d = square_to_uniform_sphere(u);
                        dist = 2.f * sc.env_radius;
                        pdf = MSK_INV_FOUR_PI_F;
                        emitter_val = emitter_radiance(tb, (int) e, wl) / pdf;
                        nee_pdf = pdf;                                     // constant.cpp:74-76
                    } else {
                    const uint32_t first_face = __float_as_uint(e1.y), n_faces = __float_as_uint(e1.z);
                    const float *cdf = tb.cdf + __float_as_uint(e1.w);
                    // Distribution1D::sample_reuse (core/distribution.h:106-116)
                    uint32_t lo = 0, hi = n_faces + 1;
                    while (lo < hi) { uint32_t mid = (lo + hi) >> 1; if (!(u.y < cdf[mid])) lo = mid + 1; else hi = mid; }
                    int fidx = (int) lo - 1;
                    fidx = fidx < 0 ? 0 : fidx; fidx = fidx > (int) n_faces - 1 ? (int) n_faces - 1 : fidx;
                    u.y = (u.y - cdf[fidx]) / (cdf[fidx + 1] - cdf[fidx]);
                    const uint32_t lprim = first_face + (uint32_t) fidx;
                    const float4 la = tb.tri_verts[(size_t) lprim * 3], lb = tb.tri_verts[(size_t) lprim * 3 + 1],
                                 lc = tb.tri_verts[(size_t) lprim * 3 + 2];
                    const f3 p0 = mk3(la.x, la.y, la.z), p1 = mk3(lb.x, lb.y, lb.z), p2 = mk3(lc.x, lc.y, lc.z);
                    const f3 ed0 = p1 - p0, ed1 = p2 - p0;                 // mesh.cpp:103-133
                    const f2 bc = square_to_uniform_triangle(u);
                    const f3 lp = p0 + ed0 * bc.x + ed1 * bc.y;
                    const float4 lf = tb.tri_frames[(size_t) lprim * 3];
                    f3 ln = mk3(lf.x, lf.y, lf.z);                          // normalized(cross(ed0, ed1)), k_tri_frames
                    const int4 lmi = tb.mesh_info[__float_as_uint(la.w)];
                    if (lmi.z & 1) {
                        const float4 na = tb.tri_normals[(size_t) lprim * 3], nb = tb.tri_normals[(size_t) lprim * 3 + 1],
                                     nc = tb.tri_normals[(size_t) lprim * 3 + 2];
                        ln = normalized(mk3(na.x, na.y, na.z) * (1.f - bc.x - bc.y) + mk3(nb.x, nb.y, nb.z) * bc.x +
                                        mk3(nc.x, nc.y, nc.z) * bc.y);
                    }
                    pdf = e0.w;
                    d = lp - si.p;                                         // shape.cpp:64-78
                    const float dist2 = dot(d, d);
                    dist = __builtin_sqrtf(dist2);
                    d = d / dist;
                    const float dp = fabsf(dot(d, ln));
                    pdf *= (dp != 0.f) ? dist2 / dp : 0.f;
                    if (!DIFFUSE_ONLY) nee_pdf = e0.w * ((dp != 0.f) ? (dist * dist) / dp : 0.f);   // shape.cpp:80-86
                    if (dot(d, ln) < 0.f && pdf != 0.f) {                  // area.cpp:39-44
                        emitter_val = (one_emitter ? le_one : emitter_radiance(tb, (int) e, wl)) / pdf;
                    } else {
                        pdf = 0.f; emitter_val = splat(0.f);
                    }
                    }
                    if (n_em > 1) { pdf *= light_sel_pdf; emitter_val = emitter_val * (float) n_em; nee_pdf = nee_pdf * (1.f / n_em); }
                    if (pdf != 0.f) {
                        f3 wo = si.sh.to_local(d);
                        if (flipped) wo.z = -wo.z;
                        spec bsdf_val; float bsdf_pdf;
                        bsdf_eval_pdf<DIFFUSE_ONLY>(tb, bs, wi_s, wo, wl, refl, &bsdf_val, &bsdf_pdf);
                        const float w = mis_weight(pdf, bsdf_pdf);
                        contrib = thr * emitter_val * bsdf_val * w;
                        if (any_nonzero(contrib)) {
                            has_shadow = true;                             // scene.cpp:91-95
                            new_sh = make_float4(d.x, d.y, d.z, dist * (1.f - MSK_SHADOW_EPS_F));
                        }
                    }
                }
                // ---- BSDF sampling (path.cpp:71-80)
                {
                    const f2 u2 = counter_pair(key, pb + 2);
                    f3 wo_l; bool ok; float bs_eta;
                    const float sample1 = DIFFUSE_ONLY ? 0.f : counter_pair(key, pb + 1).x;
                    const spec bsdf_val = bsdf_sample<DIFFUSE_ONLY>(tb, bs, wi_s, sample1, u2, wl, refl, &wo_l, &bs_pdf, &bs_eta, &ok);
                    if (!ok) {
                        // failed sample: zero direction, the reference's ray misses and the loop ends (path.cpp:89-97) — but the
                        // NEE term of this bounce was added before the sample (path.cpp:60-66).  diffuse / roughconductor fail
                        // only with cos_i <= 0, where that term is 0; roughdielectric can fail (sample_ggx pdf == 0) with a
                        // non-zero one: the slot then lives one more iteration, dead, for its shadow ray.
                        if (!has_shadow) alive = false;
                        else {
                            thr = splat(0.f);
                            new_o = make_float4(si.p.x, si.p.y, si.p.z, (1.f + max_abs(si.p)) * MSK_RAY_EPS_F);
                            new_d = make_float4(0.f, 0.f, 0.f, -0.f);
                            depth += 1;
                        }
                    } else {
                        if (flipped) wo_l.z = -wo_l.z;
                        const f3 wo = si.sh.to_world(wo_l);
                        thr = thr * bsdf_val;                              // path.cpp:99
                        eta *= bs_eta;                                     // path.cpp:100
                        // A path whose throughput is now zero can only add zeros from here on; it lives one more
                        // iteration iff a shadow ray is pending (zero direction: the extension ray finds nothing).
                        const bool dead = !any_nonzero(thr);
                        if (dead && !has_shadow) alive = false;
                        new_o = make_float4(si.p.x, si.p.y, si.p.z, (1.f + max_abs(si.p)) * MSK_RAY_EPS_F);
                        new_d = dead ? make_float4(0.f, 0.f, 0.f, -bs_pdf) : make_float4(wo.x, wo.y, wo.z, -bs_pdf);
                        depth += 1;
                    }
                }
            }
        }
        if (depth > MSK_MAX_DEPTH) alive = false;      // 12 bits of depth in the state word (never reached: see MSK_MAX_DEPTH)
        // ---- finished paths: park them; 64 parked paths become records together (emit_record)
        {
            const bool fin = active && !alive;
            const unsigned long long fm = __ballot(fin);
            if (fm != 0ull) {
                if (fin) {
                    const uint32_t q = n_queued + (uint32_t) __popcll(fm & ((1ull << lane) - 1ull));
                    dq.wl[q] = to4(wl); dq.res[q] = to4(res); dq.id[q] = make_uint2(pix, s_own);
                }
                n_queued += (uint32_t) __popcll(fm);
                wave_sync();
                if (n_queued >= MSK_WAVE) {
                    const spec qwl = from4(dq.wl[lane]), qres = from4(dq.res[lane]);
                    const uint2 qid = dq.id[lane];
                    n_invalid += emit_record<DIFFUSE_ONLY>(sc, tb, pp, qwl, qres, qid.x, qid.y);
                    // move the rest (< 64 entries) to the front
                    const uint32_t rem = n_queued - MSK_WAVE;
                    float4 a = make_float4(0, 0, 0, 0), b = a; uint2 c4 = make_uint2(0, 0);
                    if (lane < rem) { a = dq.wl[MSK_WAVE + lane]; b = dq.res[MSK_WAVE + lane]; c4 = dq.id[MSK_WAVE + lane]; }
                    wave_sync();
                    if (lane < rem) { dq.wl[lane] = a; dq.res[lane] = b; dq.id[lane] = c4; }
                    wave_sync();
                    n_queued = rem;
                }
            }
        }
        // ---- compaction of the survivors into the other half, grouped (wave ballot + prefix popcount)
        const unsigned long long m_s = __ballot(alive && has_shadow), m_n = __ballot(alive && !has_shadow);
        n_done += __popcll(__ballot(active && !alive));
        if (alive) {
            const unsigned long long below = (1ull << lane) - 1ull;
            const uint32_t o = base_out + (has_shadow ? cur_s + (uint32_t) __popcll(m_s & below) : last - (cur_n + (uint32_t) __popcll(m_n & below)));
            st2<1>(st.id + o, make_uint2(pix, s_own | (depth << MSK_DEPTH_SHIFT)));
            st4<1>(st.wl + o, to4(wl)); st4<1>(st.thr + o, to4(thr)); st4<1>(st.res + o, to4(res));
            st4<4>(st.ray_o + o, new_o); st4<4>(st.ray_d + o, new_d);
            if (has_shadow) { st4<4>(st.sh + o, new_sh); st4<1>(st.contrib + o, to4(contrib)); }
            if (!DIFFUSE_ONLY) st.aux[o] = make_float2(eta, nee_pdf);
        }
        cur_s += (uint32_t) __popcll(m_s); cur_n += (uint32_t) __popcll(m_n);
    }

    // ---- the paths still parked (< 64)
    if (lane < n_queued) {
        const spec qwl = from4(dq.wl[lane]), qres = from4(dq.res[lane]);
        const uint2 qid = dq.id[lane];
        n_invalid += emit_record<DIFFUSE_ONLY>(sc, tb, pp, qwl, qres, qid.x, qid.y);       // (lane 0, which writes the counters back, is in here whenever anything is)
    }
    // ---- regeneration: fill the free tail with new camera samples (integrator.cpp:103-116)
    const uint32_t n_free = pp.region_size - (cur_s + cur_n);
    const unsigned long long first = rc.next_sample, left = rc.end_sample - rc.next_sample;
    const uint32_t got = (uint32_t) (left < n_free ? left : n_free);
    if (got > 0) {
    // the camera, the pixel table and the sample partition: read here, where they are used (MSK_COLD_KARGS)
    struct Cold { float s2c[16], to_world[16], near_clip, far_clip; int filter_border; const uint4 *pix_table; uint32_t spp_owned, n_regions; } k;
    if (MSK_COLD_KARGS) {
        const karg_ptr ka = karg_base();
#pragma unroll
        for (int q = 0; q < 16; ++q) { k.s2c[q] = karg<float>(ka, offsetof(DeviceScene, s2c) + 4 * q); k.to_world[q] = karg<float>(ka, offsetof(DeviceScene, to_world) + 4 * q); }
        k.near_clip = MSK_KARG_SC(ka, float, near_clip); k.far_clip = MSK_KARG_SC(ka, float, far_clip); k.filter_border = MSK_KARG_SC(ka, int32_t, filter_border);
        k.pix_table = MSK_KARG_PP(ka, const uint4 *, pix_table); k.spp_owned = MSK_KARG_PP(ka, uint32_t, spp_owned); k.n_regions = MSK_KARG_PP(ka, uint32_t, n_regions);
    } else {
#pragma unroll
        for (int q = 0; q < 16; ++q) { k.s2c[q] = sc.s2c[q]; k.to_world[q] = sc.to_world[q]; }
        k.near_clip = sc.near_clip; k.far_clip = sc.far_clip; k.filter_border = sc.filter_border;
        k.pix_table = pp.pix_table; k.spp_owned = pp.spp_owned; k.n_regions = pp.n_regions;
    }
    // the linear sample index -> (pass pixel, sample) split is a 32-bit division whenever this sweep's indices fit (a pass
    // of < 2^32 samples: every configuration but the largest single-GPU ones), the 64-bit one otherwise
    const bool idx32 = (((((first + got - 1) >> 6) * k.n_regions + wave) << 6) | 63ull) < (1ull << 32);
    for (uint32_t kk = lane; kk < got; kk += MSK_WAVE) {
        const unsigned long long q = first + kk;
        const unsigned long long sidx_lin = (((q >> 6) * k.n_regions + wave) << 6) | (q & 63ull);
        uint32_t j, si;
        if (idx32) { j = (uint32_t) sidx_lin / k.spp_owned; si = (uint32_t) sidx_lin - j * k.spp_owned; }
        else { j = (uint32_t) (sidx_lin / k.spp_owned); si = (uint32_t) (sidx_lin % k.spp_owned); }
        const uint4 pt = k.pix_table[j];
        const uint32_t pix = pt.x;
        const uint64_t key = counter_key(pp.seed, pix, pp.sample_first + si * pp.sample_stride);
        const f2 jit = counter_pair(key, 0);
        const float wsample = counter_pair(key, 1).x;
        const float px = (float) pixel_x(k.filter_border, pt) + jit.x, py = (float) pixel_y(k.filter_border, pt) + jit.y;
        spec wl;
#pragma unroll
        for (int q = 0; q < 4; ++q) { wl.v[q] = wavelength_of(wsample, q); __builtin_amdgcn_sched_barrier(0); }   // one fp64 chain at a time: interleaved they cost 28 VGPRs
        // PerspectiveCamera::sample_ray (perspective.cpp:22-42), Transform4f::apply_point/apply_vector
        float r4[4];
#pragma unroll
        for (int q = 0; q < 4; ++q)
            r4[q] = ((k.s2c[q * 4 + 0] * px + k.s2c[q * 4 + 1] * py) + k.s2c[q * 4 + 2] * 0.f) + k.s2c[q * 4 + 3] * 1.f;
        const f3 near_p = mk3(r4[0] / r4[3], r4[1] / r4[3], r4[2] / r4[3]);
        const f3 dl = normalized(near_p);
        const float inv_z = 1.f / dl.z;
        float o4[4];
#pragma unroll
        for (int q = 0; q < 4; ++q)
            o4[q] = ((k.to_world[q * 4 + 0] * 0.f + k.to_world[q * 4 + 1] * 0.f) + k.to_world[q * 4 + 2] * 0.f) +
                    k.to_world[q * 4 + 3] * 1.f;
        const f3 ow = mk3(o4[0] / o4[3], o4[1] / o4[3], o4[2] / o4[3]);
        const float *m = k.to_world;
        const f3 dw = mk3(m[0] * dl.x + (m[1] * dl.y + m[2] * dl.z), m[4] * dl.x + (m[5] * dl.y + m[6] * dl.z),
                          m[8] * dl.x + (m[9] * dl.y + m[10] * dl.z));
        const uint32_t o = base_out + (last - (cur_n + kk));        // the new samples carry no shadow ray: they continue that group
        st2<1>(st.id + o, make_uint2(pix, si | (1u << MSK_DEPTH_SHIFT)));
        st4<1>(st.wl + o, to4(wl));                                      // thr = 1, res = 0: RegionCtl::n_new
        st4<4>(st.ray_o + o, make_float4(ow.x, ow.y, ow.z, k.near_clip * inv_z));
        st4<4>(st.ray_d + o, make_float4(dw.x, dw.y, dw.z, k.far_clip * inv_z));
        if (!DIFFUSE_ONLY) st.aux[o] = make_float2(1.f, 0.f);
    }
    }
    const uint32_t n_out = cur_s + cur_n + got;
    rc.count = n_out; rc.half_ns = (cur_s << 1) | ((rc.half_ns & 1u) ^ 1u); rc.n_new = got;
    if (lane == 0) {
        rc.next_sample = first + got;
        rc.segments += n_out; rc.shadow_rays += cur_s; rc.samples_done += n_done; rc.invalid += n_invalid;
        pp.regions[wave] = rc;
    }
    return region_view(wave, pp.region_size, rc.count, rc.half_ns);
}

// this wave's done-queue: `base` + MSK_DONE_Q_F4 float4 per wave of the block
MSK_DEV DoneQueue done_queue(float4 *base) {
    DoneQueue dq;
    float4 *qbase = base + (threadIdx.x / MSK_WAVE) * MSK_DONE_Q_F4;
    dq.wl = qbase; dq.res = qbase + MSK_DONE_Q; dq.id = (uint2 *) (qbase + 2 * MSK_DONE_Q);
    return dq;
}

template <bool LDS_TABLES, bool DIFFUSE_ONLY, bool REGULAR = false>
MSK_DEV void shade_gen_body(const DeviceScene &sc, const PathState &st, const PassParams &pp) {
    extern __shared__ float4 lds_dyn[];
    typename std::conditional<REGULAR, SceneTablesR, SceneTables>::type tb;
    static_cast<SceneTables &>(tb) = stage_tables<LDS_TABLES>(sc, lds_dyn);
    const uint32_t lwave = (blockIdx.x * MSK_BLOCK + threadIdx.x) / MSK_WAVE;
    const uint32_t queue_f4 = LDS_TABLES ? tables_lds_float4s(sc) : small_tables_float4s(sc);   // after the staged tables
    const DoneQueue dq = done_queue(lds_dyn + queue_f4);
    SortScratch ss{nullptr, nullptr};
    if (!DIFFUSE_ONLY && pp.sort_scratch) {         // after the queues: per wave, 3 bytes per slot of a region (host: shade LDS plan)
        uint8_t *p = (uint8_t *) (lds_dyn + queue_f4 + (MSK_BLOCK / MSK_WAVE) * MSK_DONE_Q_F4) + (size_t) (threadIdx.x / MSK_WAVE) * 3u * pp.region_size;
        ss.perm = (uint16_t *) p; ss.cls = p + 2u * pp.region_size;
    }
    if (lwave >= pp.region_count) return;
    shade_region<DIFFUSE_ONLY>(sc, tb, dq, ss, st, pp, pp.region_first + lwave, threadIdx.x & (MSK_WAVE - 1));
}
template <bool LDS_TABLES, bool DIFFUSE_ONLY, bool REGULAR = false>
__global__ void __launch_bounds__(MSK_BLOCK)
k_shade_gen(DeviceScene sc, PathState st, PassParams pp) { shade_gen_body<LDS_TABLES, DIFFUSE_ONLY, REGULAR>(sc, st, pp); }
// The general variants sit a few registers above the 168 that three waves per SIMD allow (two cost 20 % of the shading time
// on the mesh scenes): the allocator is told to stay at three.
#ifndef MSK_SHADE_GEN_WAVES
#define MSK_SHADE_GEN_WAVES 3, 3
#endif
template <>
__global__ void __launch_bounds__(MSK_BLOCK) __attribute__((amdgpu_waves_per_eu(MSK_SHADE_GEN_WAVES)))
k_shade_gen<false, false>(DeviceScene sc, PathState st, PassParams pp) { shade_gen_body<false, false>(sc, st, pp); }
template <>
__global__ void __launch_bounds__(MSK_BLOCK) __attribute__((amdgpu_waves_per_eu(MSK_SHADE_GEN_WAVES)))
k_shade_gen<true, false>(DeviceScene sc, PathState st, PassParams pp) { shade_gen_body<true, false>(sc, st, pp); }
// ... and the same two for scenes with tabulated spectra
template <>
__global__ void __launch_bounds__(MSK_BLOCK) __attribute__((amdgpu_waves_per_eu(MSK_SHADE_GEN_WAVES)))
k_shade_gen<false, false, true>(DeviceScene sc, PathState st, PassParams pp) { shade_gen_body<false, false, true>(sc, st, pp); }
template <>
__global__ void __launch_bounds__(MSK_BLOCK) __attribute__((amdgpu_waves_per_eu(MSK_SHADE_GEN_WAVES)))
k_shade_gen<true, false, true>(DeviceScene sc, PathState st, PassParams pp) { shade_gen_body<true, false, true>(sc, st, pp); }
// The diffuse-only variants fit four waves per SIMD (128 VGPRs, no scratch); left alone, the allocator spends 24 more registers
// on the explicit fp64 fma chains of det_sincos and lands at three.
#ifndef MSK_NO_SHADE4
template <>
__global__ void __launch_bounds__(MSK_BLOCK) __attribute__((amdgpu_waves_per_eu(4, 4)))
k_shade_gen<false, true>(DeviceScene sc, PathState st, PassParams pp) { shade_gen_body<false, true>(sc, st, pp); }
template <>
__global__ void __launch_bounds__(MSK_BLOCK) __attribute__((amdgpu_waves_per_eu(4, 4)))
k_shade_gen<true, true>(DeviceScene sc, PathState st, PassParams pp) { shade_gen_body<true, true>(sc, st, pp); }
#endif

// ------------------------------------------------------------------------------------------
// k_wavefront: the iteration loop itself on the device, for scenes whose tree is staged in LDS.  A region is private to
// its wave — its slots, its share of the samples, its counters — so nothing orders one region's iterations against
// another's: every wave runs  shade -> trace -> shade -> ...  on its own region until the region has no path and no sample
// left (or `max_iters` sweeps have run: a bounded launch, the host relaunches while work remains).  Tables, tree and
// triangles are staged once per block instead of once per launch.  Same arithmetic as k_shade_gen / k_trace<0>: same records.
// ------------------------------------------------------------------------------------------
template <bool DIFFUSE_ONLY, bool REGULAR = false>
__global__ void __launch_bounds__(MSK_BLOCK)
k_wavefront(DeviceScene sc, PathState st, PassParams pp, uint32_t max_iters, uint32_t queue_f4, uint32_t trace_f4) {
    extern __shared__ float4 lds_dyn[];
    // LDS: [tables][done queues][traversal stacks][tree + triangles]; offsets in float4 from the host's plan
    typename std::conditional<REGULAR, SceneTablesR, SceneTables>::type tb;
    static_cast<SceneTables &>(tb) = stage_tables<true>(sc, lds_dyn);
    const DoneQueue dq = done_queue(lds_dyn + queue_f4);
    uint32_t *stack_base = (uint32_t *) (lds_dyn + trace_f4);
    float4 *scene_lds = lds_dyn + trace_f4 + (sc.stack_entries * MSK_BLOCK) / 4;
    const TraceLds g = stage_scene(sc, scene_lds, true, false);
    const LaneStack<false> stack{stack_base + threadIdx.x, nullptr, (int) sc.stack_entries, 0};
    const uint32_t lwave = (blockIdx.x * MSK_BLOCK + threadIdx.x) / MSK_WAVE;
    const uint32_t lane = threadIdx.x & (MSK_WAVE - 1);
    if (lwave >= pp.region_count) return;
    const uint32_t wave = pp.region_first + lwave;
    for (uint32_t it = 0; it < max_iters; ++it) {
        const RegionView rv = shade_region<DIFFUSE_ONLY>(sc, tb, dq, SortScratch{nullptr, nullptr}, st, pp, wave, lane);   // (the thin end of a pass: no sort)
        // the rays this wave has just written are read back by the same wave (other lanes): program order through the
        // CU's own L1 after the stores have drained
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        if (rv.n == 0) break;                                  // no live path, and regeneration found no sample to start
        if (MSK_THIN_RAYS_LDS && rv.n + rv.ns <= MSK_WAVE) trace_thin<0>(sc, st, g, rv, stack, lane, 4);      // one ray per lane (see trace_thin)
        else
        for (uint32_t c = lane; c < rv.n; c += MSK_WAVE) {
            const uint32_t i = rv.slot(c);
            const float4 ro = st.ray_o[i];
            float4 rd = st.ray_d[i];
            const bool has_shadow = c < rv.ns;
            rd.w = slot_tmax(rd.w);
            const f3 o = mk3(ro.x, ro.y, ro.z);
            float bt, bu, bv; uint32_t bp;
            uint32_t unocc = 0;
            if (has_shadow) {
                const float4 s = st.sh[i];
                const bool occ = traverse_scene<0, true>(sc, g, o, mk3(s.x, s.y, s.z), ro.w, s.w, stack, &bt, &bu, &bv, &bp);
                unocc = occ ? 0u : MSK_HIT_UNOCCLUDED;
            }
            traverse_scene<0, false>(sc, g, o, mk3(rd.x, rd.y, rd.z), ro.w, rd.w, stack, &bt, &bu, &bv, &bp);
            const bool valid = (bp != MSK_NO_PRIM) && (bt != rd.w);           // scene.cpp:234 tfar != maxt
            st.hit[i] = make_float4(valid ? bt : MSK_INF_F, bu, bv, __uint_as_float((valid ? bp : MSK_PRIM_MASK) | unocc));
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    }
}

// k_wavefront_h (round 6): the same device-side loop for scenes whose TREE STAYS IN HBM / L2 (trace mode 6: the 4-wide tree with
// half-float boxes) — for the thin end of a pass.  Measured on the mesh configs (profiles/r06_tail_*.txt): once every sample has
// been started the pool drains for ~50 more iterations (Russian roulette's tail) with a few dozen live paths per region, and
// every iteration is two launches that end when their SLOWEST wave ends: 17 of the config-5-class render's 123 ms, 12.6 of the
// config-3-class one's 145.  Here no wave waits for another region's stragglers: each runs shade -> trace -> shade -> ... on its
// own region until it is empty.  Tables: the small ones in LDS as in k_shade_gen<false, *>; rays: whole chunks through
// traverse4h (k_trace<6>'s walk — the arithmetic of k_trace_r<6>'s lanes, so the same hits; lane replacement has nothing to
// replace with when a region holds a chunk or two).
template <bool DIFFUSE_ONLY, bool REGULAR = false>
__global__ void __launch_bounds__(MSK_BLOCK)
k_wavefront_h(DeviceScene sc, PathState st, PassParams pp, uint32_t max_iters, uint32_t queue_f4, uint32_t trace_f4) {
    extern __shared__ float4 lds_dyn[];
    // LDS: [small tables][done queues][traversal stacks + four words per lane of node4h_step]; offsets in float4 from the host's plan
    typename std::conditional<REGULAR, SceneTablesR, SceneTables>::type tb;
    static_cast<SceneTables &>(tb) = stage_tables<false>(sc, lds_dyn);
    const DoneQueue dq = done_queue(lds_dyn + queue_f4);
    uint32_t *stack_base = (uint32_t *) (lds_dyn + trace_f4);
    const TraceLds g = stage_scene(sc, nullptr, false, false);
    const LaneStack<true> stack{stack_base + threadIdx.x, pp.stack_ovf + (size_t) blockIdx.x * MSK_BLOCK + threadIdx.x, (int) sc.stack_entries,
                                (size_t) gridDim.x * MSK_BLOCK, stack_base + sc.stack_entries * MSK_BLOCK + threadIdx.x * 4};
    const uint32_t lwave = (blockIdx.x * MSK_BLOCK + threadIdx.x) / MSK_WAVE;
    const uint32_t lane = threadIdx.x & (MSK_WAVE - 1);
    if (lwave >= pp.region_count) return;
    const uint32_t wave = pp.region_first + lwave;
    for (uint32_t it = 0; it < max_iters; ++it) {
        const RegionView rv = shade_region<DIFFUSE_ONLY>(sc, tb, dq, SortScratch{nullptr, nullptr}, st, pp, wave, lane);
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");    // the rays just written are read back by this wave's other lanes
        if (rv.n == 0) break;
        if (MSK_THIN_RAYS && rv.n + rv.ns <= MSK_WAVE) trace_thin<6>(sc, st, g, rv, stack, lane, 3);
        else
        for (uint32_t c = lane; c < rv.n; c += MSK_WAVE) {
            const uint32_t i = rv.slot(c);
            const float4 ro = st.ray_o[i];
            float4 rd = st.ray_d[i];
            const bool has_shadow = c < rv.ns;
            rd.w = slot_tmax(rd.w);
            const f3 o = mk3(ro.x, ro.y, ro.z);
            float bt, bu, bv; uint32_t bp;
            uint32_t unocc = 0;
            if (has_shadow) {
                const float4 s = st.sh[i];
                const bool occ = traverse_scene<6, true>(sc, g, o, mk3(s.x, s.y, s.z), ro.w, s.w, stack, &bt, &bu, &bv, &bp);
                unocc = occ ? 0u : MSK_HIT_UNOCCLUDED;
            }
            traverse_scene<6, false>(sc, g, o, mk3(rd.x, rd.y, rd.z), ro.w, rd.w, stack, &bt, &bu, &bv, &bp);
            const bool valid = (bp != MSK_NO_PRIM) && (bt != rd.w);           // scene.cpp:234 tfar != maxt
            st.hit[i] = make_float4(valid ? bt : MSK_INF_F, bu, bv, __uint_as_float((valid ? bp : MSK_PRIM_MASK) | unocc));
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    }
}

// AOVIntegrator::sample's primary-hit channels (aov.cpp:89-122): runs after k_trace, picks the slots whose camera ray
// has just been traced (depth 1) and writes their record groups.  A miss writes zeros.
__global__ void __launch_bounds__(MSK_BLOCK)
k_aov_primary(DeviceScene sc, PathState st, PassParams pp, AovParams ap) {
    const uint32_t lwave = (blockIdx.x * MSK_BLOCK + threadIdx.x) / MSK_WAVE;
    const uint32_t lane = threadIdx.x & (MSK_WAVE - 1);
    if (lwave >= pp.region_count) return;
    const uint32_t wave = pp.region_first + lwave;
    const RegionView rv = region_view(wave, pp.region_size, pp.regions[wave].count, pp.regions[wave].half_ns);
    for (uint32_t c = lane; c < rv.n; c += MSK_WAVE) {
        const uint32_t i = rv.slot(c);
        const uint2 id = st.id[i];
        if ((id.y >> MSK_DEPTH_SHIFT) != 1u) continue;
        const float4 hit = st.hit[i];
        float val[13];
#pragma unroll
        for (int k = 0; k < 13; ++k) val[k] = 0.f;
        if (hit.x != MSK_INF_F) {
            const uint32_t prim = __float_as_uint(hit.w) & MSK_PRIM_ID;
            const float4 a = sc.tri_verts[(size_t) prim * 3], b = sc.tri_verts[(size_t) prim * 3 + 1], cc = sc.tri_verts[(size_t) prim * 3 + 2];
            const int4 mi = sc.mesh_info[__float_as_uint(a.w)];
            const f3 p0 = mk3(a.x, a.y, a.z), p1 = mk3(b.x, b.y, b.z), p2 = mk3(cc.x, cc.y, cc.z);
            const float b1 = hit.y, b2 = hit.z, b0 = 1.f - b1 - b2;                 // mesh.cpp:53-65
            const f3 p = p0 * b0 + p1 * b1 + p2 * b2;
            const f3 n = normalized(cross(p1 - p0, p2 - p0));
            f3 ns = n;
            float u = hit.y, v = hit.z;
            if (mi.z & 2) {                                                         // mesh.cpp:68-71
                const float4 ua = sc.tri_uvs[(size_t) prim * 2], ub = sc.tri_uvs[(size_t) prim * 2 + 1];
                u = ua.x * b0 + ua.z * b1 + ub.x * b2; v = ua.y * b0 + ua.w * b1 + ub.y * b2;
            }
            if (mi.z & 1) {                                                         // mesh.cpp:81-96
                const float4 na = sc.tri_normals[(size_t) prim * 3], nb = sc.tri_normals[(size_t) prim * 3 + 1],
                             nc = sc.tri_normals[(size_t) prim * 3 + 2];
                ns = normalized(mk3(na.x, na.y, na.z) * b0 + mk3(nb.x, nb.y, nb.z) * b1 + mk3(nc.x, nc.y, nc.z) * b2);
            }
            val[1] = hit.x; val[2] = p.x; val[3] = p.y; val[4] = p.z; val[5] = u; val[6] = v;
            val[7] = n.x; val[8] = n.y; val[9] = n.z; val[10] = ns.x; val[11] = ns.y; val[12] = ns.z;
        }
        const uint32_t pix = id.x, s_own = id.y & MSK_SI_MASK, j = pp.pix_to_j[pix];
        const uint64_t key = counter_key(pp.seed, pix, pp.sample_first + s_own * pp.sample_stride);
        const f2 jit = counter_pair(key, 0);
        const float px = (float) (pix % (uint32_t) sc.width) + jit.x;
        const size_t r = (size_t) j * pp.spp_owned + s_own;
        uint32_t wxw = 0;
        if (pp.packed) wxw = sample_weights(sc, pp.pix_table[j], px, (float) (pix / (uint32_t) sc.width) + jit.y).x;
        for (uint32_t g = 0; g < ap.n_groups; ++g) {
            const uint32_t code = ap.code[g];
            float o[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const uint32_t sel = (code >> (8 * k)) & 0xffu;
                float x = 0.f;
#pragma unroll
                for (int q = 1; q < 13; ++q) x = sel == (uint32_t) q ? val[q] : x;
                o[k] = x;
            }
            ap.rec[g][r] = make_float4(o[0], o[1], o[2], pp.packed ? __uint_as_float(wxw | nonfinite_flag(o[0], o[1], o[2])) : px);
        }
    }
}

// sums the per-region records for the host (termination test + statistics): each block reduces its slice and adds six
// totals to *out (zeroed by the host before the launch) — 6 atomics per block, a few dozen per launch
__global__ void __launch_bounds__(MSK_BLOCK) k_reduce_ctl(const RegionCtl *regions, uint32_t n_regions, Ctrl *out) {
    __shared__ unsigned long long sh[6][MSK_BLOCK];
    unsigned long long a[6] = {0, 0, 0, 0, 0, 0};
    for (uint32_t i = blockIdx.x * MSK_BLOCK + threadIdx.x; i < n_regions; i += gridDim.x * MSK_BLOCK) {
        const RegionCtl r = regions[i];
        a[0] += r.count; a[1] += r.end_sample - r.next_sample; a[2] += r.segments; a[3] += r.shadow_rays; a[4] += r.samples_done; a[5] += r.invalid;
    }
    for (int k = 0; k < 6; ++k) sh[k][threadIdx.x] = a[k];
    __syncthreads();
    for (uint32_t s = MSK_BLOCK / 2; s > 0; s >>= 1) {
        if (threadIdx.x < s) for (int k = 0; k < 6; ++k) sh[k][threadIdx.x] += sh[k][threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        atomicAdd(&out->live, sh[0][0]); atomicAdd(&out->remaining, sh[1][0]); atomicAdd(&out->segments, sh[2][0]);
        atomicAdd(&out->shadow_rays, sh[3][0]); atomicAdd(&out->samples_done, sh[4][0]); atomicAdd(&out->invalid, sh[5][0]);
    }
}

// ------------------------------------------------------------------------------------------
// film: records -> per-block ImageBlocks -> film, in the reference's accumulation order
// ------------------------------------------------------------------------------------------
struct BlockInfo {          // one spiral block of this pass
    int32_t off_x, off_y, size_x, size_y;
    uint32_t pixel_base;    // first pass-pixel index of the block (row-major inside the block)
    uint32_t slot;          // index of the block's accumulation buffer
};

// One thread per TX x TY tile of pixels of the bordered block: replays ImageBlock::put
// (imageblock.cpp:55-114) for every sample of the block in the order render_block issues them
// (y, x, s — integrator.cpp:89-98) and keeps what lands on its own pixels.  Every pixel therefore
// receives exactly the additions, in exactly the order, the scalar loop performs.  A tile shares
// one read of each record among its pixels ((TX+4)(TY+4)/(TX*TY) reads per pixel instead of 25).
template <int TX, int TY>
__global__ void __launch_bounds__(MSK_BLOCK)
k_resolve_blocks(DeviceScene sc, const BlockInfo *blocks, uint32_t n_blocks, const float4 *rec_a, const float *rec_b,
                 uint32_t spp_owned, float *block_buf, uint32_t buf_stride, int tiles_x, int tiles_y) {
    __shared__ float lut[36];
    if (threadIdx.x < 33) lut[threadIdx.x] = sc.lut[threadIdx.x];
    __syncthreads();
    const uint32_t per_block = (uint32_t) (tiles_x * tiles_y);
    const uint64_t gid = (uint64_t) blockIdx.x * MSK_BLOCK + threadIdx.x;
    const uint32_t bi = (uint32_t) (gid / per_block), t = (uint32_t) (gid % per_block);
    if (bi >= n_blocks) return;
    const BlockInfo b = blocks[bi];
    const int border = sc.filter_border;
    const int sx = b.size_x + 2 * border, sy = b.size_y + 2 * border;
    const int tx0 = (int) (t % (uint32_t) tiles_x) * TX, ty0 = (int) (t / (uint32_t) tiles_x) * TY;
    if (tx0 >= sx || ty0 >= sy) return;
    const float radius = sc.filter_radius, scale = sc.filter_scale;
    // a sample of source pixel x sits at bordered position px in [x + border - .5, x + border + .5] (x + u in fp32 may round
    // up to x + 1) and lands on the integer targets t with |t - px| <= r, i.e. |t - border - x| <= floor(r + .5): for the
    // Gaussian's r = 2 a target gathers from 5 x 5 source pixels
    const int span = (int) floorf(radius + 0.5f);
    // channels X, Y, Z and one accumulator for A and W (both receive w * 1, integrator.cpp:119-123: identical sums)
    float acc[TY][TX][4];
#pragma unroll
    for (int j = 0; j < TY; ++j)
#pragma unroll
        for (int i = 0; i < TX; ++i)
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[j][i][c] = 0.f;
    const int y_lo = max(0, ty0 - border - span), y_hi = min(b.size_y - 1, ty0 + TY - 1 - border + span);
    const int x_lo = max(0, tx0 - border - span), x_hi = min(b.size_x - 1, tx0 + TX - 1 - border + span);
    const float offx = (float) (b.off_x - border), offy = (float) (b.off_y - border);
    constexpr int U = 8;            // records in flight per lane
    for (int y = y_lo; y <= y_hi; ++y)
        for (int x = x_lo; x <= x_hi; ++x) {
            const size_t r0 = ((size_t) b.pixel_base + (uint32_t) (y * b.size_x + x)) * spp_owned;
            for (uint32_t s0 = 0; s0 < spp_owned; s0 += U) {
                float4 ra[U]; float rb[U];
#pragma unroll
                for (int k = 0; k < U; ++k) {
                    const uint32_t s = min(s0 + (uint32_t) k, spp_owned - 1);
                    ra[k] = rec_a[r0 + s];
                    rb[k] = rec_b[r0 + s];
                }
#pragma unroll
                for (int k = 0; k < U; ++k) {
                    if (s0 + (uint32_t) k >= spp_owned) break;
                    const float px = ra[k].w - 0.5f - offx, py = rb[k] - 0.5f - offy;
                    // imageblock.cpp:66-72: lo = max(ceil(pos - r), 0), hi = min(floor(pos + r), size - 1).  For an integer
                    // target t inside the block, lo <= t <= hi  <=>  (pos - r) <= t <= (pos + r) with the same fp32 sums.
                    const float xm = px - radius, xp = px + radius, ym = py - radius, yp = py + radius;
                    float wx[TX], wy[TY];
                    bool inx[TX], iny[TY];
#pragma unroll
                    for (int i = 0; i < TX; ++i) {
                        const float ft = (float) (tx0 + i);
                        wx[i] = lut[min((int) fabsf((ft - px) * scale), 32)];
                        inx[i] = ft >= xm && ft <= xp;
                    }
#pragma unroll
                    for (int j = 0; j < TY; ++j) {
                        const float ft = (float) (ty0 + j);
                        wy[j] = lut[min((int) fabsf((ft - py) * scale), 32)];
                        iny[j] = ft >= ym && ft <= yp;
                    }
#pragma unroll
                    for (int j = 0; j < TY; ++j)
#pragma unroll
                        for (int i = 0; i < TX; ++i) {
                            if (!(inx[i] && iny[j])) continue;
                            const float w = wx[i] * wy[j];
                            acc[j][i][0] += w * ra[k].x; acc[j][i][1] += w * ra[k].y; acc[j][i][2] += w * ra[k].z;
                            acc[j][i][3] += w * 1.f;
                        }
                }
            }
        }
#pragma unroll
    for (int j = 0; j < TY; ++j)
#pragma unroll
        for (int i = 0; i < TX; ++i) {
            const int tx = tx0 + i, ty = ty0 + j;
            if (tx >= sx || ty >= sy) continue;
            float *o = block_buf + (size_t) b.slot * buf_stride + (size_t) (ty * sx + tx) * 5;
            o[0] = acc[j][i][0]; o[1] = acc[j][i][1]; o[2] = acc[j][i][2]; o[3] = acc[j][i][3]; o[4] = acc[j][i][3];
        }
}

// ------------------------------------------------------------------------------------------
// k_resolve_rows: the same replay for the default filter (border 2, radius 2) from records that carry their weights
// (SampleWeights), reading each record ~3 times instead of ~12.
//
// Order is what makes the film bit-identical to the scalar loop: a target pixel must receive its source pixels in row-major
// order and each source pixel's samples in sample order (integrator.cpp:89-98 + imageblock.cpp:98-110).  Row-major order means
// that while source row r is being replayed exactly five target rows are live (r .. r + 4 of the bordered block) and all of them
// take row r's samples — so one wave sweeps a band of target rows DOWN the block: 12 x 5 lanes, a lane owns three adjacent
// target columns of one live row (lane row = target row mod 5; a finished row's lanes move on to the row five below).  A round
// = one source row: for each of the seven source columns a lane's three targets can see (kx, ascending), the wave stages 64
// samples of that column for all 12 lane columns in LDS — one coalesced kilobyte per column, shared by the five lane rows —
// and every lane adds them to the targets in reach (a static set per kx) in sample order.  Re-read factor: 7/3 in x (a source
// column is staged for up to three lane columns at different kx) x (rounds / source rows) in y, against 5 x 7/3 before.
// A target's additions are exactly ImageBlock::put's: weight = lut[ix] * lut[iy] (imageblock.cpp:103), += weight * value
// (:106); targets outside the sample's footprint add nothing.
// ------------------------------------------------------------------------------------------
struct RowBand { uint32_t block, row_begin, row_end, pad; };      // target rows [row_begin, row_end) of the bordered block
#define MSK_RR_COLS 12            /* lane columns: 3 target columns each -> bordered width <= 36 (block size <= 32) */
#ifndef MSK_RR_CHUNK
#define MSK_RR_CHUNK 64           /* samples staged per step (<= 64: one per lane) */
#endif
#define MSK_RR_STRIDE (MSK_RR_CHUNK + 1)   /* slot stride in records: consecutive slots land on different LDS banks */
#ifndef MSK_RR_G
#define MSK_RR_G 8                 /* samples per accumulate group */
#endif

typedef float msk_f2 __attribute__((ext_vector_type(2)));      // pairs for v_pk_mul_f32 / v_pk_add_f32: two IEEE operations each
template <bool SAFE>
MSK_DEV void rr_accumulate(const float4 *la, const uint2 *lw, const float *lut, uint32_t shy, int kx, msk_f2 (&acc)[3][2]) {
    // in reach of source column 3 lx - 4 + kx: target j of this lane with field f = j + 4 - kx in 0..4.
    // Groups of MSK_RR_G samples, the next group's records read from LDS while this one is accumulated (a wave often has its
    // SIMD to itself here: nothing else would cover the LDS round trips record -> index -> weight).
    // A staged record is {X, Y, Z, 1} + {x word, y word}: weight * {X, Y} and weight * {Z, 1} are two packed multiplies and the
    // four channel sums two packed adds — the same four products and four sums as imageblock.cpp:106, channel by channel.
    constexpr int G = MSK_RR_G;
    float4 ra[G]; uint2 rw[G];
#pragma unroll
    for (int k = 0; k < G; ++k) { ra[k] = la[k]; rw[k] = lw[k]; }
#pragma unroll 1
    for (int s0 = 0; s0 < MSK_RR_CHUNK; s0 += G) {
        float4 na[G]; uint2 nw[G];
        const int s1 = (s0 + G < MSK_RR_CHUNK) ? s0 + G : s0;          // (the last group re-reads itself: no branch in the loop)
#pragma unroll
        for (int k = 0; k < G; ++k) { na[k] = la[s1 + k]; nw[k] = lw[s1 + k]; }
        float wy[G];
#pragma unroll
        for (int k = 0; k < G; ++k) wy[k] = *(const float *) ((const char *) lut + ((rw[k].y >> shy) & 0xfcu));
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int f = j + 4 - kx;
            if (f < 0 || f > 4) continue;
            float wx[G];
#pragma unroll
            for (int k = 0; k < G; ++k) wx[k] = *(const float *) ((const char *) lut + ((rw[k].x >> (6 * f)) & 0xfcu));
#pragma unroll
            for (int k = 0; k < G; ++k) {
                const float w = wx[k] * wy[k];
                if (SAFE) {
                    // a record with inf / nan values: the scalar loop never multiplies them for targets outside the footprint
                    if (((rw[k].x >> (6 * f)) & 0xfcu) == (MSK_W_OUT << MSK_W_SHIFT) || ((rw[k].y >> shy) & 0xfcu) == (MSK_W_OUT << MSK_W_SHIFT)) continue;
                }
                const msk_f2 ww = {w, w}, xy = {ra[k].x, ra[k].y}, z1 = {ra[k].z, ra[k].w};
                acc[j][0] += ww * xy; acc[j][1] += ww * z1;
            }
        }
#pragma unroll
        for (int k = 0; k < G; ++k) { ra[k] = na[k]; rw[k] = nw[k]; }
    }
}

__global__ void __launch_bounds__(MSK_WAVE)
k_resolve_rows(DeviceScene sc, const BlockInfo *blocks, const RowBand *bands, uint32_t n_bands, const float4 *rec_a,
               const uint32_t *rec_b, uint32_t spp_owned, float *block_buf, uint32_t buf_stride) {
    __shared__ float4 lds_a[MSK_RR_COLS * MSK_RR_STRIDE];
    __shared__ uint2 lds_w[MSK_RR_COLS * MSK_RR_STRIDE];
    __shared__ float lut[36];
    const uint32_t lane = threadIdx.x;
    if (lane < 36) lut[lane] = lane < 33 ? sc.lut[lane] : 0.f;
    if (blockIdx.x >= n_bands) return;
    const RowBand band = bands[blockIdx.x];
    const BlockInfo b = blocks[band.block];
    const int lx = (int) (lane % MSK_RR_COLS), lr = (int) (lane / MSK_RR_COLS);      // lanes 60..63 (lr = 5) only help staging
    const int sx_t = b.size_x + 4;                                                    // bordered width
    const int r_begin = max(0, (int) band.row_begin - 4), r_end = min(b.size_y - 1, (int) band.row_end - 1);
    msk_f2 acc[3][2];                                         // per target {X, Y} {Z, W}
#pragma unroll
    for (int j = 0; j < 3; ++j) { acc[j][0] = msk_f2{0.f, 0.f}; acc[j][1] = msk_f2{0.f, 0.f}; }
    const float4 *my_a = lds_a + lx * MSK_RR_STRIDE;
    const uint2 *my_w = lds_w + lx * MSK_RR_STRIDE;
    const uint32_t n_chunks = (spp_owned + MSK_RR_CHUNK - 1) / MSK_RR_CHUNK;
    // One step = (source row r, source-column offset kx, chunk c of 64 samples), in exactly that nesting order.  The records
    // of step i + 1 are fetched into registers while step i is accumulated out of LDS (one wave owns its LDS: the only
    // synchronisation is between the lanes of this wave).
    float4 pa[MSK_RR_COLS]; uint32_t pb[MSK_RR_COLS];
    bool p_lane_ok = false; int p_kx = 0;
    // Unconditional loads from clamped addresses (a conditional load would have to be waited for at the join, one column at a
    // time); what is not a real record — a column outside the block, a sample index past the last one — is replaced when the
    // step is stored to LDS.
    auto fetch = [&](int r, int kx, uint32_t c) {
        const uint32_t sidx = c * MSK_RR_CHUNK + lane % MSK_RR_CHUNK;      // (lanes past the chunk repeat addresses: no extra traffic)
        p_lane_ok = sidx < spp_owned && lane < MSK_RR_CHUNK; p_kx = kx;
        const size_t row_rec = ((size_t) b.pixel_base + (size_t) r * b.size_x) * spp_owned + min(sidx, spp_owned - 1u);
#pragma unroll
        for (int q = 0; q < MSK_RR_COLS; ++q) {
            const int sx = min(max(3 * q - 4 + kx, 0), b.size_x - 1);
            const size_t rr = row_rec + (size_t) sx * spp_owned;
            pa[q] = rec_a[rr]; pb[q] = rec_b[rr];
        }
    };
    int r = r_begin, kx = 0; uint32_t c = 0;
    if (r_begin > r_end) return;
    fetch(r, kx, c);
    wave_sync();                                              // lut
    int m = 0, ty = 0; uint32_t shy = 0; bool mine = false;
    while (r <= r_end) {
        if (kx == 0 && c == 0) {                              // a round begins
            m = ((lr - r) % 5 + 5) % 5;                       // this lane's live target row is r + m; its y field is m
            ty = r + m;
            mine = lr < 5 && ty >= (int) band.row_begin && ty < (int) band.row_end;
            shy = 6u * (uint32_t) m;
            if (m == 4 || r == r_begin) {                     // a new target row starts here
#pragma unroll
                for (int j = 0; j < 3; ++j) { acc[j][0] = msk_f2{0.f, 0.f}; acc[j][1] = msk_f2{0.f, 0.f}; }
            }
        }
        // ---- this step's records: registers -> LDS as {X, Y, Z, 1} + {x word, y word}; a flagged record anywhere sends the
        //      whole step down the exact path
        uint32_t flags = 0;
#pragma unroll
        for (int q = 0; q < MSK_RR_COLS; ++q) {
            const int sx = 3 * q - 4 + p_kx;
            float4 va = pa[q]; uint32_t vb = pb[q];
            if (!(p_lane_ok && sx >= 0 && sx < b.size_x)) {   // no such column / sample: zero weights (five fields of MSK_W_OUT), zero values
                va = make_float4(0.f, 0.f, 0.f, __uint_as_float(MSK_W_ALL_OUT));
                vb = MSK_W_ALL_OUT;
            }
            const uint32_t wxw = __float_as_uint(va.w);
            flags |= wxw;
            if (MSK_RR_CHUNK == MSK_WAVE || lane < MSK_RR_CHUNK) {
                lds_a[q * MSK_RR_STRIDE + lane] = make_float4(va.x, va.y, va.z, 1.f);
                lds_w[q * MSK_RR_STRIDE + lane] = make_uint2(wxw, vb);
            }
        }
        const bool safe = __ballot((flags & MSK_W_NONFINITE) != 0u) != 0ull;
        wave_sync();
        // ---- the next step's records, in flight while this one is accumulated
        const int kx_now = kx;
        const bool round_ends = kx == 6 && c + 1 == n_chunks;
        const int r_now = r;
        if (++c == n_chunks) { c = 0; if (++kx == 7) { kx = 0; ++r; } }
        fetch(min(r, r_end), kx, c);                         // (past the last step: a valid address, never stored)
        switch (kx_now) {
#define MSK_RR_CASE(K) case K: if (safe) rr_accumulate<true>(my_a, my_w, lut, shy, K, acc); else rr_accumulate<false>(my_a, my_w, lut, shy, K, acc); break;
            MSK_RR_CASE(0) MSK_RR_CASE(1) MSK_RR_CASE(2) MSK_RR_CASE(3) MSK_RR_CASE(4) MSK_RR_CASE(5) MSK_RR_CASE(6)
#undef MSK_RR_CASE
        }
        wave_sync();                                          // every lane has read this step before the next one is stored
        // ---- a target row is complete after its last source row (ty == r), or when the block's source rows end
        if (round_ends && mine && (m == 0 || r_now == r_end)) {
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int tx = 3 * lx + j;
                if (tx >= sx_t) continue;
                float *o = block_buf + (size_t) b.slot * buf_stride + (size_t) (ty * sx_t + tx) * 5;
                o[0] = acc[j][0].x; o[1] = acc[j][0].y; o[2] = acc[j][1].x; o[3] = acc[j][1].y; o[4] = acc[j][1].y;
            }
        }
    }
}

// Film::put for every block in spiral order (imageblock.cpp:36-53,133-173; D6: ascending block id).
// One thread per pixel of the film's crop window; block_of[by*nbx+bx] = slot of that block's buffer or -1 (not rendered
// by this rank), spiral_id gives the order.
__global__ void __launch_bounds__(MSK_BLOCK)
k_film_put(DeviceScene sc, const BlockInfo *blocks, const int32_t *block_of, const uint32_t *spiral_id, int nbx, int nby,
           int block_size, const float *block_buf, uint32_t buf_stride, FilmOut out) {
    const uint32_t gid = blockIdx.x * MSK_BLOCK + threadIdx.x;            // a pixel of the crop window (the film that is written)
    if (gid >= (uint32_t) (sc.crop_w * sc.crop_h)) return;
    const int x = sc.crop_x + (int) (gid % (uint32_t) sc.crop_w), y = sc.crop_y + (int) (gid / (uint32_t) sc.crop_w);
    const int border = sc.filter_border;
    int cand[9]; uint32_t cid[9]; int nc = 0;
    const int bx0 = x / block_size, by0 = y / block_size;
    for (int dy = -1; dy <= 1; ++dy)
        for (int dx = -1; dx <= 1; ++dx) {
            const int bx = bx0 + dx, by = by0 + dy;
            if (bx < 0 || by < 0 || bx >= nbx || by >= nby) continue;
            const int s = block_of[by * nbx + bx];
            if (s < 0) continue;
            const BlockInfo b = blocks[s];
            const int lx = x - (b.off_x - border), ly = y - (b.off_y - border);
            if (lx < 0 || ly < 0 || lx >= b.size_x + 2 * border || ly >= b.size_y + 2 * border) continue;
            // insertion sort by spiral id
            const uint32_t id = spiral_id[by * nbx + bx];
            int k = nc++;
            while (k > 0 && cid[k - 1] > id) { cid[k] = cid[k - 1]; cand[k] = cand[k - 1]; --k; }
            cid[k] = id; cand[k] = s;
        }
    float acc[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < nc; ++k) {
        const BlockInfo b = blocks[cand[k]];
        const int lx = x - (b.off_x - border), ly = y - (b.off_y - border);
        const float *src = block_buf + (size_t) b.slot * buf_stride + (size_t) (ly * (b.size_x + 2 * border) + lx) * 5;
#pragma unroll
        for (int c = 0; c < 5; ++c) acc[c] += src[c];
    }
    float *o = out.film + (size_t) gid * out.stride;
#pragma unroll
    for (int c = 0; c < 5; ++c) if (out.ch[c] >= 0) o[out.ch[c]] = acc[c];
}

// records of listed pixels (unpacked: they carry positions) -> {X,Y,Z} + position, [pixel][sample] on both sides
__global__ void k_export_records(const float4 *rec_a, const float *rec_b, uint64_t n_pix, uint32_t spp, float *out_xyz,
                                 float *out_pos) {
    const uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_pix * spp) return;
    const uint64_t o = i;
    const float4 a = rec_a[i];
    out_xyz[o * 3] = a.x; out_xyz[o * 3 + 1] = a.y; out_xyz[o * 3 + 2] = a.z;
    if (out_pos) { out_pos[o * 2] = a.w; out_pos[o * 2 + 1] = rec_b[i]; }
}

}  // namespace msk
