// msk_gpu.hip — implementation of the C ABI in include/msk_gpu.h (MI355X / gfx950 only).
//
// Host responsibilities of this file: scene upload + BVH build (replaces Scene::accel_init,
// src/librender/scene.cpp:201-212), area-light tables (mesh.cpp:39-48), the spiral block
// schedule (imageblock.cpp:176-247), the wavefront iteration loop, and the ordered film
// resolve.  Everything per-sample runs in the kernels of msk_kernels.h.
#include "msk_kernels.h"
#include "msk_serial.h"
#include "msk_bvh.h"
#include "msk_lbvh.h"
#define MSK_WATCHDOG_SYNC
#include "msk_watchdog.h"
#include "../../include/msk_gpu.h"
#include <hip/hip_ext.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

using namespace msk;

static thread_local std::string g_last_error;

#define MSK_MAX_GROUP 8                 /* devices behind one context (msk_multi.h) */
struct msk_group;
struct msk_ctx {
    msk_group *group = nullptr;        // non-null: a group context (msk_gpu_init with n > 1); the fields below describe its first device
    int device = 0;
    hipStream_t stream = nullptr;
    hipDeviceProp_t prop;
    std::string last_error;
#ifndef MSK_MAX_STREAMS
#define MSK_MAX_STREAMS 4
#endif
    hipStream_t more_streams[MSK_MAX_STREAMS - 1] = {};   // the other parts of the pool run here (run_wavefront)
    Ctrl *h_ctrl = nullptr;            // pinned, [MSK_MAX_STREAMS]: one per part
    std::vector<hipEvent_t> events, more_events[MSK_MAX_STREAMS - 1];
    uint32_t timing_phase = 0;         // which sync groups carry timing events rotates from render to render (MSK_TIMING_EVERY);
                                       // a context is used by one host thread at a time (msk_gpu.h), so a plain counter
    bool lost = false;                 // the watchdog gave up on a render (msk_watchdog.h): every later call fails at once, shutdown releases host memory only
    mskwd::Hub *hub = nullptr;         // what the host functions of MSK_WAIT=callback signal; leaked with a lost context (msk_watchdog.h)
    hipEvent_t wait_events[MSK_MAX_STREAMS] = {};      // MSK_WAIT=event: hipEventBlockingSync events, one per part of the pool
    uint32_t device_sharers = 1;       // member contexts of a group that sit on this context's device (repeated ordinals): their
                                       // renders run side by side, each plans its record buffers within 1 / device_sharers of the free HBM
};

static int fail(msk_ctx *ctx, int code, const char *fmt, ...) {
    char buf[1024];
    va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
    g_last_error = buf;
    if (ctx) ctx->last_error = buf;
    return code;
}
// The same for code that runs on the helper threads of a render (run_wavefront): the text goes to a slot the thread
// owns; the calling thread copies the first failure into the context after the join.  ctx->last_error is only ever
// written by the thread that called into the library.
static int fail_to(std::string *slot, int code, const char *fmt, ...) {
    char buf[1024];
    va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
    *slot = buf;
    return code;
}
#define HIP_TRY_SLOT(slot, expr)                                                                 \
    do { hipError_t e_ = (expr); if (e_ != hipSuccess)                                          \
        return fail_to(slot, e_ == hipErrorOutOfMemory ? MSK_ERR_OOM : MSK_ERR_HIP, "%s failed: %s (%s:%d)", #expr, \
                       hipGetErrorString(e_), __FILE__, __LINE__); } while (0)
#define HIP_TRY(ctx, expr)                                                                       \
    do { hipError_t e_ = (expr); if (e_ != hipSuccess)                                          \
        return fail(ctx, e_ == hipErrorOutOfMemory ? MSK_ERR_OOM : MSK_ERR_HIP, "%s failed: %s (%s:%d)", #expr, \
                    hipGetErrorString(e_), __FILE__, __LINE__); } while (0)

// a context the watchdog gave up on (msk_watchdog.h): nothing runs on it any more
#define MSK_REFUSE_LOST(ctx) do { if ((ctx) && (ctx)->lost) return fail(ctx, MSK_ERR_HIP, "this context is lost: an earlier render made no progress " \
                                                                         "(msk_watchdog.h); shut it down and start over in a new process"); } while (0)
struct DevBuf {
    void *p = nullptr; size_t bytes = 0;
    ~DevBuf() { if (p) (void) hipFree(p); }
    hipError_t alloc(size_t n) { if (p) (void) hipFree(p); p = nullptr; bytes = n; return n ? hipMalloc(&p, n) : hipSuccess; }
    hipError_t reserve(size_t n) { return (p && bytes >= n) ? hipSuccess : alloc(n); }     // grow-only (workspace reuse)
    template <typename T> hipError_t upload(const std::vector<T> &v) {
        hipError_t e = reserve(std::max<size_t>(v.size() * sizeof(T), 16));
        if (e != hipSuccess) return e;
        return v.empty() ? hipSuccess : hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice);
    }
    template <typename T> T *as() const { return (T *) p; }
    void leak() { p = nullptr; bytes = 0; }        // a lost context: hipFree would wait for a kernel that never finishes; the memory goes with the process
};

static uint32_t env_u32(const char *name, uint32_t def);
// width of the tree for scenes that do not fit LDS: 4 (full-precision boxes) or 8 (quantised boxes); MSK_WIDE_BVH overrides, 0 = binary
#ifndef MSK_WIDE_BVH_DEFAULT
#define MSK_WIDE_BVH_DEFAULT 4
#endif
struct Workspace;
struct msk_scene {
    msk_ctx *ctx = nullptr;
    // a scene of a group context: one ordinary scene per member, the members' films, staging copies for members without
    // peer access, and the summed film (msk_multi.h); everything below `ws` is then unused but dev.width / dev.height
    std::vector<msk_scene *> parts;
    DevBuf part_films[MSK_MAX_GROUP], staged[MSK_MAX_GROUP], group_film;
    void *group_host_film = nullptr; size_t group_host_film_bytes = 0;      // pinned staging of a group's copy-back (film_to_host)
    Workspace *ws = nullptr;           // render buffers, kept between calls (hipMalloc/hipFree of GBs costs milliseconds)
    DeviceScene dev;
    DevBuf nodes, nodes4, nodes4q, nodes8, tris, tris3, tri_bounds, tri_verts, tri_frames, tri_normals, tri_uvs, mesh_info, bsdfs, emitters, emitter_d65, emitter_grid, spectra, cdf, cie;
    bool lds_scene = false, lds_tables = false, all_diffuse = true;
    bool has_regular = false;          // the scene holds tabulated spectra (ABI v7): the shading instantiations that evaluate them run
    int trace_mode = 0;                // 0 binary tree in LDS, 1 binary tree in HBM/L2, 2 4-wide tree in HBM/L2, 4 8-wide quantised tree in HBM/L2,
                                       // 5 4-wide tree with quantised boxes in HBM/L2 (64-byte nodes; the default for trees in HBM)
    size_t trace_lds_bytes = 0, shade_lds_bytes = 0;
    int bvh_depth = 0;
    uint32_t n_tris = 0;
    size_t tree_bytes = 0;             // node array the traversal walks
};

static int ctx_sync(msk_ctx *ctx, hipStream_t stream, const char *what, double limit_scale = 1.0);     // (below: every wait of a render is timed)
static int film_to_host(msk_ctx *ctx, hipStream_t stream, const void *d_film, float *h_film, size_t bytes, void **stage, size_t *stage_bytes);
#include "msk_multi.h"

#ifdef MSK_COUNT
// instrumented builds only: reads and clears the traversal counters of msk_kernels.h
extern "C" int msk_gpu_debug_counts(unsigned long long *out16) {
    unsigned long long z[16] = {};
    if (hipDeviceSynchronize() != hipSuccess || hipMemcpyFromSymbol(out16, HIP_SYMBOL(msk_counts), sizeof z) != hipSuccess) return -1;
    return hipMemcpyToSymbol(HIP_SYMBOL(msk_counts), z, sizeof z) == hipSuccess ? 0 : -1;
}
#endif
// ------------------------------------------------------------------------------------------
extern "C" const char *msk_gpu_last_error(const msk_ctx *ctx) {
    return ctx ? ctx->last_error.c_str() : g_last_error.c_str();
}

extern "C" int msk_gpu_init(const int *device_ids, int n, msk_ctx **out_ctx) {
    if (!out_ctx) return fail(nullptr, MSK_ERR_INVALID_ARG, "msk_gpu_init: out_ctx is NULL");
    *out_ctx = nullptr;
    if (n < 1 || !device_ids)
        return fail(nullptr, MSK_ERR_INVALID_ARG, "msk_gpu_init: at least one device per context (got n=%d)", n);
    if (n > 1) return group_init(device_ids, n, out_ctx);          // msk_multi.h: one member context per entry
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count == 0)
        return fail(nullptr, MSK_ERR_NO_DEVICE, "msk_gpu_init: no HIP device visible");
    if (device_ids[0] < 0 || device_ids[0] >= count)
        return fail(nullptr, MSK_ERR_INVALID_ARG, "msk_gpu_init: device %d out of range (0..%d)", device_ids[0], count - 1);
    msk_ctx *ctx = new msk_ctx();
    ctx->device = device_ids[0];
    hipError_t e = hipSetDevice(ctx->device);
    if (e == hipSuccess) e = hipGetDeviceProperties(&ctx->prop, ctx->device);
    // The library's streams and the process's other streams.  The HIP runtime maps streams onto a few hardware queues PER PRIORITY
    // LEVEL (GPU_MAX_HW_QUEUES, default 4); streams beyond that share a queue and their launches serialise.  In a process that also
    // runs torch and RCCL (bench.py's ranks: torch's stream, the communicator's streams) two of the four wavefront loops ended up
    // behind each other: the N > 1 step took 39.4 ms against the 32.5 of the same render in a plain process (round 6,
    // profiles/r06_hw_queues.txt) — a fifth of the weak-scaling budget before any GPU talks to another.  MSK_STREAM_PRIORITY = high
    // (default) creates the library's streams at the device's highest stream priority: their own pool of hardware queues, which
    // nothing else in the process uses; normal / low: priority 0 / the lowest (then raise GPU_MAX_HW_QUEUES in the environment).
    int prio = 0;
    {
        int least = 0, greatest = 0;
        const char *ps = getenv("MSK_STREAM_PRIORITY");
        if (e == hipSuccess && hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess)
            prio = (!ps || !strcmp(ps, "high")) ? greatest : !strcmp(ps, "low") ? least : 0;
        (void) hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamCreateWithPriority(&ctx->stream, hipStreamNonBlocking, prio);
    for (int k = 0; k < MSK_MAX_STREAMS - 1; ++k) if (e == hipSuccess) e = hipStreamCreateWithPriority(&ctx->more_streams[k], hipStreamNonBlocking, prio);
    if (e == hipSuccess) e = hipHostMalloc((void **) &ctx->h_ctrl, MSK_MAX_STREAMS * sizeof(Ctrl), hipHostMallocDefault);
    ctx->hub = new mskwd::Hub();
    for (int k = 0; k < MSK_MAX_STREAMS; ++k) if (e == hipSuccess) e = hipEventCreateWithFlags(&ctx->wait_events[k], hipEventBlockingSync | hipEventDisableTiming);
    if (e != hipSuccess) {
        int rc = fail(nullptr, MSK_ERR_HIP, "msk_gpu_init: %s", hipGetErrorString(e));
        msk_gpu_shutdown(ctx);             // releases whichever of stream / pinned block exist
        return rc;
    }
    if (std::string(ctx->prop.gcnArchName).find("gfx950") == std::string::npos &&
        !getenv("MSK_ALLOW_ANY_ARCH")) {
        int rc = fail(nullptr, MSK_ERR_UNSUPPORTED, "msk_gpu_init: device is %s, this library is built for gfx950 only",
                      ctx->prop.gcnArchName);
        msk_gpu_shutdown(ctx);
        return rc;
    }
    *out_ctx = ctx;
    return MSK_OK;
}

extern "C" void msk_gpu_shutdown(msk_ctx *ctx) {
    if (!ctx) return;
    if (ctx->group) { group_shutdown(ctx); return; }
    if (ctx->lost) { delete ctx; return; }          // a stream that holds a kernel which never finished: destroying it would wait for it
                                                    // (ctx->hub stays: a host function queued behind that kernel may still fire)
    (void) hipSetDevice(ctx->device);
    for (auto ev : ctx->events) (void) hipEventDestroy(ev);
    for (auto ev : ctx->wait_events) if (ev) (void) hipEventDestroy(ev);
    for (int k = 0; k < MSK_MAX_STREAMS - 1; ++k) {
        for (auto ev : ctx->more_events[k]) (void) hipEventDestroy(ev);
        if (ctx->more_streams[k]) (void) hipStreamDestroy(ctx->more_streams[k]);
    }
    if (ctx->h_ctrl) (void) hipHostFree(ctx->h_ctrl);
    if (ctx->stream) (void) hipStreamDestroy(ctx->stream);      // (waits for what the stream holds: every host function has fired)
    delete ctx->hub;
    delete ctx;
}

extern "C" int msk_gpu_describe(const msk_ctx *ctx, char *buf, uint64_t buf_size) {
    if (!ctx || !buf || !buf_size) return fail(nullptr, MSK_ERR_INVALID_ARG, "msk_gpu_describe: bad argument");
    if (ctx->group) {
        char one[512];
        msk_gpu_describe(ctx->group->ctxs[0], one, sizeof one);
        std::string ids;
        for (msk_ctx *c : ctx->group->ctxs) ids += (ids.empty() ? "" : ",") + std::to_string(c->device);
        snprintf(buf, (size_t) buf_size, "%zu devices [%s], sample-sharded, film summed on device %d; each: %s", ctx->group->ctxs.size(), ids.c_str(), ctx->device, one);
        return MSK_OK;
    }
    snprintf(buf, (size_t) buf_size, "%s (%s), %d CUs, %.1f GiB HBM, LDS/block %zu KiB; libmsk_gpu ABI %d, fp-contract off; host side: "
             "%u loop thread(s) per render, wait = %s",
             ctx->prop.name, ctx->prop.gcnArchName, ctx->prop.multiProcessorCount,
             ctx->prop.totalGlobalMem / 1073741824.0, ctx->prop.sharedMemPerBlock / 1024, MSK_ABI_VERSION,
             std::max(1u, env_u32("MSK_HOST_THREADS", 1)), mskwd::wait_mode_name(mskwd::wait_mode_from_env()));
    return MSK_OK;
}

// ------------------------------------------------------------------------------------------
// scene
// ------------------------------------------------------------------------------------------
static inline float h_dot(const float *a, const float *b) { return a[0] * b[0] + (a[1] * b[1] + a[2] * b[2]); }

extern "C" int msk_gpu_scene_create(msk_ctx *ctx, const msk_scene_desc *d, msk_scene **out) {
    if (!ctx || !d || !out) return fail(ctx, MSK_ERR_INVALID_ARG, "msk_gpu_scene_create: NULL argument");
    *out = nullptr;
    MSK_REFUSE_LOST(ctx);
    if (ctx->group) return group_scene_create(ctx, d, out);
    if (d->abi_version != MSK_ABI_VERSION)
        return fail(ctx, MSK_ERR_INVALID_ARG, "msk_gpu_scene_create: abi_version %u != %u", d->abi_version, MSK_ABI_VERSION);
    if (d->film.width <= 0 || d->film.height <= 0 || !(d->film.filter_radius > 0.f))
        return fail(ctx, MSK_ERR_INVALID_ARG, "msk_gpu_scene_create: invalid film %dx%d radius %g", d->film.width,
                    d->film.height, d->film.filter_radius);
    {   // Film::set_crop_window (film.cpp:51-63); {0, 0} = the whole film
        const int32_t cx = d->film.crop_offset[0], cy = d->film.crop_offset[1], cw = d->film.crop_size[0], ch = d->film.crop_size[1];
        const bool whole = cw == 0 && ch == 0 && cx == 0 && cy == 0;
        if (!whole && (cx < 0 || cy < 0 || cw <= 0 || ch <= 0 || (int64_t) cx + cw > d->film.width || (int64_t) cy + ch > d->film.height))
            return fail(ctx, MSK_ERR_INVALID_ARG, "Invalid crop window specification! offset (%d, %d) + crop size (%d, %d) vs full size (%d, %d)",
                        cx, cy, cw, ch, d->film.width, d->film.height);
    }
    if (!d->cie1931_xyz || !d->d65) return fail(ctx, MSK_ERR_INVALID_ARG, "msk_gpu_scene_create: spectral tables missing");
    if ((d->n_faces && (!d->vertices || !d->faces)) || (d->n_meshes && !d->meshes))
        return fail(ctx, MSK_ERR_INVALID_ARG, "msk_gpu_scene_create: geometry arrays missing");
    if ((d->n_bsdfs && !d->bsdfs) || (d->n_emitters && !d->emitters))
        return fail(ctx, MSK_ERR_INVALID_ARG, "msk_gpu_scene_create: bsdf / emitter arrays missing");
    // leaf references pack first_tri << 5 | count into 31 bits (msk_bvh.h) and bit 31 of a hit's prim word is the shadow flag
    if (d->n_faces >= (1u << 26))
        return fail(ctx, MSK_ERR_UNSUPPORTED, "msk_gpu_scene_create: %u triangles, this back end addresses fewer than 2^26", d->n_faces);
    HIP_TRY(ctx, hipSetDevice(ctx->device));

    // ---- validate + gather per-triangle data (operand shapes are checked here, on the host,
    //      before any kernel can index with them)
    std::vector<float> pos((size_t) d->n_faces * 9);
    std::vector<float> tv((size_t) d->n_faces * 12), tn, tuv;
    std::vector<int32_t> mesh_info((size_t) std::max(1u, d->n_meshes) * 4, 0);
    bool any_normals = false, any_uvs = false;
    for (uint32_t m = 0; m < d->n_meshes; ++m) {
        const msk_mesh_desc &md = d->meshes[m];
        if ((uint64_t) md.first_vertex + md.vertex_count > d->n_vertices || (uint64_t) md.first_face + md.face_count > d->n_faces)
            return fail(ctx, MSK_ERR_INVALID_ARG, "mesh %u: vertex/face range exceeds the arrays", m);
        if (md.bsdf_id < 0 || (uint32_t) md.bsdf_id >= d->n_bsdfs)
            return fail(ctx, MSK_ERR_INVALID_ARG, "mesh %u: bsdf_id %d out of range", m, md.bsdf_id);
        if (md.emitter_id >= (int32_t) d->n_emitters || md.emitter_id < -1)
            return fail(ctx, MSK_ERR_INVALID_ARG, "mesh %u: emitter_id %d out of range", m, md.emitter_id);
        // one mesh per area emitter, named from both sides (Shape::m_emitter / Emitter::m_shape, shape.cpp:27-37): a second
        // mesh pointing at the same emitter would be lit with the first one's area pdf and CDF
        if (md.emitter_id >= 0 && (!d->emitters || d->emitters[md.emitter_id].mesh_id != (int32_t) m))
            return fail(ctx, MSK_ERR_INVALID_ARG, "mesh %u: emitter %d does not point back at it", m, md.emitter_id);
        any_normals |= md.has_normals != 0; any_uvs |= md.has_texcoords != 0;
    }
    if (any_normals) tn.assign((size_t) d->n_faces * 12, 0.f);
    if (any_uvs) tuv.assign((size_t) d->n_faces * 8, 0.f);
    int rc_spec = MSK_OK;
    std::vector<uint8_t> covered(d->n_faces, 0);
    std::vector<float> mesh_area(d->n_meshes, 0.f);
    std::vector<std::vector<float>> mesh_cdf(d->n_meshes);
    for (uint32_t m = 0; m < d->n_meshes; ++m) {
        const msk_mesh_desc &md = d->meshes[m];
        mesh_info[m * 4 + 0] = md.bsdf_id; mesh_info[m * 4 + 1] = md.emitter_id;
        mesh_info[m * 4 + 2] = (md.has_normals ? 1 : 0) | (md.has_texcoords ? 2 : 0);
        mesh_info[m * 4 + 3] = (int32_t) md.first_face;
        // mesh.cpp:39-48 area_distr_build + core/distribution.h:88-96 (fp32, started from 0)
        float surface_area = 0.f, run = 0.f;
        std::vector<float> &cdf = mesh_cdf[m];
        cdf.push_back(0.f);
        for (uint32_t f = 0; f < md.face_count; ++f) {
            const uint32_t g = md.first_face + f;
            if (covered[g]) return fail(ctx, MSK_ERR_INVALID_ARG, "face %u belongs to two meshes", g);
            covered[g] = 1;
            const uint32_t *fi = d->faces + (size_t) g * 3;
            const float *v[3];
            for (int k = 0; k < 3; ++k) {
                if (fi[k] >= md.vertex_count) return fail(ctx, MSK_ERR_INVALID_ARG, "mesh %u face %u: vertex index %u out of range", m, f, fi[k]);
                v[k] = d->vertices + (size_t) (md.first_vertex + fi[k]) * 8;
                if (!(std::isfinite(v[k][0]) && std::isfinite(v[k][1]) && std::isfinite(v[k][2])))
                    return fail(ctx, MSK_ERR_INVALID_ARG, "mesh %u vertex %u: non-finite position", m, fi[k]);
                for (int c = 0; c < 3; ++c) { pos[(size_t) g * 9 + k * 3 + c] = v[k][c]; tv[(size_t) g * 12 + k * 4 + c] = v[k][c]; }
                if (any_normals) for (int c = 0; c < 3; ++c) tn[(size_t) g * 12 + k * 4 + c] = v[k][3 + c];
            }
            uint32_t mb = m; std::memcpy(&tv[(size_t) g * 12 + 3], &mb, 4);
            if (any_uvs) {
                float *u = &tuv[(size_t) g * 8];
                u[0] = v[0][6]; u[1] = v[0][7]; u[2] = v[1][6]; u[3] = v[1][7]; u[4] = v[2][6]; u[5] = v[2][7];
            }
            // mesh.h:54-61 face_area = 0.5 * |(p1-p0) x (p2-p0)|
            const float a[3] = {v[1][0] - v[0][0], v[1][1] - v[0][1], v[1][2] - v[0][2]};
            const float b[3] = {v[2][0] - v[0][0], v[2][1] - v[0][1], v[2][2] - v[0][2]};
            const float cr[3] = {a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]};
            const float area = 0.5f * std::sqrt(h_dot(cr, cr));
            surface_area += area;
            run = (f == 0) ? area : run + area;
            cdf.push_back(run);
        }
        if (md.face_count) { const float inv_sum = 1.f / cdf.back(); for (auto &c : cdf) c *= inv_sum; }
        mesh_area[m] = surface_area;
    }
    for (uint32_t g = 0; g < d->n_faces; ++g)
        if (!covered[g]) return fail(ctx, MSK_ERR_INVALID_ARG, "face %u belongs to no mesh", g);

    // ---- tabulated (`regular`) spectra, ABI v7 (spectra/regular.cpp:27-70): validated here, before any kernel indexes with them
    if ((d->n_regular_spectra && !d->regular_spectra) || (d->n_regular_values && !d->regular_values))
        return fail(ctx, MSK_ERR_INVALID_ARG, "msk_gpu_scene_create: regular spectrum arrays missing");
    if (d->n_regular_values >= (1u << 24))
        return fail(ctx, MSK_ERR_UNSUPPORTED, "msk_gpu_scene_create: %u tabulated spectrum values, a spectrum record addresses fewer than 2^24", d->n_regular_values);
    for (uint32_t k = 0; k < d->n_regular_spectra; ++k) {
        const msk_regular_spectrum_desc &r = d->regular_spectra[k];
        if (r.size < 2) return fail(ctx, MSK_ERR_INVALID_ARG, "regular spectrum %u: ContinuousDistribution: needs at least two entries!", k);     // regular.cpp:30-31
        if (r.size > MSK_REGULAR_MAX) return fail(ctx, MSK_ERR_UNSUPPORTED, "regular spectrum %u: %u values, this back end takes at most %d", k, r.size, MSK_REGULAR_MAX);
        if (!(r.lambda_min < r.lambda_max) || !std::isfinite(r.lambda_min) || !std::isfinite(r.lambda_max))
            return fail(ctx, MSK_ERR_INVALID_ARG, "regular spectrum %u: ContinuousDistribution: invalid range!", k);                                // regular.cpp:33-34
        if ((uint64_t) r.first_value + r.size > d->n_regular_values)
            return fail(ctx, MSK_ERR_INVALID_ARG, "regular spectrum %u: values [%u, %u) exceed the %u in regular_values", k, r.first_value, r.first_value + r.size, d->n_regular_values);
        bool mass = false;
        for (uint32_t i = 0; i < r.size; ++i) {
            const float v = d->regular_values[r.first_value + i];
            if (!(v >= 0.f) || !std::isfinite(v)) return fail(ctx, MSK_ERR_INVALID_ARG, "regular spectrum %u: ContinuousDistribution: entries must be non-negative!", k);   // regular.cpp:59-60
            mass |= v > 0.f;
        }
        if (!mass) return fail(ctx, MSK_ERR_INVALID_ARG, "regular spectrum %u: ContinuousDistribution: no probability mass found!", k);            // regular.cpp:73-74
    }
    // 1 / interval as RegularSpectrum keeps it: the interval in double, its reciprocal rounded to float (regular.cpp:42-43,66-70)
    auto inv_interval = [&](const msk_regular_spectrum_desc &r) { return (float) (1.0 / (((double) r.lambda_max - (double) r.lambda_min) / (double) (r.size - 1))); };
    bool any_regular = false;
    // device form of an msk_spectrum_desc: {c0, c1, c2, scale} or {lambda_min, inv_interval, first | last << 24, -1} (msk_kernels.h: spectrum_eval)
    auto spectrum_record = [&](const msk_spectrum_desc &sp, float *o, const char *what, uint32_t b) -> int {
        if (sp.regular) {
            if (sp.regular > d->n_regular_spectra) return fail(ctx, MSK_ERR_INVALID_ARG, "bsdf %u: %s names regular spectrum %u of %u", b, what, sp.regular, d->n_regular_spectra);
            const msk_regular_spectrum_desc &r = d->regular_spectra[sp.regular - 1];
            const uint32_t w = r.first_value | ((r.size - 2u) << 24);
            o[0] = r.lambda_min; o[1] = inv_interval(r); std::memcpy(&o[2], &w, 4); o[3] = -1.f;
            any_regular = true;
            return MSK_OK;
        }
        if (!(sp.scale >= 0.f)) return fail(ctx, MSK_ERR_INVALID_ARG, "bsdf %u: the scale of %s must not be negative", b, what);
        o[0] = sp.coeff[0]; o[1] = sp.coeff[1]; o[2] = sp.coeff[2]; o[3] = sp.scale;
        return MSK_OK;
    };
    if (d->n_textures && !d->textures) return fail(ctx, MSK_ERR_INVALID_ARG, "msk_gpu_scene_create: texture array missing");
    const uint32_t n_bsdf_f4 = std::max(1u, d->n_bsdfs) * MSK_BSDF_F4 + d->n_textures * 3;
    std::vector<float> bsdfs((size_t) n_bsdf_f4 * 4, 0.f);
    for (uint32_t t = 0; t < d->n_textures; ++t) {
        const msk_texture_desc &td = d->textures[t];
        if (td.type != MSK_TEXTURE_CHECKERBOARD)
            return fail(ctx, MSK_ERR_UNSUPPORTED, "texture %u: type %d is not supported by this back end (checkerboard)", t, td.type);
        float *o = &bsdfs[((size_t) std::max(1u, d->n_bsdfs) * MSK_BSDF_F4 + (size_t) t * 3) * 4];
        o[0] = td.color0[0]; o[1] = td.color0[1]; o[2] = td.color0[2]; o[3] = td.to_uv[2];
        o[4] = td.color1[0]; o[5] = td.color1[1]; o[6] = td.color1[2]; o[7] = td.to_uv[5];
        o[8] = td.to_uv[0]; o[9] = td.to_uv[1]; o[10] = td.to_uv[3]; o[11] = td.to_uv[4];
    }
    bool all_diffuse = true;
    for (uint32_t b = 0; b < d->n_bsdfs; ++b) {
        const msk_bsdf_desc &bd = d->bsdfs[b];
        if (bd.type != MSK_BSDF_DIFFUSE && bd.type != MSK_BSDF_ROUGHCONDUCTOR && bd.type != MSK_BSDF_ROUGHDIELECTRIC)
            return fail(ctx, MSK_ERR_UNSUPPORTED,
                        "bsdf %u: type %d is not supported by this back end (diffuse, roughconductor, roughdielectric)", b, bd.type);
        if (bd.back_bsdf >= (int32_t) d->n_bsdfs || bd.back_bsdf < -1)
            return fail(ctx, MSK_ERR_INVALID_ARG, "bsdf %u: back_bsdf %d out of range", b, bd.back_bsdf);
        if (bd.type != MSK_BSDF_DIFFUSE && !(bd.alpha_u >= 0.f && bd.alpha_v >= 0.f))
            return fail(ctx, MSK_ERR_INVALID_ARG, "bsdf %u: negative roughness", b);
        if (bd.type == MSK_BSDF_ROUGHDIELECTRIC && !(bd.ior_eta > 0.f && bd.ior_inv_eta > 0.f))
            return fail(ctx, MSK_ERR_INVALID_ARG, "bsdf %u: the relative index of refraction must be positive", b);
        if (bd.reflectance_texture > d->n_textures)
            return fail(ctx, MSK_ERR_INVALID_ARG, "bsdf %u: reflectance_texture %u out of range", b, bd.reflectance_texture);
        if (bd.type == MSK_BSDF_DIFFUSE && !bd.reflectance_regular && !(bd.reflectance_scale >= 0.f && bd.reflectance_scale < INFINITY))
            return fail(ctx, MSK_ERR_INVALID_ARG, "bsdf %u: reflectance_scale must be finite and non-negative (1 for an srgb reflectance)", b);
        if (bd.reflectance_texture && bd.type != MSK_BSDF_DIFFUSE)
            return fail(ctx, MSK_ERR_UNSUPPORTED, "bsdf %u: only the diffuse reflectance can be textured", b);
        if (bd.reflectance_regular && (bd.type != MSK_BSDF_DIFFUSE || bd.reflectance_texture))
            return fail(ctx, MSK_ERR_INVALID_ARG, "bsdf %u: reflectance_regular names the reflectance of an untextured diffuse BSDF", b);
        if (bd.type != MSK_BSDF_DIFFUSE || bd.back_bsdf >= 0 || bd.reflectance_texture || bd.reflectance_regular) all_diffuse = false;
        // the device record, MSK_BSDF_F4 float4 (msk_kernels.h: BsdfRec): {type, back, r0, r1} {r2, au, av, sample_visible} eta k
        // spec trans {ior_eta, ior_inv_eta, float4 offset of the texture record, reflectance_scale}
        float *o = &bsdfs[(size_t) b * 4 * MSK_BSDF_F4];
        std::memcpy(&o[0], &bd.type, 4); std::memcpy(&o[1], &bd.back_bsdf, 4);
        msk_spectrum_desc refl;
        refl.coeff[0] = bd.reflectance[0]; refl.coeff[1] = bd.reflectance[1]; refl.coeff[2] = bd.reflectance[2];
        refl.scale = bd.type == MSK_BSDF_DIFFUSE ? bd.reflectance_scale : 0.f; refl.regular = bd.reflectance_regular;
        float r4[4];
        if ((rc_spec = spectrum_record(refl, r4, "reflectance", b))) return rc_spec;
        o[2] = r4[0]; o[3] = r4[1]; o[4] = r4[2];
        o[5] = bd.alpha_u; o[6] = bd.alpha_v; std::memcpy(&o[7], &bd.sample_visible, 4);
        if ((rc_spec = spectrum_record(bd.eta, &o[8], "eta", b)) || (rc_spec = spectrum_record(bd.k, &o[12], "k", b)) ||
            (rc_spec = spectrum_record(bd.specular_reflectance, &o[16], "specular_reflectance", b)) ||
            (rc_spec = spectrum_record(bd.specular_transmittance, &o[20], "specular_transmittance", b))) return rc_spec;
        o[24] = bd.ior_eta; o[25] = bd.ior_inv_eta;
        const uint32_t tex_off = bd.reflectance_texture ? std::max(1u, d->n_bsdfs) * MSK_BSDF_F4 + (bd.reflectance_texture - 1) * 3 : 0u;
        std::memcpy(&o[26], &tex_off, 4);
        o[27] = r4[3];                                             // reflectance_scale, or -1 for a tabulated reflectance
    }
    std::vector<float> emitters((size_t) std::max(1u, d->n_emitters) * 8, 0.f), d65((size_t) std::max(1u, d->n_emitters) * 95, 0.f), cdf_all;
    std::vector<float> emitter_grid((size_t) std::max(1u, d->n_emitters) * 4, 0.f);
    int env_emitter = -1;
    for (uint32_t e = 0; e < d->n_emitters; ++e) {
        const msk_emitter_desc &ed = d->emitters[e];
        float *o = &emitters[e * 8];
        o[0] = ed.radiance[0]; o[1] = ed.radiance[1]; o[2] = ed.radiance[2];
        if (ed.type == MSK_EMITTER_CONSTANT) {
            if (env_emitter >= 0) return fail(ctx, MSK_ERR_INVALID_ARG, "Can only have one environment light");   // scene.cpp:38-39
            if (ed.mesh_id != -1) return fail(ctx, MSK_ERR_INVALID_ARG, "emitter %u: an environment emitter has no mesh (mesh_id must be -1)", e);
            env_emitter = (int) e;
            o[3] = 0.f;
            const uint32_t meta[4] = {0xffffffffu, 0u, 0u, 0u};
            std::memcpy(&o[4], meta, 16);
        } else if (ed.type == MSK_EMITTER_AREA) {
            if (ed.mesh_id < 0 || (uint32_t) ed.mesh_id >= d->n_meshes || d->meshes[ed.mesh_id].emitter_id != (int32_t) e)
                return fail(ctx, MSK_ERR_INVALID_ARG, "emitter %u: mesh_id %d does not point back at it", e, ed.mesh_id);
            const msk_mesh_desc &md = d->meshes[ed.mesh_id];
            if (md.face_count == 0) return fail(ctx, MSK_ERR_INVALID_ARG, "emitter %u: its mesh has no faces", e);
            o[3] = 1.f / mesh_area[ed.mesh_id];                               // mesh.cpp:129 ps.pdf
            uint32_t meta[4] = {(uint32_t) ed.mesh_id, md.first_face, md.face_count, (uint32_t) cdf_all.size()};
            std::memcpy(&o[4], meta, 16);
            cdf_all.insert(cdf_all.end(), mesh_cdf[ed.mesh_id].begin(), mesh_cdf[ed.mesh_id].end());
        } else {
            return fail(ctx, MSK_ERR_UNSUPPORTED, "emitter %u: type %d is not supported (area, constant)", e, ed.type);
        }
        float *gr = &emitter_grid[(size_t) e * 4];
        if (ed.radiance_regular) {          // a `regular` radiance: its own table on its own grid, no sigmoid factor (area.cpp:51-54, regular.cpp:148)
            if (ed.radiance_regular > d->n_regular_spectra) return fail(ctx, MSK_ERR_INVALID_ARG, "emitter %u: radiance names regular spectrum %u of %u", e, ed.radiance_regular, d->n_regular_spectra);
            const msk_regular_spectrum_desc &r = d->regular_spectra[ed.radiance_regular - 1];
            for (uint32_t i = 0; i < r.size; ++i) d65[e * 95 + i] = d->regular_values[r.first_value + i];
            const uint32_t last = r.size - 2u;
            gr[0] = r.lambda_min; gr[1] = inv_interval(r); std::memcpy(&gr[2], &last, 4); gr[3] = 1.f;
            o[0] = o[1] = 0.f; o[2] = INFINITY;
            any_regular = true;
        } else {
            for (int i = 0; i < 95; ++i) d65[e * 95 + i] = d->d65[i] * ed.d65_scale;   // d65.cpp:41-42
            const uint32_t last = 93u;
            gr[0] = 360.f; gr[1] = (float) (1.0 / ((830.0 - 360.0) / 94.0)); std::memcpy(&gr[2], &last, 4); gr[3] = 0.f;
        }
    }
    if (any_regular) all_diffuse = false;              // tabulated spectra are evaluated by the general shading variant only
    std::vector<float> spectra_pool(d->regular_values, d->regular_values + d->n_regular_values);
    if (spectra_pool.empty()) spectra_pool.push_back(0.f);
    if (cdf_all.empty()) cdf_all.push_back(0.f);

    // oracle D10: the triangle-bounds predicate's padding, computed exactly as the oracle does (0.5e-5 of the scene's scale =
    // max(diagonal, largest coordinate magnitude)); node boxes are padded by twice that
    float tri_pad, scene_lo[3] = {INFINITY, INFINITY, INFINITY}, scene_hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    {
        float *lo = scene_lo, *hi = scene_hi;
        for (uint32_t t = 0; t < d->n_faces; ++t)
            for (int v = 0; v < 3; ++v)
                for (int k = 0; k < 3; ++k) { const float q = pos[(size_t) t * 9 + v * 3 + k]; lo[k] = std::min(lo[k], q); hi[k] = std::max(hi[k], q); }
        const float ex = hi[0] - lo[0], ey = hi[1] - lo[1], ez = hi[2] - lo[2];
        const float diag = d->n_faces ? std::sqrt(ex * ex + (ey * ey + ez * ez)) : 1.f;
        float amax = 0.f;
        if (d->n_faces) amax = std::max(std::max(std::max(std::fabs(lo[0]), std::fabs(hi[0])), std::max(std::fabs(lo[1]), std::fabs(hi[1]))), std::max(std::fabs(lo[2]), std::fabs(hi[2])));
        // (MSK_PAD_SCALE, read by the oracle too: the margin tests shrink the padding on both sides to show how far the rule is from
        // failing — tree and brute force agree down to 1e-7 of the scale and part at 1e-8, tests/test_padding_margin.py.  A value that
        // is not a number in [1e-7, 1e-3] is refused: zero, garbage or a stray setting would silently remove the padding on both
        // sides of the parity tests at once.)
        float pad_scale = 1e-5f;
        if (const char *ps = getenv("MSK_PAD_SCALE")) {
            char *end = nullptr;
            const double v = strtod(ps, &end);
            if (end == ps || *end != '\0' || !std::isfinite(v) || v < 1e-7 || v > 1e-3)
                return fail(ctx, MSK_ERR_INVALID_ARG, "MSK_PAD_SCALE=\"%s\": the padding of the boxes must be a number in [1e-7, 1e-3] of the scene's scale "
                            "(default 1e-5; the slab arithmetic needs a few 1e-7)", ps);
            pad_scale = (float) v;
        }
        tri_pad = (0.5f * pad_scale) * std::max(diag, amax);
    }
    // MSK_BVH_BUILD=gpu: the tree is built on the device (msk_lbvh.hip, a linear BVH) once the vertices are uploaded — the
    // option for scenes that change between renders; default: the host's binned-SAH builder (msk_bvh.h), the better tree
    const char *build_env = getenv("MSK_BVH_BUILD");
    const bool gpu_build = build_env && !strcmp(build_env, "gpu") && d->n_faces > 0;
    mskbvh::Built bvh;
    if (!gpu_build) bvh = mskbvh::build(pos.data(), d->n_faces, tri_pad);
    // material class of every triangle into its leaf record's prim word (MSK_CLASS_SHIFT): the traversal hands it to the
    // shading kernel with the hit, which sorts its paths by it
    if (!all_diffuse && !gpu_build)
        for (size_t k = 0; k < d->n_faces; ++k) {
            uint32_t w, mesh;
            std::memcpy(&w, &bvh.tris[k * 16 + 3], 4);
            std::memcpy(&mesh, &tv[(size_t) w * 12 + 3], 4);
            const int32_t b = mesh_info[(size_t) mesh * 4];
            const uint32_t cls = (b >= 0 && (uint32_t) b < d->n_bsdfs) ? (uint32_t) d->bsdfs[b].type : 0u;      // MSK_BSDF_* = 0, 1, 2
            w |= (cls & (MSK_N_CLASSES - 1u)) << MSK_CLASS_SHIFT;
            std::memcpy(&bvh.tris[k * 16 + 3], &w, 4);
        }

    msk_scene *s = new msk_scene();
    s->ctx = ctx; s->n_tris = d->n_faces; s->bvh_depth = bvh.max_depth; s->all_diffuse = all_diffuse; s->has_regular = any_regular;
    std::vector<float> cie(d->cie1931_xyz, d->cie1931_xyz + 3 * MSK_CIE_SAMPLES);
    hipError_t e = hipSuccess;
    auto up = [&](DevBuf &b, const std::vector<float> &v) { if (e == hipSuccess) e = b.upload(v); };
    if (!gpu_build) { up(s->nodes, bvh.nodes); up(s->tris, bvh.tris); up(s->tri_bounds, bvh.bounds); }
    up(s->tri_verts, tv); up(s->tri_normals, tn); up(s->tri_uvs, tuv);
    up(s->bsdfs, bsdfs); up(s->emitters, emitters); up(s->emitter_d65, d65); up(s->cdf, cdf_all); up(s->cie, cie);
    up(s->emitter_grid, emitter_grid); up(s->spectra, spectra_pool);
    if (e == hipSuccess) e = s->mesh_info.upload(mesh_info);
    if (e == hipSuccess && gpu_build) {
        const size_t nf = d->n_faces;
        if ((e = s->nodes.alloc(nf * 64)) == hipSuccess && (e = s->tris.alloc(nf * 64)) == hipSuccess) e = s->tri_bounds.alloc(nf * 32);
    }
    if (e != hipSuccess) { delete s; return fail(ctx, e == hipErrorOutOfMemory ? MSK_ERR_OOM : MSK_ERR_HIP, "scene upload: %s", hipGetErrorString(e)); }
    if (gpu_build) {
        msklbvh::Input in;
        in.tri_verts = s->tri_verts.as<float4>(); in.n_tris = d->n_faces;
        for (int k = 0; k < 3; ++k) { in.lo[k] = scene_lo[k]; in.hi[k] = scene_hi[k]; }
        in.box_pad = 2.f * tri_pad; in.tri_pad = tri_pad;
        in.leaf_size = getenv("MSK_BVH_LEAF") ? (uint32_t) std::max(1, std::min(8, atoi(getenv("MSK_BVH_LEAF")))) : 2u;
        in.mesh_info = s->mesh_info.as<int4>(); in.bsdfs = s->bsdfs.as<float4>(); in.n_bsdfs = d->n_bsdfs; in.bsdf_f4 = MSK_BSDF_F4;
        in.class_shift = all_diffuse ? 0u : (uint32_t) MSK_CLASS_SHIFT;
        msklbvh::Result res;
        char msg[256] = "";
        if (msklbvh::build(ctx->stream, in, s->nodes.as<float4>(), s->tris.as<float4>(), s->tri_bounds.as<float4>(), &res, msg, sizeof msg)) {
            delete s;
            return fail(ctx, MSK_ERR_HIP, "device BVH build: %s", msg);
        }
        // the node records back on the host: the size / depth bookkeeping below and the wide collapse read them
        bvh.root_ref = res.root_ref; bvh.max_depth = res.depth; s->bvh_depth = res.depth;      // (k_path_serial sizes its stack by it)
        bvh.nodes.resize((size_t) res.n_nodes * 16);
        if (res.n_nodes) {
            hipError_t ec = hipMemcpy(bvh.nodes.data(), s->nodes.p, (size_t) res.n_nodes * 64, hipMemcpyDeviceToHost);
            if (ec != hipSuccess) { delete s; return fail(ctx, MSK_ERR_HIP, "device BVH build: %s", hipGetErrorString(ec)); }
        }
    }

    DeviceScene &ds = s->dev;
    std::memset(&ds, 0, sizeof ds);
    ds.nodes = s->nodes.as<float4>(); ds.tris = s->tris.as<float4>(); ds.tri_bounds = s->tri_bounds.as<float4>(); ds.tri_pad = tri_pad; ds.tri_verts = s->tri_verts.as<float4>();
    ds.tri_normals = any_normals ? s->tri_normals.as<float4>() : nullptr;
    ds.tri_uvs = any_uvs ? s->tri_uvs.as<float4>() : nullptr;
    ds.mesh_info = s->mesh_info.as<int4>(); ds.bsdfs = s->bsdfs.as<float4>(); ds.emitters = s->emitters.as<float4>();
    ds.emitter_d65 = s->emitter_d65.as<float>(); ds.cdf = s->cdf.as<float>(); ds.cie = s->cie.as<float>();
    ds.emitter_grid = s->emitter_grid.as<float4>(); ds.spectra = s->spectra.as<float>(); ds.n_spectra = d->n_regular_values;
    ds.n_nodes = (uint32_t) (bvh.nodes.size() / 16); ds.n_tris = d->n_faces; ds.n_emitters = d->n_emitters;
    ds.n_meshes = d->n_meshes; ds.n_bsdfs = d->n_bsdfs; ds.n_bsdf_f4 = n_bsdf_f4; ds.cdf_len = (uint32_t) cdf_all.size();
    ds.root_ref = bvh.root_ref;
    ds.stack_entries = (uint32_t) ((bvh.max_depth + 2 + 3) & ~3);
    std::memcpy(ds.s2c, d->camera.sample_to_camera, 64); std::memcpy(ds.to_world, d->camera.to_world, 64);
    ds.near_clip = d->camera.near_clip; ds.far_clip = d->camera.far_clip;
    ds.width = d->film.width; ds.height = d->film.height;
    ds.crop_x = d->film.crop_offset[0]; ds.crop_y = d->film.crop_offset[1];
    ds.crop_w = d->film.crop_size[0]; ds.crop_h = d->film.crop_size[1];
    if (ds.crop_w == 0 && ds.crop_h == 0) { ds.crop_x = ds.crop_y = 0; ds.crop_w = ds.width; ds.crop_h = ds.height; }      // the whole film
    ds.filter_radius = d->film.filter_radius;
    ds.filter_scale = float(MSK_FILTER_RESOLUTION) / d->film.filter_radius;           // rfilter.cpp:21
    ds.filter_border = (int) std::ceil(d->film.filter_radius - .5f);                  // rfilter.cpp:22
    std::memcpy(ds.lut, d->film.filter_lut, sizeof ds.lut);
    // constant.cpp:21-28 set_scene: the sphere around Scene::bbox() (bbox.h:105-112), in fp32 exactly as the oracle does
    ds.env_emitter = env_emitter; ds.env_radius = 0.f;
    if (env_emitter >= 0) {
        float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
        for (uint32_t v = 0; v < d->n_vertices; ++v)
            for (int k = 0; k < 3; ++k) { const float q = d->vertices[(size_t) v * 8 + k]; lo[k] = std::min(lo[k], q); hi[k] = std::max(hi[k], q); }
        float r[3];
        for (int k = 0; k < 3; ++k) { const float c = (lo[k] + hi[k]) * .5f; r[k] = c - hi[k]; }
        const float radius = std::sqrt(r[0] * r[0] + (r[1] * r[1] + r[2] * r[2]));
        ds.env_radius = std::max(MSK_RAY_EPS_F, radius * (1.f + MSK_RAY_EPS_F));
        s->all_diffuse = false;             // the environment terms live in the general shading variant
    }
    // LDS plan of k_trace: per-lane stack + (when it fits) the whole BVH
    const size_t stack_bytes = (size_t) ds.stack_entries * MSK_BLOCK * 4;
    const size_t scene_bytes = (size_t) ds.n_nodes * 64 + (size_t) ds.n_tris * 96;
    const size_t lds_cap = getenv("MSK_LDS_SCENE_KB") ? (size_t) atoi(getenv("MSK_LDS_SCENE_KB")) * 1024 : 48 * 1024;
    s->lds_scene = scene_bytes <= lds_cap && stack_bytes + scene_bytes <= 64 * 1024;
    s->trace_lds_bytes = stack_bytes + (s->lds_scene ? scene_bytes : 0);
    s->trace_mode = s->lds_scene ? 0 : 1;
    s->tree_bytes = bvh.nodes.size() * 4;
    // (a tree whose boxes cannot be quantised conservatively — extents beyond the exponent range — keeps the 4-wide form)
    if (!s->lds_scene && env_u32("MSK_WIDE_BVH", MSK_WIDE_BVH_DEFAULT) == 8 && !(bvh.root_ref & MSK_LEAF_BIT) && mskbvh::collapse8(bvh)) {
        // the tree stays in HBM/L2: eight quantised child boxes per 128-byte line (msk_bvh.h: collapse8)
        hipError_t e8 = s->nodes8.upload(bvh.nodes8);
        if (e8 != hipSuccess) { delete s; return fail(ctx, MSK_ERR_OOM, "scene upload: %s", hipGetErrorString(e8)); }
        ds.nodes8 = s->nodes8.as<float4>(); ds.root_ref8 = bvh.root_ref8; ds.n_nodes8 = (uint32_t) (bvh.nodes8.size() / 32);
        ds.stack_entries = (uint32_t) ((7 * bvh.max_depth8 + 2 + 3) & ~3);     // up to seven pushes per level
        s->trace_mode = 4;
        s->tree_bytes = bvh.nodes8.size() * 4;
    } else if (!s->lds_scene && env_u32("MSK_WIDE_BVH", MSK_WIDE_BVH_DEFAULT) && !(bvh.root_ref & MSK_LEAF_BIT)) {
        // the tree stays in HBM/L2: walk it four children at a time, one 128-byte line per visit
        mskbvh::collapse4(bvh, env_u32("MSK_COLLAPSE_OPTIMAL", 1) != 0);
        // the traversal addresses nodes and triangle records through buffer resources with 32-bit byte offsets
        if ((uint64_t) (bvh.nodes4.size() / 32) * 128u >= (1ull << 32) || (uint64_t) d->n_faces * MSK_TRI_REC_BYTES >= (1ull << 32)) {
            delete s;
            return fail(ctx, MSK_ERR_UNSUPPORTED, "a tree of %zu 4-wide nodes exceeds the 4 GB a buffer resource addresses", bvh.nodes4.size() / 32);
        }
        hipError_t e4 = s->nodes4.upload(bvh.nodes4);
        if (e4 != hipSuccess) { delete s; return fail(ctx, MSK_ERR_OOM, "scene upload: %s", hipGetErrorString(e4)); }
        ds.nodes4 = s->nodes4.as<float4>(); ds.root_ref4 = bvh.root_ref4; ds.n_nodes4 = (uint32_t) (bvh.nodes4.size() / 32);
        ds.stack_entries = (uint32_t) ((3 * bvh.max_depth4 + 2 + 3) & ~3);     // up to three pushes per level
        s->trace_mode = 2;
        s->tree_bytes = bvh.nodes4.size() * 4;
        // the walked form (MSK_QUANT_BVH): 2 (default since round 5) 80-byte nodes with half-float child boxes read by v_fma_mix_f32,
        // 1 64-byte nodes with byte-quantised boxes (rounds 3-4), 0 the full-precision 128-byte ones; + three-load triangle
        // records, packed on the device from `tris`.  Config-5 / config-3 class renders (round 5, same box): 124.7 / 159.5 ms
        // with 2 against 126.9 / 168.3 with 1.
        if (env_u32("MSK_QUANT_BVH", 2) == 2 && !bvh.nodes4h.empty()) {
            hipError_t eq = s->nodes4q.upload(bvh.nodes4h);
            if (eq != hipSuccess) { delete s; return fail(ctx, MSK_ERR_OOM, "scene upload: %s", hipGetErrorString(eq)); }
            ds.nodes4q = s->nodes4q.as<float4>();
            if (!(ds.root_ref4 & MSK_LEAF_BIT)) ds.root_ref4 *= 80u;     // inner references of this form are byte offsets (msk_bvh.h: quantise_h)
            s->trace_mode = 6;
            s->tree_bytes = bvh.nodes4h.size() * 4;
        } else if (env_u32("MSK_QUANT_BVH", 2) && !bvh.nodes4q.empty()) {
            hipError_t eq = s->nodes4q.upload(bvh.nodes4q);
            if (eq != hipSuccess) { delete s; return fail(ctx, MSK_ERR_OOM, "scene upload: %s", hipGetErrorString(eq)); }
            ds.nodes4q = s->nodes4q.as<float4>();
            s->trace_mode = 5;
            s->tree_bytes = bvh.nodes4q.size() * 4;
        }
        {
            hipError_t et = s->tris3.alloc(std::max<size_t>((size_t) d->n_faces * MSK_TRI_REC_BYTES, 16));
            if (et != hipSuccess) { delete s; return fail(ctx, MSK_ERR_OOM, "scene upload: %s", hipGetErrorString(et)); }
            ds.tris3 = s->tris3.as<float4>();
            if (d->n_faces) hipLaunchKernelGGL(k_pack_tris3, dim3((d->n_faces + MSK_BLOCK - 1) / MSK_BLOCK), dim3(MSK_BLOCK), 0, ctx->stream,
                                               s->tris.as<float4>(), s->tri_bounds.as<float4>(), d->n_faces, s->tris3.as<float4>());
        }
    } else if (s->lds_scene && env_u32("MSK_WIDE_LDS", 0) && !(bvh.root_ref & MSK_LEAF_BIT)) {
        // experiment knob, off by default: the 4-wide tree staged in LDS (half the dependent LDS round trips per ray).
        // Measured on cbox: trace 12.45 vs 12.34 ms for the binary tree — no gain.
        mskbvh::collapse4(bvh, env_u32("MSK_COLLAPSE_OPTIMAL", 1) != 0);
        hipError_t e4 = s->nodes4.upload(bvh.nodes4);
        if (e4 != hipSuccess) { delete s; return fail(ctx, MSK_ERR_OOM, "scene upload: %s", hipGetErrorString(e4)); }
        ds.nodes4 = s->nodes4.as<float4>(); ds.root_ref4 = bvh.root_ref4; ds.n_nodes4 = (uint32_t) (bvh.nodes4.size() / 32);
        ds.stack_entries = (uint32_t) ((3 * bvh.max_depth4 + 2 + 3) & ~3);
        s->trace_lds_bytes = (size_t) ds.stack_entries * MSK_BLOCK * 4 + ((size_t) ds.n_nodes4 * 128 + (size_t) ds.n_tris * 96);
        s->trace_mode = 3;
    }
    ds.stack_total = ds.stack_entries;
    if (s->trace_mode == 1 || s->trace_mode == 2 || s->trace_mode == 4 || s->trace_mode == 5 || s->trace_mode == 6) {
        // trees in HBM: only the first MSK_STACK_CAP entries of a lane's stack live in LDS, the rest in an HBM overflow
        // array (LaneStack) — any tree depth works within a fixed 16 KB (+ 4 KB of node4_step scratch) of LDS per block, which
        // leaves room for six blocks per CU.  (Measured: the cap does not change the trace time between 8 and 40 entries.)
        ds.stack_entries = std::min(ds.stack_total, std::max(4u, env_u32("MSK_STACK_CAP", 16) & ~3u));
        s->trace_lds_bytes = (size_t) ds.stack_entries * MSK_BLOCK * 4 + (size_t) MSK_BLOCK * 16;       // + four words per lane (node4_step)
    }
    // LDS plan of k_shade_gen: the small lookup tables (tri_verts, mesh/bsdf/emitter records, cdf, d65, cie)
    const size_t table_bytes = ((size_t) ds.n_tris * 6 + ds.n_meshes + ds.n_bsdf_f4 + ds.n_emitters * 3 +
                                (ds.n_emitters * 95 + 3) / 4 + (ds.cdf_len + 3) / 4 + 72 + (ds.n_spectra + 3) / 4) * 16;
    s->lds_tables = table_bytes <= 40 * 1024;
    // + the waves' done-queues (k_shade_gen: MSK_DONE_Q_F4 float4 per wave)
    // (a scene whose per-triangle tables stay in HBM still stages the small ones: msk_kernels.h, small_tables_float4s — the same formula)
    const size_t small_bytes = ((size_t) ds.n_meshes + ds.n_bsdf_f4 + ds.n_emitters * 3 + (ds.n_emitters * 95 + 3) / 4 + (ds.cdf_len + 3) / 4 + 72 + (ds.n_spectra + 3) / 4) * 16;
    const size_t small_staged = small_bytes <= (size_t) MSK_SMALL_TABLES_KB * 1024 ? small_bytes : 0;
    s->shade_lds_bytes = (s->lds_tables ? table_bytes : small_staged) + (size_t) (MSK_BLOCK / MSK_WAVE) * MSK_DONE_Q_F4 * 16;
    if (s->trace_lds_bytes > ctx->prop.sharedMemPerBlock) {
        delete s;
        return fail(ctx, MSK_ERR_UNSUPPORTED, "BVH depth %d needs %zu B of traversal stack per block", bvh.max_depth, stack_bytes);
    }
    // per-triangle shading constants, computed on the device by the code the per-hit path used to run (k_tri_frames)
    {
        hipError_t ef = s->tri_frames.alloc(std::max<size_t>((size_t) ds.n_tris * 48, 16));
        if (ef != hipSuccess) { delete s; return fail(ctx, MSK_ERR_OOM, "scene upload: %s", hipGetErrorString(ef)); }
        ds.tri_frames = s->tri_frames.as<float4>();
        if (ds.n_tris) {
            hipLaunchKernelGGL(k_tri_frames, dim3((ds.n_tris + MSK_BLOCK - 1) / MSK_BLOCK), dim3(MSK_BLOCK), 0, ctx->stream, ds,
                               s->tri_frames.as<float4>());
            ef = hipStreamSynchronize(ctx->stream);
            if (ef != hipSuccess) { delete s; return fail(ctx, MSK_ERR_HIP, "k_tri_frames: %s", hipGetErrorString(ef)); }
        }
    }
    *out = s;
    return MSK_OK;
}

void free_workspace(msk_scene *scene);
extern "C" void msk_gpu_scene_destroy(msk_scene *scene) {
    if (!scene) return;
    if (scene->ctx->group) { group_scene_destroy(scene); return; }
    if (scene->ctx->lost) return;         // hipFree waits for the device: under a kernel that never finished it would wait for ever; the memory goes with the process
    (void) hipSetDevice(scene->ctx->device);
    free_workspace(scene);
    delete scene;
}

// ------------------------------------------------------------------------------------------
// spiral block schedule — BlockGenerator (imageblock.cpp:176-247)
// ------------------------------------------------------------------------------------------
struct HostBlock { int off_x, off_y, size_x, size_y, bx, by; };
static std::vector<HostBlock> spiral_blocks(int w, int h, int bs, int *nbx, int *nby) {
    const int bx = (int) std::ceil(w / (float) bs), by = (int) std::ceil(h / (float) bs);
    *nbx = bx; *nby = by;
    const int count = bx * by;
    std::vector<HostBlock> out;
    out.reserve(count);
    int dir = 0, px = bx / 2, py = by / 2, steps_left = 1, steps = 1;
    for (int counter = 0; counter < count;) {
        const int ox = px * bs, oy = py * bs;
        out.push_back(HostBlock{ox, oy, std::min(w - ox, bs), std::min(h - oy, bs), px, py});
        ++counter;
        if (counter == count) break;
        do {
            switch (dir) { case 0: ++px; break; case 1: ++py; break; case 2: --px; break; default: --py; break; }
            if (--steps_left == 0) {
                dir = (dir + 1) % 4;
                if (dir == 2 || dir == 0) ++steps;
                steps_left = steps;
            }
        } while (px < 0 || py < 0 || px >= bx || py >= by);
    }
    return out;
}

// ------------------------------------------------------------------------------------------
// wavefront driver
// ------------------------------------------------------------------------------------------
struct StateBufs {
    DevBuf id, wl, thr, res, ray_o, ray_d, sh, contrib, hit, aux, counts, ctrl, stack_ovf[MSK_MAX_STREAMS];
    PathState st;
    hipError_t alloc(size_t n, uint32_t n_regions) {      // n = slots a sweep can hold live; every region has two halves of them
        hipError_t e;
#define A_(b, sz) if ((e = b.reserve(2 * n * (sz))) != hipSuccess) return e;
        A_(id, 8) A_(wl, 16) A_(thr, 16) A_(res, 16) A_(ray_o, 16) A_(ray_d, 16) A_(sh, 16) A_(contrib, 16) A_(hit, 16)
        A_(aux, 8)
#undef A_
        if ((e = counts.reserve((size_t) n_regions * sizeof(RegionCtl))) != hipSuccess) return e;
        if ((e = ctrl.reserve(MSK_MAX_STREAMS * sizeof(Ctrl))) != hipSuccess) return e;
        st.id = id.as<uint2>(); st.wl = wl.as<float4>(); st.thr = thr.as<float4>(); st.res = res.as<float4>();
        st.ray_o = ray_o.as<float4>(); st.ray_d = ray_d.as<float4>(); st.sh = sh.as<float4>();
        st.contrib = contrib.as<float4>(); st.hit = hit.as<float4>(); st.aux = aux.as<float2>();
        return hipSuccess;
    }
    void leak() { for (DevBuf *b : {&id, &wl, &thr, &res, &ray_o, &ray_d, &sh, &contrib, &hit, &aux, &counts, &ctrl}) b->leak(); for (DevBuf &b : stack_ovf) b.leak(); }
};

struct Workspace {
    StateBufs sb;
    // Single-pass renders repeated with the same shape (bench, animation frames) reuse the uploaded plan: the pass pixel
    // table (4 MB at 512^2), block tables and the regions' initial share of the samples.
    std::vector<uint64_t> plan_key;          // empty = nothing cached
    uint64_t plan_n_pix = 0;
    DevBuf counts_init; unsigned long long counts_total = 0; uint32_t counts_regions = 0;
    DevBuf block_buf, blocks, block_of, spiral, pix, pix_inv, rec_a, rec_b, film, bands;
    void *host_film = nullptr; size_t host_film_bytes = 0;       // pinned staging buffer of msk_gpu_render's film copy-back
    ~Workspace() { if (host_film) (void) hipHostFree(host_film); }
    uint32_t n_bands = 0;
    DevBuf aov_rec[MSK_MAX_AOV_GROUPS + 1], aov_block_buf[MSK_MAX_AOV_GROUPS + 1];   // [n_groups] = the nested path's RGB
};

// what msk_gpu_render_aov asked for, in record groups of three channels (see AovParams)
struct AovPlan {
    uint32_t n_channels = 0;                 // film channels after X,Y,Z,A,W
    uint32_t n_groups = 0;                   // primary-hit groups
    uint32_t code[MSK_MAX_AOV_GROUPS] = {};
    int32_t out_ch[MSK_MAX_AOV_GROUPS + 1][5];   // film channel of each block channel, -1 = unused
    bool rgba = false;                       // group n_groups: nested path integrator R,G,B (+ A = the weight sum)
};

static uint32_t env_u32(const char *name, uint32_t def) {
    const char *v = getenv(name);
    return v && *v ? (uint32_t) strtoul(v, nullptr, 10) : def;
}

struct EventPool {
    msk_ctx *ctx; size_t next = 0; std::vector<hipEvent_t> *pool = nullptr;       // pool: ctx->events unless told otherwise
    // nullptr when the runtime cannot create another event: the dispatch then simply carries no timestamps
    // (hipExtLaunchKernelGGL takes null events) and the statistics miss that launch; nothing else depends on events
    hipEvent_t get() {
        std::vector<hipEvent_t> &v = pool ? *pool : ctx->events;
        while (next >= v.size()) {
            hipEvent_t e = nullptr;
            if (hipEventCreate(&e) != hipSuccess || !e) { (void) hipGetLastError(); return nullptr; }
            v.push_back(e);
        }
        return v[next++];
    }
};

void free_workspace(msk_scene *scene) { delete scene->ws; scene->ws = nullptr; }

static void sum_events(std::vector<std::pair<hipEvent_t, hipEvent_t>> &v, float *ms) {
    for (auto &p : v) { float t = 0; if (p.first && p.second && hipEventElapsedTime(&t, p.first, p.second) == hipSuccess) *ms += t; }
}

// Every wait of a render for its device outside the wavefront loop (which has wait_any): the wait mode of MSK_WAIT and the wall
// limit of MSK_WATCHDOG_S x `limit_scale` (the loop's limit is per sync group — a few milliseconds of work; MSK_RNG_PCG_BLOCK
// renders in ONE kernel that legitimately runs for as long as the job takes, and passes 30: an hour at the default).  A wait
// that runs out loses the context exactly as one inside the loop does.  Calling thread only.
static int ctx_sync(msk_ctx *ctx, hipStream_t stream, const char *what, double limit_scale) {
    mskwd::Limits lim = mskwd::limits_from_env();
    lim.wall_s *= limit_scale;
    const mskwd::Progress p(lim);
    const mskwd::Waiter w{mskwd::wait_mode_from_env(), ctx->hub};
    mskwd::Ticket t;
    t.stream = stream; t.slot = 0; t.event = ctx->wait_events[0];
    HIP_TRY(ctx, mskwd::arm(w, t));
    mskwd::Ticket *tp = &t;
    hipError_t es = hipSuccess;
    if (mskwd::wait_any(w, &tp, 1, p, &es) < 0) {
        ctx->lost = true;
        return fail(ctx, MSK_ERR_HIP, "no progress: %s did not finish within %g s (MSK_WATCHDOG_S); the context is lost", what, lim.wall_s);
    }
    HIP_TRY(ctx, es);
    return MSK_OK;
}

// t0 / t1: events that take the kernel's own start / end timestamps (hipExtLaunchKernelGGL: no extra packets in the queue,
// unlike hipEventRecord, which cost 2 % of a bench step at three records per iteration), or nullptr
static void launch_trace(msk_scene *sc, hipStream_t stream, const PathState &st, const PassParams &pp, hipEvent_t t0 = nullptr, hipEvent_t t1 = nullptr) {
    const uint32_t grid = (pp.region_count * MSK_WAVE + MSK_BLOCK - 1) / MSK_BLOCK;
    // Lane replacement pays when rays are long (tree in HBM/L2: trace -35 % on the 70 k-triangle scene) and costs when they
    // are short (LDS-resident cbox: +50 %): on by default for modes 1 and 2 only.  MSK_TRACE_REFILL=0 turns it off.
    const int refill_env = getenv("MSK_TRACE_REFILL") ? atoi(getenv("MSK_TRACE_REFILL")) : -1;
    const int max_inner = (int) env_u32("MSK_TRACE_QUANTUM", 3);       // (4 until round 5's better tree: 10.7 instead of 12.9 node visits per ray)
    const int refill = (sc->trace_mode == 3) ? 0 : refill_env >= 0 ? refill_env : (sc->trace_mode == 0 ? 0 : 16);
    const size_t lds = sc->trace_lds_bytes + (size_t) env_u32("MSK_TRACE_PAD_LDS_KB", 0) * 1024;      // occupancy experiments only
    if (refill > 0) {        // k_trace_r
        if (sc->trace_mode == 4) hipExtLaunchKernelGGL(k_trace_r<4>, dim3(grid), dim3(MSK_BLOCK), lds, stream, t0, t1, 0, sc->dev, st, pp, refill, max_inner);
        else if (sc->trace_mode == 0) hipExtLaunchKernelGGL(k_trace_r<0>, dim3(grid), dim3(MSK_BLOCK), lds, stream, t0, t1, 0, sc->dev, st, pp, refill, max_inner);
        else if (sc->trace_mode == 1) hipExtLaunchKernelGGL(k_trace_r<1>, dim3(grid), dim3(MSK_BLOCK), lds, stream, t0, t1, 0, sc->dev, st, pp, refill, max_inner);
        else if (sc->trace_mode == 5) hipExtLaunchKernelGGL(k_trace_r<5>, dim3(grid), dim3(MSK_BLOCK), lds, stream, t0, t1, 0, sc->dev, st, pp, refill, max_inner);
        else if (sc->trace_mode == 6) hipExtLaunchKernelGGL(k_trace_r<6>, dim3(grid), dim3(MSK_BLOCK), lds, stream, t0, t1, 0, sc->dev, st, pp, refill, max_inner);
        else hipExtLaunchKernelGGL(k_trace_r<2>, dim3(grid), dim3(MSK_BLOCK), lds, stream, t0, t1, 0, sc->dev, st, pp, refill, max_inner);
        return;
    }
    // LDS-resident scenes: k_trace_q (job queues with lane replacement, refill when 32 lanes are idle); MSK_TRACE_QUEUE=0: k_trace<0>
    const uint32_t queue_refill = refill_env >= 0 ? 0u : std::min(64u, env_u32("MSK_TRACE_QUEUE", 32));
    const size_t bits_off = (lds + 15) & ~(size_t) 15;
    const size_t lds_q = bits_off + (size_t) (MSK_BLOCK / MSK_WAVE) * (pp.region_size / 8);       // + one bit per slot and wave
    if (sc->trace_mode == 0 && queue_refill && sc->lds_scene && lds_q <= 64 * 1024) {
        const uint32_t grid_s = (pp.region_count * pp.trace_split * MSK_WAVE + MSK_BLOCK - 1) / MSK_BLOCK;
        hipExtLaunchKernelGGL(k_trace_q, dim3(grid_s), dim3(MSK_BLOCK), lds_q, stream, t0, t1, 0, sc->dev, st, pp, queue_refill, (uint32_t) (bits_off / 16));
        return;
    }
    if (sc->trace_mode == 0) {     // pp.trace_split waves per region (LDS-resident scene: no stack overflow array to size).
        // Measured: 2 waves per region -6 % trace on the cbox (twice the waves to balance the tail of a launch), 4 the same.
        const uint32_t grid_s = (pp.region_count * pp.trace_split * MSK_WAVE + MSK_BLOCK - 1) / MSK_BLOCK;
        hipExtLaunchKernelGGL(k_trace<0>, dim3(grid_s), dim3(MSK_BLOCK), lds, stream, t0, t1, 0, sc->dev, st, pp);
    }
    else if (sc->trace_mode == 1) hipExtLaunchKernelGGL(k_trace<1>, dim3(grid), dim3(MSK_BLOCK), lds, stream, t0, t1, 0, sc->dev, st, pp);
    else if (sc->trace_mode == 2) hipExtLaunchKernelGGL(k_trace<2>, dim3(grid), dim3(MSK_BLOCK), lds, stream, t0, t1, 0, sc->dev, st, pp);
    else if (sc->trace_mode == 4) hipExtLaunchKernelGGL(k_trace<4>, dim3(grid), dim3(MSK_BLOCK), lds, stream, t0, t1, 0, sc->dev, st, pp);
    else if (sc->trace_mode == 5) hipExtLaunchKernelGGL(k_trace<5>, dim3(grid), dim3(MSK_BLOCK), lds, stream, t0, t1, 0, sc->dev, st, pp);
    else if (sc->trace_mode == 6) hipExtLaunchKernelGGL(k_trace<6>, dim3(grid), dim3(MSK_BLOCK), lds, stream, t0, t1, 0, sc->dev, st, pp);
    else hipExtLaunchKernelGGL(k_trace<3>, dim3(grid), dim3(MSK_BLOCK), lds, stream, t0, t1, 0, sc->dev, st, pp);
}

// Renders the samples of `pix` (pass pixel table, host) into records; leaves records on device.
static int run_wavefront(msk_scene *sc, hipStream_t stream, const msk_render_params *prm, uint32_t spp_owned,
                         const uint4 *d_pix, const uint32_t *d_pix_to_j, uint64_t n_pix, float4 *rec_a, float *rec_b, StateBufs &sb,
                         uint32_t region_size, uint32_t n_regions, msk_stats *stats, EventPool &ev,
                         std::vector<std::pair<hipEvent_t, hipEvent_t>> &ev_trace,
                         std::vector<std::pair<hipEvent_t, hipEvent_t>> &ev_shade,
                         const AovParams *aov = nullptr, float4 *aov_rgb = nullptr, bool packed = false) {
    msk_ctx *ctx = sc->ctx;
    int rc_sync = MSK_OK;
    const unsigned long long total = (unsigned long long) n_pix * spp_owned;
    // static, interleaved partition of the pass's samples over the regions (see RegionCtl); the initial records are kept
    // on the device and copied from there when the next render has the same shape
    Workspace *wsp = sc->ws;
    if (!(wsp && wsp->counts_total == total && wsp->counts_regions == n_regions && wsp->counts_init.p)) {
        std::vector<RegionCtl> init(n_regions);
        std::memset(init.data(), 0, init.size() * sizeof(RegionCtl));
        const unsigned long long n_chunks = (total + 63) / 64;
        for (uint32_t r = 0; r < n_regions; ++r) {
            const unsigned long long mine = n_chunks > r ? (n_chunks - r + n_regions - 1) / n_regions : 0;
            unsigned long long n = mine * 64;
            if (mine && (mine - 1) * n_regions + r == n_chunks - 1) n -= n_chunks * 64 - total;   // partial last chunk
            init[r].next_sample = 0; init[r].end_sample = n;
        }
        if (wsp) {
            HIP_TRY(ctx, wsp->counts_init.reserve(init.size() * sizeof(RegionCtl)));
            HIP_TRY(ctx, hipMemcpy(wsp->counts_init.p, init.data(), init.size() * sizeof(RegionCtl), hipMemcpyHostToDevice));
            wsp->counts_total = total; wsp->counts_regions = n_regions;
        } else {
            HIP_TRY(ctx, hipMemcpyAsync(sb.counts.p, init.data(), init.size() * sizeof(RegionCtl), hipMemcpyHostToDevice, stream));
            if ((rc_sync = ctx_sync(ctx, stream, "the upload of the regions' records"))) return rc_sync;
        }
    }
    if (wsp) HIP_TRY(ctx, hipMemcpyAsync(sb.counts.p, wsp->counts_init.p, (size_t) n_regions * sizeof(RegionCtl), hipMemcpyDeviceToDevice, stream));
    PassParams pp0;
    pp0.seed = prm->seed; pp0.spp_owned = spp_owned; pp0.sample_first = prm->sample_first;
    pp0.sample_stride = prm->sample_stride ? prm->sample_stride : 1;
    pp0.rr_depth = prm->rr_depth; pp0.max_depth = prm->max_depth; pp0.hide_emitters = prm->hide_emitters;
    pp0.pix_table = d_pix; pp0.pix_to_j = d_pix_to_j; pp0.rec_a = rec_a; pp0.rec_b = rec_b;
    pp0.region_size = region_size; pp0.n_regions = n_regions; pp0.regions = sb.counts.as<RegionCtl>();
    pp0.region_first = 0; pp0.region_count = n_regions;
    pp0.trace_split = sc->trace_mode == 0 ? std::max(1u, env_u32("MSK_TRACE_SPLIT", 2)) : 1u;
    pp0.aov_rgb = aov_rgb;
    pp0.aov_groups = aov ? aov->n_groups : 0u;
    for (uint32_t g = 0; g < MSK_MAX_AOV_GROUPS; ++g) pp0.aov_rec[g] = aov && g < aov->n_groups ? aov->rec[g] : nullptr;
    pp0.packed = packed ? 1u : 0u;
    pp0.stack_ovf = nullptr;
    // (MSK_FORCE_GENERAL_SHADE=1, measurements only: an all-diffuse scene through the general variant — what a per-class diffuse
    // instantiation could save a mixed scene's diffuse chunks)
    const bool force_general = env_u32("MSK_FORCE_GENERAL_SHADE", 0) != 0;
    // (the AOV RGB record and the validity test over an "aov" render's record groups live in the general shading variant)
    const bool diffuse_only = sc->all_diffuse && !aov_rgb && !(aov && aov->n_groups) && !force_general;
    // material-sorted shading (general variant): LDS for the permutation, 3 bytes per slot of a region and wave (MSK_SORT=0: off)
    const size_t sort_lds = (size_t) (MSK_BLOCK / MSK_WAVE) * 3 * region_size;
    const bool sort_on = !diffuse_only && (!sc->all_diffuse || force_general) && region_size <= 4096 && env_u32("MSK_SORT", 1) &&
                         sc->shade_lds_bytes + sort_lds <= 64 * 1024;
    pp0.sort_scratch = sort_on ? 1u : 0u;
    const size_t shade_lds = sc->shade_lds_bytes + (sort_on ? sort_lds : 0);
    const uint32_t group = env_u32("MSK_SYNC_GROUP", 8);
    static const size_t shade_pad_lds = (size_t) env_u32("MSK_SHADE_PAD_LDS_KB", 0) * 1024;   // occupancy experiments only
    const bool timing = stats != nullptr;
    // A timed dispatch costs ~6 us more than an untimed one (completion signal + timestamps): 2 % of a bench step when every
    // launch is timed.  MSK_TIMING_EVERY=n times the launches of every n-th sync group only (rotating from render to render so
    // that repeated renders cover all groups); msk_stats::ms_trace / ms_shade / n_*_launches then describe that sample.
    const uint32_t every = std::max(1u, env_u32("MSK_TIMING_EVERY", 1));
    const uint32_t phase = ctx->timing_phase++;
    // k_wavefront (iterations on the device): possible when tables and tree are LDS-resident and everything fits one block's LDS
    // next to each other, and there is no per-iteration AOV kernel.  MSK_FUSED=1: the whole pass; MSK_FUSED_TAIL_PCT=p: from
    // the point where every sample has been started and fewer than p % of the slots are live.
    const uint32_t fused_queue_f4 = (uint32_t) ((sc->shade_lds_bytes - (size_t) (MSK_BLOCK / MSK_WAVE) * MSK_DONE_Q_F4 * 16) / 16);     // after the staged tables
    const uint32_t fused_trace_f4 = (uint32_t) (sc->shade_lds_bytes / 16);
    const size_t fused_lds = sc->shade_lds_bytes + sc->trace_lds_bytes;
    // ... and k_wavefront_h for the default tree in HBM (trace mode 6): the VERY thin end only (MSK_FUSED_HBM=0: off).  Round 6, same
    // box, config-5 / config-3 class renders: from 10 % live slots on (the LDS-resident scenes' threshold) 117.9 / 137.1 ms against
    // 116.2 / 138.6 without — a wave's own longest rays bound both, and the fused kernel walks them at two waves per SIMD without lane
    // replacement; from 1-3 % on — the ~40 last iterations, whose launches are a few dozen microseconds of latency each —
    // 118.1-118.2 / 143.9-144.2 ms against 119.2 / 146.2 (profiles/r06_ab_fused_hbm.txt): the default, at 2 %.
    const bool fused_h = sc->trace_mode == 6 && !sc->lds_tables && env_u32("MSK_FUSED_HBM", 1) != 0;
    const bool fused_ok = ((sc->trace_mode == 0 && sc->lds_tables) || fused_h) && fused_lds <= 64 * 1024 && !(aov && aov->n_groups);
    const bool fused_all = fused_ok && env_u32("MSK_FUSED", 0) != 0;
    const uint32_t fused_iters = std::max(1u, env_u32("MSK_FUSED_ITERS", 16));
    const uint32_t fused_tail_pct = fused_ok ? env_u32("MSK_FUSED_TAIL_PCT", fused_h ? 2 : 10) : 0u;

    // The wavefront loop over the regions [first, first + count) on one stream.  The pool's two halves run this at the same
    // time on two streams (two host threads): regions are independent — each owns its slots, its share of the samples and
    // its counters — and a shading launch of one half fills the gaps of a traversal launch of the other (and the other way
    // round), which one launch at a time leaves open at its start, its end and wherever its waves wait.  Measured with two
    // concurrent half-size renders before this was built: 49.3 against 55.0 ms for the bench step.
    // One part of the pool: the regions [first, first + count) and their wavefront loop on one stream.  The parts' loops run at the
    // same time: regions are independent — each owns its slots, its share of the samples and its counters — and a shading
    // launch of one part fills the gaps of a traversal launch of another (and the other way round), which one launch at a time
    // leaves open at its start, its end and wherever its waves wait.  Measured with two concurrent half-size renders before
    // this was built: 49.3 against 55.0 ms for the bench step.  WHO drives the loops is a separate choice (MSK_HOST_THREADS below).
    struct Part {
        uint32_t first = 0, count = 0; hipStream_t stream = nullptr; Ctrl *d_ctrl = nullptr, *h_ctrl = nullptr; EventPool ev{nullptr}; uint32_t *stack_ovf = nullptr;
        msk_stats st; int rc = MSK_OK; unsigned long long expected = 0; std::string err; bool lost = false;
        // the loop's state between two sync groups
        PassParams pp; uint32_t grid = 0, it = 0, gi = 0, parity = 0, last_iters = 0; bool fused_now = false, done = false; size_t ev_mark = 0;
        // Two alternating sets of events: a group's timestamps are read (hipEventElapsedTime is a host call of a few
        // microseconds, 32 of them per group) after the NEXT group has been queued, not while the GPU waits for work.
        std::vector<std::pair<hipEvent_t, hipEvent_t>> cur_shade, cur_trace, pend_shade, pend_trace;
        mskwd::Progress watchdog;                          // msk_watchdog.h: a wall limit per sync group + "the counters stand still"
        mskwd::Ticket ticket;
        explicit Part(const mskwd::Limits &l) : watchdog(l) { std::memset(&st, 0, sizeof st); }
    };
    const mskwd::Limits wd_limits = mskwd::limits_from_env();
    const mskwd::Waiter waiter{mskwd::wait_mode_from_env(), ctx->hub};
    auto read_pending = [&](Part &p) {
        if (!timing) return;
        sum_events(p.pend_shade, &p.st.ms_shade); sum_events(p.pend_trace, &p.st.ms_trace);
        p.st.n_shade_launches += (uint32_t) p.pend_shade.size(); p.st.n_trace_launches += (uint32_t) p.pend_trace.size();
        p.pend_shade.clear(); p.pend_trace.clear();
    };
    // MSK_DUMP_RAYS=<file> (+ MSK_DUMP_ITER, default 12; MSK_DUMP_STRIDE, default 32), measurements only: the rays the traversal
    // launch of that iteration is about to walk — every MSK_DUMP_STRIDE-th region of the first part, its live slots in the
    // order the kernel takes them (shadow-carrying slots first) — written to the file for the host-side scheduler model
    // (tools/micro/sched_model.cpp).  Header {'MSKR', regions, region_size, stride}; per region {count, ns}, then count x
    // {ray_o, ray_d (w = tmax as the kernel reads it), sh} float4.
    const char *dump_path = getenv("MSK_DUMP_RAYS");
    const uint32_t dump_iter = env_u32("MSK_DUMP_ITER", 12), dump_stride = std::max(1u, env_u32("MSK_DUMP_STRIDE", 32));
    auto dump_rays = [&](Part &p) {
        if (hipStreamSynchronize(p.stream) != hipSuccess) return;
        std::vector<RegionCtl> ctl(p.count);
        if (hipMemcpy(ctl.data(), sb.counts.as<RegionCtl>() + p.first, p.count * sizeof(RegionCtl), hipMemcpyDeviceToHost) != hipSuccess) return;
        FILE *f = std::fopen(dump_path, "wb");
        if (!f) return;
        const uint32_t n_dump = (p.count + dump_stride - 1) / dump_stride;
        const uint32_t head[4] = {0x524b534du, n_dump, region_size, dump_stride};
        std::fwrite(head, 4, 4, f);
        std::vector<float4> ho(2 * (size_t) region_size), hd(2 * (size_t) region_size), hs(2 * (size_t) region_size);
        for (uint32_t r = 0; r < p.count; r += dump_stride) {
            const size_t base = (size_t) (p.first + r) * 2 * region_size;
            (void) hipMemcpy(ho.data(), sb.st.ray_o + base, ho.size() * 16, hipMemcpyDeviceToHost);
            (void) hipMemcpy(hd.data(), sb.st.ray_d + base, hd.size() * 16, hipMemcpyDeviceToHost);
            (void) hipMemcpy(hs.data(), sb.st.sh + base, hs.size() * 16, hipMemcpyDeviceToHost);
            const uint32_t count = ctl[r].count, ns = ctl[r].half_ns >> 1, half = ctl[r].half_ns & 1u;
            const uint32_t cn[2] = {count, ns};
            std::fwrite(cn, 4, 2, f);
            for (uint32_t c = 0; c < count; ++c) {
                const size_t slot = (size_t) half * region_size + (c < ns ? c : region_size - 1 - (c - ns));
                float4 rec[3] = {ho[slot], hd[slot], c < ns ? hs[slot] : make_float4(0, 0, 0, 0)};
                if (std::signbit(rec[1].w)) rec[1].w = INFINITY;          // slot_tmax: a bounce ray keeps -pdf there
                std::fwrite(rec, 16, 3, f);
            }
        }
        std::fclose(f);
    };
    // queues one sync group of the part's loop: `group` iterations (or one k_wavefront launch), the counters' reduction and
    // their copy to the host, and whatever the wait mode needs behind them
    auto queue_group = [&](Part &p) -> int {
        const PassParams &pp = p.pp;
        hipStream_t stream_h = p.stream;
        const uint32_t grid = p.grid;
        p.ev.next = p.ev_mark + (size_t) p.parity * 4 * group;
        const bool timed = timing && (p.gi + phase) % every == 0;
        if (p.fused_now) {
            // the iteration loop on the device (k_wavefront): one bounded launch = up to fused_iters sweeps of every region
            // (not timed: msk_stats::ms_shade / ms_trace stay the sums of k_shade_gen / k_trace launches)
            hipEvent_t a = nullptr, b = nullptr;
            if (fused_h) {
                if (diffuse_only) hipExtLaunchKernelGGL((k_wavefront_h<true>), dim3(grid), dim3(MSK_BLOCK), fused_lds, stream_h, a, b, 0, sc->dev, sb.st, pp, fused_iters, fused_queue_f4, fused_trace_f4);
                else if (sc->has_regular) hipExtLaunchKernelGGL((k_wavefront_h<false, true>), dim3(grid), dim3(MSK_BLOCK), fused_lds, stream_h, a, b, 0, sc->dev, sb.st, pp, fused_iters, fused_queue_f4, fused_trace_f4);
                else hipExtLaunchKernelGGL((k_wavefront_h<false>), dim3(grid), dim3(MSK_BLOCK), fused_lds, stream_h, a, b, 0, sc->dev, sb.st, pp, fused_iters, fused_queue_f4, fused_trace_f4);
            }
            else if (diffuse_only) hipExtLaunchKernelGGL((k_wavefront<true>), dim3(grid), dim3(MSK_BLOCK), fused_lds, stream_h, a, b, 0, sc->dev, sb.st, pp, fused_iters, fused_queue_f4, fused_trace_f4);
            else if (sc->has_regular) hipExtLaunchKernelGGL((k_wavefront<false, true>), dim3(grid), dim3(MSK_BLOCK), fused_lds, stream_h, a, b, 0, sc->dev, sb.st, pp, fused_iters, fused_queue_f4, fused_trace_f4);
            else hipExtLaunchKernelGGL((k_wavefront<false>), dim3(grid), dim3(MSK_BLOCK), fused_lds, stream_h, a, b, 0, sc->dev, sb.st, pp, fused_iters, fused_queue_f4, fused_trace_f4);
            p.it += fused_iters; p.last_iters = fused_iters;
            p.st.launches_wavefront += 1;
        } else {
            for (uint32_t g = 0; g < group; ++g, ++p.it) {
                hipEvent_t a = nullptr, b = nullptr, c = nullptr, d = nullptr;
                if (timed) { a = p.ev.get(); b = p.ev.get(); c = p.ev.get(); d = p.ev.get(); }
                const bool have_ev = a && b && c && d;
#define MSK_SHADE(...) hipExtLaunchKernelGGL((k_shade_gen<__VA_ARGS__>), dim3(grid), dim3(MSK_BLOCK), shade_lds + shade_pad_lds, stream_h, a, b, 0, sc->dev, sb.st, pp)
                if (sc->lds_tables) { if (diffuse_only) MSK_SHADE(true, true); else if (sc->has_regular) MSK_SHADE(true, false, true); else MSK_SHADE(true, false); }
                else { if (diffuse_only) MSK_SHADE(false, true); else if (sc->has_regular) MSK_SHADE(false, false, true); else MSK_SHADE(false, false); }
#undef MSK_SHADE
                if (dump_path && p.it == dump_iter && p.first == 0) dump_rays(p);      // (measurements only: MSK_DUMP_RAYS)
                launch_trace(sc, stream_h, sb.st, pp, c, d);
                p.st.launches_shade += 1; p.st.launches_trace += 1;
                if (aov && aov->n_groups) hipLaunchKernelGGL(k_aov_primary, dim3(grid), dim3(MSK_BLOCK), 0, stream_h, sc->dev, sb.st, pp, *aov);
                if (timed && have_ev) { p.cur_shade.push_back({a, b}); p.cur_trace.push_back({c, d}); }
            }
            p.last_iters = group;
        }
        read_pending(p);                    // the previous group's, while this one runs
        HIP_TRY_SLOT(&p.err, hipMemsetAsync(p.d_ctrl, 0, sizeof(Ctrl), stream_h));
        hipLaunchKernelGGL(k_reduce_ctl, dim3(std::min(64u, (p.count + MSK_BLOCK - 1) / MSK_BLOCK)), dim3(MSK_BLOCK), 0, stream_h,
                           sb.counts.as<RegionCtl>() + p.first, p.count, p.d_ctrl);
        HIP_TRY_SLOT(&p.err, hipGetLastError());
        HIP_TRY_SLOT(&p.err, hipMemcpyAsync(p.h_ctrl, p.d_ctrl, sizeof(Ctrl), hipMemcpyDeviceToHost, stream_h));
        HIP_TRY_SLOT(&p.err, mskwd::arm(waiter, p.ticket));
        return MSK_OK;
    };
    // after the group's last command has completed: is the part done, stalled, or ready for its thin end?
    auto finish_group = [&](Part &p) -> int {
        p.pend_shade.swap(p.cur_shade); p.pend_trace.swap(p.cur_trace); p.cur_shade.clear(); p.cur_trace.clear(); p.parity ^= 1u; ++p.gi;
        const Ctrl &h = *p.h_ctrl;
        if (h.remaining == 0 && h.live == 0) {
            read_pending(p);
            p.ev.next = p.ev_mark;                            // every timestamp has been read: the events are free again
            p.st.samples = h.samples_done; p.st.segments = h.segments; p.st.shadow_rays = h.shadow_rays;
            p.st.invalid_samples = h.invalid;
            p.st.iterations = p.it;
            p.done = true;
            if (h.samples_done != p.expected)
                return fail_to(&p.err, MSK_ERR_HIP, "internal error: %llu of %llu samples finished", h.samples_done, p.expected);
            return MSK_OK;
        }
        if (p.watchdog.group_done(mskwd::Counters{h.samples_done, h.segments, h.remaining, h.live}) == mskwd::STALLED) {
            p.lost = true;
            return fail_to(&p.err, MSK_ERR_HIP, "no progress: %u sync groups of the wavefront loop changed nothing (%llu samples finished, %llu live paths, "
                           "%llu samples not started; MSK_WATCHDOG_GROUPS); the context is lost", p.watchdog.stalled(), h.samples_done, h.live, h.remaining);
        }
        if (fused_ok && !p.fused_now && h.remaining == 0 && h.live * 100ull < (unsigned long long) fused_tail_pct * p.count * region_size)
            p.fused_now = true;                       // the thinning end of the pass: no more launches and host round trips per sweep
        if (p.it > 100000000u) return fail_to(&p.err, MSK_ERR_HIP, "wavefront loop did not terminate");
        return MSK_OK;
    };
    // One host thread drives n of the parts: queues a group on each, then serves whichever finishes (wait_any), oldest first.
    auto drive = [&](Part *const *mine, int n) {
        (void) hipSetDevice(ctx->device);               // the current device is per host thread
        Part *active[MSK_MAX_STREAMS];
        int n_active = 0;
        for (int i = 0; i < n; ++i) {
            mine[i]->rc = queue_group(*mine[i]);
            if (mine[i]->rc == MSK_OK) active[n_active++] = mine[i];
        }
        while (n_active) {
            mskwd::Ticket *tickets[MSK_MAX_STREAMS];
            for (int i = 0; i < n_active; ++i) tickets[i] = &active[i]->ticket;
            hipError_t es = hipSuccess;
            const int hit = mskwd::wait_any(waiter, tickets, n_active, active[0]->watchdog, &es);
            if (hit < 0) {                              // the wall limit: nothing this thread waits for will be waited for again
                for (int i = 0; i < n_active; ++i) {
                    Part &q = *active[i];
                    q.lost = true;
                    q.rc = fail_to(&q.err, MSK_ERR_HIP, "no progress: a sync group of the wavefront loop (iterations %u..%u, regions %u..%u) did not finish within "
                                   "%g s (MSK_WATCHDOG_S); the context is lost", q.it - q.last_iters, q.it, q.first, q.first + q.count, q.watchdog.limits().wall_s);
                }
                return;
            }
            Part &p = *active[hit];
            for (int i = hit; i + 1 < n_active; ++i) active[i] = active[i + 1];
            --n_active;
            if (es != hipSuccess) { p.rc = fail_to(&p.err, MSK_ERR_HIP, "the wavefront loop's stream reported: %s", hipGetErrorString(es)); continue; }
            p.rc = finish_group(p);
            if (p.lost) return;                         // (stalled: the context is done, the other parts are not driven further)
            if (p.rc == MSK_OK && !p.done) {
                p.rc = queue_group(p);
                if (p.rc == MSK_OK) active[n_active++] = &p;
            }
        }
    };

    // samples the regions [first, first + count) own (the same static partition as the init above)
    auto share = [&](uint32_t first, uint32_t count) {
        const unsigned long long n_chunks = (total + 63) / 64;
        unsigned long long sum = 0;
        for (uint32_t r = first; r < first + count; ++r) {
            const unsigned long long mine = n_chunks > r ? (n_chunks - r + n_regions - 1) / n_regions : 0;
            unsigned long long n = mine * 64;
            if (mine && (mine - 1) * n_regions + r == n_chunks - 1) n -= n_chunks * 64 - total;
            sum += n;
        }
        return sum;
    };
    // how many loops: 4 by default (DESIGN.md §6 has 1 / 2 / 3 / 4); one for small jobs and caller-supplied streams
    uint32_t n_parts = std::min<uint32_t>(MSK_MAX_STREAMS, std::max(1u, env_u32("MSK_STREAMS", 4)));
    if (n_regions < 1024 || stream != ctx->stream) n_parts = 1;
    const uint32_t ovf_words = sc->dev.stack_total > sc->dev.stack_entries ? sc->dev.stack_total - sc->dev.stack_entries : 0;
    std::vector<Part> parts;
    parts.reserve(n_parts);
    // Parts of slightly different sizes: equal parts can fall into step (all launches starting and draining together, which
    // is one big launch again; measured as a bimodal 48 / 52 ms), unequal ones keep sliding past each other.
    const double skew = env_u32("MSK_STREAM_SKEW", 10) / 100.0;          // relative size step between neighbouring parts
    std::vector<double> cum(n_parts + 1, 0.0);
    for (uint32_t k = 0; k < n_parts; ++k) cum[k + 1] = cum[k] + 1.0 + skew * ((double) (n_parts - 1) / 2.0 - k);
    for (uint32_t k = 0; k < n_parts; ++k) {
        const uint32_t first = (uint32_t) (n_regions * (cum[k] / cum[n_parts])), last = k + 1 == n_parts ? n_regions : (uint32_t) (n_regions * (cum[k + 1] / cum[n_parts]));
        parts.emplace_back(wd_limits);
        Part &p = parts.back();
        p.first = first; p.count = last - first; p.stream = k ? ctx->more_streams[k - 1] : stream;
        p.d_ctrl = sb.ctrl.as<Ctrl>() + k; p.h_ctrl = ctx->h_ctrl + k;
        p.ev = k ? EventPool{ctx, 0, &ctx->more_events[k - 1]} : EventPool{ctx, ev.next, ev.pool};
        p.ev_mark = p.ev.next;
        p.expected = share(first, last - first);
        p.pp = pp0; p.pp.region_first = first; p.pp.region_count = p.count;
        p.grid = (p.count * MSK_WAVE + MSK_BLOCK - 1) / MSK_BLOCK;
        p.fused_now = fused_all;
        p.ticket.stream = p.stream; p.ticket.slot = (int) k; p.ticket.event = ctx->wait_events[k];
        if (ovf_words) {                                // LaneStack overflow: one word per lane per extra entry, per launch
            const size_t lanes = (size_t) (((last - first) * MSK_WAVE + MSK_BLOCK - 1) / MSK_BLOCK) * MSK_BLOCK;
            HIP_TRY(ctx, sb.stack_ovf[k].reserve((size_t) ovf_words * lanes * 4));
            p.stack_ovf = sb.stack_ovf[k].as<uint32_t>();
        }
        p.pp.stack_ovf = p.stack_ovf;
    }
    // Who drives the parts' loops (MSK_HOST_THREADS): ONE host thread for all of them by default since round 6 — it queues a
    // group on every stream and then serves whichever finishes; the device always holds the other streams' queued groups while
    // the host turns one around.  Rounds 2-5 used one thread per part (MSK_HOST_THREADS=4): three more threads per context,
    // spinning with the poll wait (DESIGN.md §7 has both under a CPU quota).
    const uint32_t n_threads = std::min(n_parts, std::max(1u, env_u32("MSK_HOST_THREADS", 1)));
    if (n_parts > 1) {
        rc_sync = ctx_sync(ctx, stream, "the regions' initial records");     // (queued above) before the other streams read them
        if (rc_sync) return rc_sync;
    }
    {
        std::vector<std::vector<Part *>> mine(n_threads);
        for (uint32_t k = 0; k < n_parts; ++k) mine[k % n_threads].push_back(&parts[k]);
        std::vector<std::thread> others;
        for (uint32_t j = 1; j < n_threads; ++j) others.emplace_back([&, j]() { drive(mine[j].data(), (int) mine[j].size()); });
        drive(mine[0].data(), (int) mine[0].size());
        for (auto &t : others) t.join();
    }
    for (auto &hf : parts) if (hf.lost) ctx->lost = true;                                 // the watchdog gave up on a part: the context is done
    for (auto &hf : parts) if (hf.rc) return fail(ctx, hf.rc, "%s", hf.err.c_str());     // first failing part, after the join
    if (stats) {
        for (const Part &hf : parts) {
            stats->samples += hf.st.samples; stats->segments += hf.st.segments; stats->shadow_rays += hf.st.shadow_rays;
            stats->invalid_samples += hf.st.invalid_samples;
            stats->iterations += hf.st.iterations;
            stats->ms_trace += hf.st.ms_trace; stats->ms_shade += hf.st.ms_shade;
            stats->n_trace_launches += hf.st.n_trace_launches; stats->n_shade_launches += hf.st.n_shade_launches;
            stats->launches_trace += hf.st.launches_trace; stats->launches_shade += hf.st.launches_shade; stats->launches_wavefront += hf.st.launches_wavefront;
            // ABI v7: the bytes of SoA path state this pass's launches were asked to move (counted from what the kernels
            // read and write per live slot, not measured): a segment = a slot that is live after a shading sweep.
            //   shading  176 B/segment (id 8, wl, thr, res, ray_d, hit in; id 8, wl, thr, res, ray_o, ray_d out; + aux 8 in / 8 out in the
            //            general variant) - 64 B/sample (a new sample's thr = 1 and res = 0 are neither written nor read)
            //            + 48 B/shadow ray (contrib in; sh, contrib out) + 20 B/sample (its record)
            //   traversal 48 B/segment (ray_o, ray_d in; hit out) + per shadow ray: sh in, and ray_o again where the shadow rays are a
            //            second queue of the launch (k_trace_q / k_trace: 32 B; k_trace_r walks both rays of a slot together: 16 B)
            const unsigned long long seg = hf.st.segments, smp = hf.st.samples, shd = hf.st.shadow_rays;
            const bool lane_refill = sc->trace_mode != 0 && sc->trace_mode != 3 && (getenv("MSK_TRACE_REFILL") ? atoi(getenv("MSK_TRACE_REFILL")) != 0 : true);
            stats->bytes_shade += seg * (176ull + (diffuse_only ? 0ull : 16ull)) + shd * 48ull + smp * 20ull - std::min(smp * 64ull, seg * 176ull);
            stats->bytes_trace += seg * 48ull + shd * (lane_refill ? 16ull : 32ull);
        }
    }
    (void) ev_trace; (void) ev_shade;
    return MSK_OK;
}

static int check_params(msk_ctx *ctx, const msk_render_params *p, int block_min) {
    if (!p) return fail(ctx, MSK_ERR_INVALID_ARG, "render params are NULL");
    if (p->rng_mode != MSK_RNG_COUNTER && p->rng_mode != MSK_RNG_PCG_BLOCK)
        return fail(ctx, MSK_ERR_INVALID_ARG, "rng_mode %d is neither MSK_RNG_COUNTER nor MSK_RNG_PCG_BLOCK", p->rng_mode);
    if (p->spp == 0) return fail(ctx, MSK_ERR_INVALID_ARG, "spp must be > 0");
    if (p->spp > (1u << MSK_DEPTH_SHIFT))
        return fail(ctx, MSK_ERR_UNSUPPORTED, "spp %u: at most %u samples per pixel and call (the path state holds 20 bits of sample index); "
                    "shard the samples with sample_first / sample_stride", p->spp, 1u << MSK_DEPTH_SHIFT);
    if (p->rr_depth <= 0) return fail(ctx, MSK_ERR_INVALID_ARG, "\"rr_depth\" must be set to a value greater than zero!");
    if (p->max_depth < 0 && p->max_depth != -1)
        return fail(ctx, MSK_ERR_INVALID_ARG, "\"max_depth\" must be set to -1 (infinite) or a value >= 0");
    // the path state holds 12 bits of depth: a path is cut after bounce MSK_MAX_DEPTH.  Unreachable with Russian roulette from a
    // depth below that (survival <= 0.95 per bounce); a bound or a roulette start beyond it would be cut silently, so refuse
    if (p->max_depth > (int) MSK_MAX_DEPTH)
        return fail(ctx, MSK_ERR_UNSUPPORTED, "max_depth %d: the path state holds bounces up to %u", p->max_depth, MSK_MAX_DEPTH);
    if (p->max_depth < 0 && p->rr_depth > (int) MSK_MAX_DEPTH)
        return fail(ctx, MSK_ERR_UNSUPPORTED, "rr_depth %d with unbounded max_depth: the path state holds bounces up to %u", p->rr_depth, MSK_MAX_DEPTH);
    if (p->block_size < block_min || p->block_size > 4096)        // (a pixel's place in its block is two 16-bit fields: PassParams::pix_table)
        return fail(ctx, MSK_ERR_INVALID_ARG, "block_size %d outside [%d, 4096]", p->block_size, block_min);
    const uint32_t bs = p->block_stride ? p->block_stride : 1;
    if (p->block_first >= bs) return fail(ctx, MSK_ERR_INVALID_ARG, "shard selector out of range");
    // MSK_RNG_PCG_BLOCK: a block's samples draw from ONE sequential PCG32 stream (samplers/independent.cpp:9-35), and how far a
    // sample advances it is only known once its path has been traced: a shard of the sample indices would start the stream at
    // the same place as every other shard and draw the same numbers (n shards summed = n copies of one spp / n render).  The
    // blocks' streams are independent: this mode shards by blocks (block_first / block_stride) only.
    if (p->rng_mode == MSK_RNG_PCG_BLOCK && (p->sample_first != 0 || (p->sample_stride != 0 && p->sample_stride != 1)))
        return fail(ctx, MSK_ERR_UNSUPPORTED, "MSK_RNG_PCG_BLOCK: the samples of a block share one sequential PCG32 stream and cannot be sharded "
                    "(sample_first %u, sample_stride %u); shard the blocks with block_first / block_stride", p->sample_first, p->sample_stride);
    return MSK_OK;
}

static uint32_t owned_spp(const msk_render_params *p) {
    const uint32_t ss = p->sample_stride ? p->sample_stride : 1;
    return p->sample_first < p->spp ? (p->spp - p->sample_first + ss - 1) / ss : 0;
}

// Short rays (LDS-resident scene): many small regions, one chunk loop per wave: 8192 x 512 = 4 M path slots (0.6 GB of state;
// measured 16384 / 12288 / 8192 / 6144 regions: 42.7 / 41.7 / 41.2 / 41.6 ms for the bench step — the shading kernel streams the
// whole pool's state every iteration and a smaller pool keeps more of it in the 256 MB Infinity Cache, the traversal kernel
// wants many waves per launch).  Long rays (k_trace_r): 4096 regions of 2048 slots = 8 M, so that lane replacement has a long
// list of rays to keep the lanes busy with.
static void pool_shape(const msk_scene *sc, uint64_t total_samples, uint32_t *region_size, uint32_t *n_regions) {
    const bool big = sc->trace_mode == 1 || sc->trace_mode == 2 || sc->trace_mode == 4 || sc->trace_mode == 5 || sc->trace_mode == 6;
    // trees in HBM: one traversal wave per region at 5 waves per SIMD = 5120 resident waves; with 4096 regions the four loops'
    // launches never filled the GPU (8192 regions: config-5-class render 173 vs 191 ms, config-3-class 205 vs 227 ms)
    // LDS-resident scenes: 6144 x 1024 (with the state's cache policy in place — msk_kernels.h, MSK_NT — fewer, longer regions
    // win over round 1's 8192 x 512: 36.2 vs 37.0 ms per bench step; 5120 … 8192 x 896 … 1280 are within 1 % of each other)
    uint32_t rs = env_u32("MSK_REGION_SIZE", big ? 2048 : 1024), nr = env_u32("MSK_REGIONS", big ? 8192 : 6144);
    rs = std::max(64u, (rs + 63u) & ~63u);
    while (rs > 256 && total_samples / rs < nr) rs = std::max(256u, rs / 2);       // small jobs: keep the GPU full first
    const uint64_t need = (total_samples + rs - 1) / rs;
    if (need < nr) nr = (uint32_t) std::max<uint64_t>(need, 1);
    nr = (nr + 3u) & ~3u;
    *region_size = rs; *n_regions = nr;
}


// MSK_RNG_PCG_BLOCK: the reference's sampler as written — one PCG32 stream per block, so one lane per block (msk_serial.h).
// Same block schedule, shards, crop window and Film::put as the wavefront path; a fidelity mode, not a fast one.
static int render_serial(msk_scene *sc, const msk_render_params *prm, float *d_film, hipStream_t stream, msk_stats *stats) {
    msk_ctx *ctx = sc->ctx;
    const int border = sc->dev.filter_border;
    EventPool ev{ctx, 0};
    hipEvent_t t_begin = ev.get(), t_end = ev.get();
    if (t_begin) (void) hipEventRecord(t_begin, stream);
    const int W = sc->dev.width, H = sc->dev.height, bs = prm->block_size;
    int nbx, nby;
    std::vector<HostBlock> all = spiral_blocks(W, H, bs, &nbx, &nby);
    const uint32_t bstride = prm->block_stride ? prm->block_stride : 1;
    std::vector<int32_t> block_of((size_t) nbx * nby, -1);
    std::vector<uint32_t> spiral_id((size_t) nbx * nby, 0);
    std::vector<BlockInfo> owned;
    for (size_t id = 0; id < all.size(); ++id) {
        spiral_id[(size_t) all[id].by * nbx + all[id].bx] = (uint32_t) id;
        if (id % bstride != prm->block_first || owned_spp(prm) == 0) continue;
        if (all[id].off_x - border >= sc->dev.crop_x + sc->dev.crop_w || all[id].off_x + all[id].size_x + border <= sc->dev.crop_x ||
            all[id].off_y - border >= sc->dev.crop_y + sc->dev.crop_h || all[id].off_y + all[id].size_y + border <= sc->dev.crop_y) continue;
        block_of[(size_t) all[id].by * nbx + all[id].bx] = (int32_t) owned.size();
        owned.push_back(BlockInfo{all[id].off_x, all[id].off_y, all[id].size_x, all[id].size_y, 0u, (uint32_t) owned.size()});
    }
    const uint32_t buf_stride = (uint32_t) ((bs + 2 * border) * (bs + 2 * border)) * 5;
    if (!sc->ws) sc->ws = new Workspace();
    Workspace &ws = *sc->ws;
    ws.plan_key.clear();                                 // the block tables below replace whatever plan a wavefront render cached
    const size_t buf_bytes = std::max<size_t>(owned.size(), 1) * buf_stride * 4;
    HIP_TRY(ctx, ws.block_buf.reserve(buf_bytes));
    HIP_TRY(ctx, ws.blocks.upload(owned)); HIP_TRY(ctx, ws.block_of.upload(block_of)); HIP_TRY(ctx, ws.spiral.upload(spiral_id));
    HIP_TRY(ctx, hipMemsetAsync(ws.block_buf.p, 0, buf_bytes, stream));
    DevBuf counters, ovf;
    HIP_TRY(ctx, counters.reserve(32));
    HIP_TRY(ctx, hipMemsetAsync(counters.p, 0, 32, stream));
    // few blocks (BASELINE config 1 has 64): one block per WAVE — the GPU holds ~2000 of this kernel's waves at once, and a wave that
    // runs one scalar loop does not pay for the branches of 63 others (config 1: 3.0 s -> see profiles/r04_pcg_block_config1.txt)
    const bool per_wave = owned.size() <= env_u32("MSK_SERIAL_PER_WAVE_MAX", 16384);
    const uint32_t grid = (uint32_t) ((owned.size() * (per_wave ? MSK_WAVE : 1) + MSK_BLOCK - 1) / MSK_BLOCK);
    // the binary tree's stack: depth + 2 entries, the first sc->dev.stack_entries of them in LDS
    DeviceScene ds = sc->dev;
    const uint32_t need = (uint32_t) sc->bvh_depth + 2u;
    if (ds.stack_entries > need) ds.stack_entries = (need + 3u) & ~3u;
    const uint32_t ovf_words = need > ds.stack_entries ? need - ds.stack_entries : 0u;
    if (ovf_words && grid) HIP_TRY(ctx, ovf.reserve((size_t) ovf_words * grid * MSK_BLOCK * 4));
    if (grid) {
        SerialParams sp;
        sp.seed = prm->seed; sp.spp = prm->spp; sp.sample_first = prm->sample_first; sp.sample_stride = prm->sample_stride;
        sp.rr_depth = prm->rr_depth; sp.max_depth = prm->max_depth; sp.hide_emitters = prm->hide_emitters;
        sp.blocks = ws.blocks.as<BlockInfo>(); sp.n_blocks = (uint32_t) owned.size();
        sp.block_buf = ws.block_buf.as<float>(); sp.buf_stride = buf_stride;
        sp.stack_ovf = ovf.as<uint32_t>(); sp.counters = counters.as<unsigned long long>(); sp.per_wave = per_wave ? 1u : 0u;
        hipLaunchKernelGGL(k_path_serial, dim3(grid), dim3(MSK_BLOCK), (size_t) ds.stack_entries * MSK_BLOCK * 4, stream, ds, sp);
    }
    FilmOut fo;
    fo.film = d_film; fo.stride = 5;
    for (int c = 0; c < 5; ++c) fo.ch[c] = c;
    hipLaunchKernelGGL(k_film_put, dim3((uint32_t) (((size_t) sc->dev.crop_w * sc->dev.crop_h + MSK_BLOCK - 1) / MSK_BLOCK)), dim3(MSK_BLOCK), 0, stream,
                       sc->dev, ws.blocks.as<BlockInfo>(), ws.block_of.as<int32_t>(), ws.spiral.as<uint32_t>(), nbx, nby, bs,
                       ws.block_buf.as<float>(), buf_stride, fo);
    if (t_end) (void) hipEventRecord(t_end, stream);
    HIP_TRY(ctx, hipGetLastError());
    unsigned long long h[4] = {0, 0, 0, 0};
    HIP_TRY(ctx, hipMemcpyAsync(h, counters.p, 32, hipMemcpyDeviceToHost, stream));
    // (one kernel for the whole job: the wall limit is 30 x MSK_WATCHDOG_S, an hour at the default)
    if (int rcw = ctx_sync(ctx, stream, "the MSK_RNG_PCG_BLOCK render (k_path_serial + Film::put)", 30.0)) { counters.leak(); ovf.leak(); return rcw; }
    if (stats) {
        stats->samples = h[0]; stats->segments = h[1]; stats->shadow_rays = h[2]; stats->invalid_samples = h[3]; stats->passes = 1;
        if (t_begin && t_end) (void) hipEventElapsedTime(&stats->ms_total, t_begin, t_end);
    }
    return MSK_OK;
}

static int render_impl(msk_scene *sc, const msk_render_params *prm, float *d_film, hipStream_t user_stream, msk_stats *stats,
                       const AovPlan *aov = nullptr) {
    msk_ctx *ctx = sc->ctx;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const int border = sc->dev.filter_border;
    int rc = check_params(ctx, prm, std::max(4, 2 * border));
    if (rc) return rc;
    if (prm->rng_mode == MSK_RNG_PCG_BLOCK) {
        if (aov) return fail(ctx, MSK_ERR_UNSUPPORTED, "MSK_RNG_PCG_BLOCK renders the \"path\" integrator only (the \"aov\" integrator uses MSK_RNG_COUNTER)");
        if (stats) std::memset(stats, 0, sizeof *stats);
        return render_serial(sc, prm, d_film, user_stream ? user_stream : ctx->stream, stats);
    }
    hipStream_t stream = user_stream ? user_stream : ctx->stream;
    if (stats) std::memset(stats, 0, sizeof *stats);
    EventPool ev{ctx, 0};
    hipEvent_t t_begin = ev.get(), t_end = ev.get();
    if (t_begin) (void) hipEventRecord(t_begin, stream);
    const int W = sc->dev.width, H = sc->dev.height, bs = prm->block_size;
    int nbx, nby;
    std::vector<HostBlock> all = spiral_blocks(W, H, bs, &nbx, &nby);
    const uint32_t bstride = prm->block_stride ? prm->block_stride : 1;
    const uint32_t spp_owned = owned_spp(prm);
    std::vector<int32_t> block_of((size_t) nbx * nby, -1);
    std::vector<uint32_t> spiral_id((size_t) nbx * nby, 0);
    std::vector<BlockInfo> owned;
    std::vector<size_t> owned_all_index;
    for (size_t id = 0; id < all.size(); ++id) {
        spiral_id[(size_t) all[id].by * nbx + all[id].bx] = (uint32_t) id;
        if (id % bstride != prm->block_first || spp_owned == 0) continue;
        // a block whose bordered area misses the crop window adds nothing to the film (accumulate_2d clips it to nothing,
        // imageblock.cpp:133-150): it is not rendered
        if (all[id].off_x - border >= sc->dev.crop_x + sc->dev.crop_w || all[id].off_x + all[id].size_x + border <= sc->dev.crop_x ||
            all[id].off_y - border >= sc->dev.crop_y + sc->dev.crop_h || all[id].off_y + all[id].size_y + border <= sc->dev.crop_y) continue;
        block_of[(size_t) all[id].by * nbx + all[id].bx] = (int32_t) owned.size();
        owned.push_back(BlockInfo{all[id].off_x, all[id].off_y, all[id].size_x, all[id].size_y, 0u, (uint32_t) owned.size()});
        owned_all_index.push_back(id);
    }
    const uint32_t per_block = (uint32_t) ((bs + 2 * border) * (bs + 2 * border));
    const uint32_t buf_stride = per_block * 5;
    if (!sc->ws) sc->ws = new Workspace();
    Workspace &ws = *sc->ws;
    DevBuf &d_block_buf = ws.block_buf, &d_blocks = ws.blocks, &d_block_of = ws.block_of, &d_spiral = ws.spiral;
    HIP_TRY(ctx, d_block_buf.reserve(std::max<size_t>(owned.size(), 1) * buf_stride * 4));
    const uint32_t n_aov_bufs = aov ? aov->n_groups + (aov->rgba ? 1u : 0u) : 0u;
    for (uint32_t g = 0; g < n_aov_bufs; ++g) {
        const uint32_t slot = g < aov->n_groups ? g : MSK_MAX_AOV_GROUPS;
        HIP_TRY(ctx, ws.aov_block_buf[slot].reserve(std::max<size_t>(owned.size(), 1) * buf_stride * 4));
    }
    const size_t rec_bytes = 20 + 16 * (size_t) n_aov_bufs;      // per sample
    // ---- plan passes: consecutive owned blocks whose records fit the budget
    size_t free_b = 0, total_b = 0;
    HIP_TRY(ctx, hipMemGetInfo(&free_b, &total_b));
    uint32_t region_size, n_regions;
    uint64_t all_samples = 0;
    for (auto &b : owned) all_samples += (uint64_t) b.size_x * b.size_y * spp_owned;
    pool_shape(sc, all_samples, &region_size, &n_regions);
    const size_t n_slots = (size_t) region_size * n_regions;
    const size_t state_bytes = 2 * n_slots * 144 + 4096;                              // two halves per region (StateBufs::alloc)
    const size_t held = ws.rec_a.bytes + ws.rec_b.bytes + ws.sb.id.bytes * 144 / 8;     // reusable: counts as free
    free_b += held;
    free_b /= std::max(1u, ctx->device_sharers);
    size_t budget = getenv("MSK_RECORD_BUDGET_MB") ? (size_t) atoll(getenv("MSK_RECORD_BUDGET_MB")) << 20
                                                   : (free_b > state_bytes ? (size_t) ((free_b - state_bytes) * 0.8) : 0);
    const size_t rec_bytes_per_block_max = (size_t) bs * bs * spp_owned * rec_bytes;
    if (!owned.empty() && budget < rec_bytes_per_block_max)
        return fail(ctx, MSK_ERR_OOM, "not enough HBM for one block of sample records (%zu B needed, %zu B budget)",
                    rec_bytes_per_block_max, budget);
    std::vector<std::pair<size_t, size_t>> passes;   // [b0, b1) over `owned`
    for (size_t b0 = 0; b0 < owned.size();) {
        size_t b1 = b0, bytes = 0; uint32_t pixel_base = 0;
        while (b1 < owned.size()) {
            const size_t add = (size_t) owned[b1].size_x * owned[b1].size_y * spp_owned * rec_bytes;
            if (b1 > b0 && bytes + add > budget) break;
            owned[b1].pixel_base = pixel_base; pixel_base += (uint32_t) (owned[b1].size_x * owned[b1].size_y);
            bytes += add; ++b1;
        }
        passes.push_back({b0, b1}); b0 = b1;
    }
    // Default filter (border 2, footprint of five pixels) and blocks whose bordered width fits 12 lane columns: the records carry
    // their filter weights and the replay is k_resolve_rows.  Anything else: positions + k_resolve_blocks.
    // source rows per band of k_resolve_rows: 8 cuts a 32-pixel block's 36 target rows into 8 + 7 x 4, eight equal chains per block
    // (and 2048 waves for the 512^2 bench = two per SIMD, what the kernel's registers allow): 3.03 vs 3.35 ms with 10
    const uint32_t band_rounds = std::max(5u, env_u32("MSK_RESOLVE_ROUNDS", 8));
    const bool packed = border == 2 && (int) std::floor(sc->dev.filter_radius + 0.5f) == 2 && bs + 2 * border <= 3 * MSK_RR_COLS &&
                        !env_u32("MSK_RESOLVE_GENERIC", 0);
    const std::vector<uint64_t> plan_key = {(uint64_t) W, (uint64_t) H, (uint64_t) bs, (uint64_t) border, prm->block_first, bstride,
                                            spp_owned, passes.size(), owned.size(), (uint64_t) packed, band_rounds};
    const bool plan_cached = passes.size() == 1 && ws.plan_key == plan_key;
    if (!plan_cached) {
        ws.plan_key.clear();
        HIP_TRY(ctx, d_blocks.upload(owned)); HIP_TRY(ctx, d_block_of.upload(block_of)); HIP_TRY(ctx, d_spiral.upload(spiral_id));
    }
    StateBufs &sb = ws.sb;
    if (!owned.empty()) HIP_TRY(ctx, sb.alloc(n_slots, n_regions));
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_trace, ev_shade, ev_resolve;
    DevBuf &d_pix = ws.pix, &d_rec_a = ws.rec_a, &d_rec_b = ws.rec_b;
    for (auto &ps : passes) {
        uint64_t n_pix = ws.plan_n_pix;
        if (!plan_cached) {
            std::vector<uint4> pix;
            for (size_t b = ps.first; b < ps.second; ++b) {
                for (int y = 0; y < owned[b].size_y; ++y)
                    for (int x = 0; x < owned[b].size_x; ++x)       // PassParams::pix_table; the record of (pixel j, sample) is j * spp + sample
                        pix.push_back(make_uint4((uint32_t) ((owned[b].off_y + y) * W + owned[b].off_x + x), (uint32_t) x | ((uint32_t) y << 16),
                                                 (uint32_t) (owned[b].off_x - border), (uint32_t) (owned[b].off_y - border)));
            }
            HIP_TRY(ctx, d_pix.upload(pix));
            std::vector<uint32_t> inv((size_t) W * H, 0xffffffffu);       // film index -> pass pixel (PassParams::pix_to_j)
            for (size_t j = 0; j < pix.size(); ++j) inv[pix[j].x] = (uint32_t) j;
            HIP_TRY(ctx, ws.pix_inv.upload(inv));
            n_pix = pix.size();
            // k_resolve_rows work list: per block, bands of target rows that each cost at most `band_rounds` source rows
            // (the first band has no rows above it to wait for and the last one none below, so they take more target rows)
            std::vector<RowBand> bands;
            for (size_t b = ps.first; packed && b < ps.second; ++b) {
                const uint32_t rows = (uint32_t) owned[b].size_y + 4u;
                uint32_t a = 0;
                while (a < rows) {
                    const uint32_t rem = rows - a;
                    const uint32_t h = (a == 0) ? std::min(band_rounds, rem) : (rem <= band_rounds ? rem : band_rounds - 4u);
                    bands.push_back(RowBand{(uint32_t) (b - ps.first), a, a + h, 0u});
                    a += h;
                }
            }
            HIP_TRY(ctx, ws.bands.upload(bands));
            ws.n_bands = (uint32_t) bands.size();
            if (passes.size() == 1) { ws.plan_key = plan_key; ws.plan_n_pix = n_pix; }
        }
        const uint64_t n_rec = n_pix * spp_owned;
        HIP_TRY(ctx, d_rec_a.reserve(n_rec * 16)); HIP_TRY(ctx, d_rec_b.reserve(n_rec * 4));
        AovParams ap;
        std::memset(&ap, 0, sizeof ap);
        float4 *aov_rgb = nullptr;
        if (aov) {
            ap.n_groups = aov->n_groups;
            for (uint32_t g = 0; g < aov->n_groups; ++g) {
                HIP_TRY(ctx, ws.aov_rec[g].reserve(n_rec * 16));
                ap.rec[g] = ws.aov_rec[g].as<float4>(); ap.code[g] = aov->code[g];
            }
            if (aov->rgba) { HIP_TRY(ctx, ws.aov_rec[MSK_MAX_AOV_GROUPS].reserve(n_rec * 16)); aov_rgb = ws.aov_rec[MSK_MAX_AOV_GROUPS].as<float4>(); }
        }
        rc = run_wavefront(sc, stream, prm, spp_owned, d_pix.as<uint4>(), ws.pix_inv.as<uint32_t>(), n_pix, d_rec_a.as<float4>(),
                           d_rec_b.as<float>(), sb, region_size, n_regions, stats, ev, ev_trace, ev_shade, aov ? &ap : nullptr, aov_rgb, packed);
        if (rc) return rc;
        const uint32_t nb = (uint32_t) (ps.second - ps.first);
        // tile of film pixels per thread: 1 wide (adjacent lanes read adjacent records -> full cache lines),
        // TY tall (a record is shared by TY pixels: 5*(TY+4)/TY reads per pixel)
        const int tile_x = (int) env_u32("MSK_RESOLVE_TX", 1), tile_y = (int) env_u32("MSK_RESOLVE_TY", 3);
        const int tiles_x = (bs + 2 * border + tile_x - 1) / tile_x, tiles_y = (bs + 2 * border + tile_y - 1) / tile_y;
        const uint64_t threads = (uint64_t) nb * tiles_x * tiles_y;
        hipEvent_t a = ev.get(), b = ev.get();
        if (a) (void) hipEventRecord(a, stream);
        const dim3 rgrid((uint32_t) ((threads + MSK_BLOCK - 1) / MSK_BLOCK));
#define MSK_RESOLVE(TX, TY) hipLaunchKernelGGL((k_resolve_blocks<TX, TY>), rgrid, dim3(MSK_BLOCK), 0, stream, sc->dev,      \
                           d_blocks.as<BlockInfo>() + ps.first, nb, res_rec, d_rec_b.as<float>(), spp_owned,               \
                           res_buf, buf_stride, tiles_x, tiles_y)
        // the XYZAW records, then every AOV record group through the same ordered replay (same weights, same order)
        for (uint32_t g = 0; g <= n_aov_bufs; ++g) {
            const uint32_t slot = g == 0 ? 0 : (g - 1 < (aov ? aov->n_groups : 0u) ? g - 1 : MSK_MAX_AOV_GROUPS);
            const float4 *res_rec = g == 0 ? d_rec_a.as<float4>() : ws.aov_rec[slot].as<float4>();
            float *res_buf = g == 0 ? d_block_buf.as<float>() : ws.aov_block_buf[slot].as<float>();
            if (packed) {
                if (ws.n_bands)
                    hipLaunchKernelGGL(k_resolve_rows, dim3(ws.n_bands), dim3(MSK_WAVE), (size_t) env_u32("MSK_RESOLVE_PAD_LDS_KB", 0) * 1024, stream, sc->dev, d_blocks.as<BlockInfo>() + ps.first,
                                       ws.bands.as<RowBand>(), ws.n_bands, res_rec, (const uint32_t *) d_rec_b.as<float>(), spp_owned, res_buf, buf_stride);
            }
            else if (tile_x == 2 && tile_y == 2) MSK_RESOLVE(2, 2);
            else if (tile_x == 2 && tile_y == 3) MSK_RESOLVE(2, 3);
            else if (tile_x == 2 && tile_y == 4) MSK_RESOLVE(2, 4);
            else if (tile_x == 1 && tile_y == 1) MSK_RESOLVE(1, 1);
            else if (tile_x == 1 && tile_y == 2) MSK_RESOLVE(1, 2);
            else if (tile_x == 1 && tile_y == 3) MSK_RESOLVE(1, 3);
            else if (tile_x == 1 && tile_y == 6) MSK_RESOLVE(1, 6);
            else MSK_RESOLVE(1, 4);
        }
#undef MSK_RESOLVE
        if (b) (void) hipEventRecord(b, stream);
        ev_resolve.push_back({a, b});
        if ((rc = ctx_sync(ctx, stream, "the film replay of a pass"))) return rc;   // d_pix / records are reused by the next pass
        if (stats) stats->passes++;
    }
    {
        hipEvent_t a = ev.get(), b = ev.get();
        if (a) (void) hipEventRecord(a, stream);
        for (uint32_t g = 0; g <= n_aov_bufs; ++g) {
            const uint32_t slot = g == 0 ? 0 : (g - 1 < (aov ? aov->n_groups : 0u) ? g - 1 : MSK_MAX_AOV_GROUPS);
            FilmOut fo;
            fo.film = d_film; fo.stride = 5 + (int32_t) (aov ? aov->n_channels : 0u);
            for (int c = 0; c < 5; ++c) fo.ch[c] = g == 0 ? c : aov->out_ch[g - 1 < aov->n_groups ? g - 1 : MSK_MAX_AOV_GROUPS][c];
            hipLaunchKernelGGL(k_film_put, dim3((uint32_t) (((size_t) sc->dev.crop_w * sc->dev.crop_h + MSK_BLOCK - 1) / MSK_BLOCK)), dim3(MSK_BLOCK), 0, stream,
                               sc->dev, d_blocks.as<BlockInfo>(), d_block_of.as<int32_t>(), d_spiral.as<uint32_t>(), nbx, nby, bs,
                               g == 0 ? d_block_buf.as<float>() : ws.aov_block_buf[slot].as<float>(), buf_stride, fo);
        }
        if (b) (void) hipEventRecord(b, stream);
        ev_resolve.push_back({a, b});
    }
    if (t_end) (void) hipEventRecord(t_end, stream);
    HIP_TRY(ctx, hipGetLastError());
    if ((rc = ctx_sync(ctx, stream, "Film::put"))) return rc;
    if (stats) {
        if (t_begin && t_end) (void) hipEventElapsedTime(&stats->ms_total, t_begin, t_end);
        sum_events(ev_resolve, &stats->ms_resolve);      // trace / shade were summed group by group in run_wavefront
    }
    return MSK_OK;
}

// The film's copy-back (msk_gpu_render, and a group's summed film).  A caller's array that is PINNED host memory (hipHostMalloc /
// hipHostRegister, a torch pinned tensor) takes the DMA directly: 5 MB in ~0.2 ms.  A pageable one goes through a pinned staging
// buffer kept with the scene (*stage): device -> pinned at link speed, then one host memcpy into the caller's array (~0.4 ms for
// the 5 MB bench film; a pageable hipMemcpy took 2-3 ms).  MSK_COPYBACK_STAGED=1 (tests): always through the staging buffer.
static int film_to_host(msk_ctx *ctx, hipStream_t stream, const void *d_film, float *h_film, size_t bytes, void **stage, size_t *stage_bytes) {
    hipPointerAttribute_t attr;
    const hipError_t ea = hipPointerGetAttributes(&attr, h_film);
    if (ea != hipSuccess) (void) hipGetLastError();                      // (an ordinary malloc'ed pointer is "invalid value" to older runtimes)
    if (ea == hipSuccess && attr.type == hipMemoryTypeHost && !env_u32("MSK_COPYBACK_STAGED", 0)) {
        HIP_TRY(ctx, hipMemcpyAsync(h_film, d_film, bytes, hipMemcpyDeviceToHost, stream));
        return ctx_sync(ctx, stream, "the film's copy-back");
    }
    if (*stage_bytes < bytes) {
        if (*stage) (void) hipHostFree(*stage);
        *stage = nullptr; *stage_bytes = 0;
        if (hipHostMalloc(stage, bytes, hipHostMallocDefault) == hipSuccess) *stage_bytes = bytes;
        else { (void) hipGetLastError(); *stage = nullptr; }
    }
    if (*stage) {
        HIP_TRY(ctx, hipMemcpyAsync(*stage, d_film, bytes, hipMemcpyDeviceToHost, stream));
        if (const int rc = ctx_sync(ctx, stream, "the film's copy-back")) return rc;
        std::memcpy(h_film, *stage, bytes);
    } else {
        HIP_TRY(ctx, hipMemcpy(h_film, d_film, bytes, hipMemcpyDeviceToHost));
    }
    return MSK_OK;
}

extern "C" int msk_gpu_render_device(msk_scene *scene, const msk_render_params *params, float *d_film_xyzaw, void *hip_stream,
                                     msk_stats *stats) {
    if (!scene || !d_film_xyzaw) return fail(scene ? scene->ctx : nullptr, MSK_ERR_INVALID_ARG, "msk_gpu_render_device: NULL argument");
    MSK_REFUSE_LOST(scene->ctx);
    if (scene->ctx->group) return group_render_device(scene, params, d_film_xyzaw, (hipStream_t) hip_stream, stats);
    return render_impl(scene, params, d_film_xyzaw, (hipStream_t) hip_stream, stats);
}

extern "C" int msk_gpu_render(msk_scene *scene, const msk_render_params *params, float *film_xyzaw, msk_stats *stats) {
    if (!scene || !film_xyzaw) return fail(scene ? scene->ctx : nullptr, MSK_ERR_INVALID_ARG, "msk_gpu_render: NULL argument");
    MSK_REFUSE_LOST(scene->ctx);
    if (scene->ctx->group) return group_render(scene, params, film_xyzaw, stats);
    msk_ctx *ctx = scene->ctx;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (!scene->ws) scene->ws = new Workspace();
    DevBuf &film = scene->ws->film;
    const size_t bytes = (size_t) scene->dev.crop_w * scene->dev.crop_h * 5 * 4;
    HIP_TRY(ctx, film.reserve(bytes));
    int rc = render_impl(scene, params, film.as<float>(), nullptr, stats);
    if (rc) return rc;
    return film_to_host(ctx, ctx->stream, film.p, film_xyzaw, bytes, &scene->ws->host_film, &scene->ws->host_film_bytes);
}

static const int kAovWidth[6] = {1, 3, 2, 3, 3, 4};
extern "C" uint32_t msk_gpu_aov_channels(const int32_t *aov_types, uint32_t n_aovs) {
    uint32_t n = 0;
    for (uint32_t i = 0; i < n_aovs; ++i) {
        if (!aov_types || aov_types[i] < 0 || aov_types[i] > MSK_AOV_PATH_RGBA) return 0;
        n += (uint32_t) kAovWidth[aov_types[i]];
    }
    return n;
}

extern "C" int msk_gpu_render_aov(msk_scene *scene, const msk_render_params *params, const int32_t *aov_types, uint32_t n_aovs,
                                  float *film, msk_stats *stats) {
    if (!scene || !film || !params || (n_aovs && !aov_types))
        return fail(scene ? scene->ctx : nullptr, MSK_ERR_INVALID_ARG, "msk_gpu_render_aov: NULL argument");
    MSK_REFUSE_LOST(scene->ctx);
    if (scene->ctx->group) return group_render_aov(scene, params, aov_types, n_aovs, film, stats);
    msk_ctx *ctx = scene->ctx;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    AovPlan plan;
    for (auto &row : plan.out_ch) for (int &c : row) c = -1;
    // selectors of the primary-hit channels, in film order (AovParams::code)
    static const int first_code[5] = {1, 2, 5, 7, 10};
    std::vector<std::pair<int, int>> prim;        // (selector, film channel)
    int ch = 5;
    for (uint32_t i = 0; i < n_aovs; ++i) {
        const int t = aov_types[i];
        if (t < 0 || t > MSK_AOV_PATH_RGBA) return fail(ctx, MSK_ERR_INVALID_ARG, "Invalid AOV type %d!", t);
        if (t == MSK_AOV_PATH_RGBA) {
            if (plan.rgba) return fail(ctx, MSK_ERR_UNSUPPORTED, "at most one nested integrator is supported by the \"aov\" integrator of this back end");
            plan.rgba = true;
            for (int c = 0; c < 4; ++c) plan.out_ch[MSK_MAX_AOV_GROUPS][c] = ch + c;     // block channel 3 = the weight sum = A
        } else {
            for (int c = 0; c < kAovWidth[t]; ++c) prim.push_back({first_code[t] + c, ch + c});
        }
        ch += kAovWidth[t];
    }
    plan.n_channels = (uint32_t) ch - 5;
    if (prim.size() > 3 * MSK_MAX_AOV_GROUPS)
        return fail(ctx, MSK_ERR_UNSUPPORTED, "too many AOV channels (%zu, at most %d besides the nested integrator)", prim.size(), 3 * MSK_MAX_AOV_GROUPS);
    plan.n_groups = (uint32_t) ((prim.size() + 2) / 3);
    for (size_t k = 0; k < prim.size(); ++k) {
        plan.code[k / 3] |= (uint32_t) prim[k].first << (8 * (k % 3));
        plan.out_ch[k / 3][k % 3] = prim[k].second;
    }
    msk_render_params p = *params;
    if (!plan.rgba) p.max_depth = 0;      // no nested integrator: only the camera ray is traced and XYZ stays 0 (aov.cpp:91)
    if (!scene->ws) scene->ws = new Workspace();
    DevBuf &d_film = scene->ws->film;
    const size_t bytes = (size_t) scene->dev.crop_w * scene->dev.crop_h * (5 + plan.n_channels) * 4;
    HIP_TRY(ctx, d_film.reserve(bytes));
    HIP_TRY(ctx, hipMemsetAsync(d_film.p, 0, bytes, ctx->stream));
    int rc = render_impl(scene, &p, d_film.as<float>(), nullptr, stats, &plan);
    if (rc) return rc;
    HIP_TRY(ctx, hipMemcpy(film, d_film.p, bytes, hipMemcpyDeviceToHost));
    return MSK_OK;
}

extern "C" int msk_gpu_sample_pixels(msk_scene *scene, const msk_render_params *prm, uint64_t n_pixels, const int32_t *pixels,
                                     float *out_xyz, float *out_pos) {
    if (!scene || !pixels || !out_xyz) return fail(scene ? scene->ctx : nullptr, MSK_ERR_INVALID_ARG, "msk_gpu_sample_pixels: NULL argument");
    MSK_REFUSE_LOST(scene->ctx);
    if (scene->ctx->group) {             // sub-stage entry points of a group run on its first member
        const int rc = msk_gpu_sample_pixels(scene->parts[0], prm, n_pixels, pixels, out_xyz, out_pos);
        return rc ? group_fail(scene->ctx, 0, rc) : MSK_OK;
    }
    msk_ctx *ctx = scene->ctx;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    int rc = check_params(ctx, prm, 1);
    if (rc) return rc;
    if (prm->rng_mode != MSK_RNG_COUNTER)
        return fail(ctx, MSK_ERR_UNSUPPORTED, "msk_gpu_sample_pixels: rng_mode %d (a pixel's samples are only addressable with the counter RNG: "
                    "a per-block PCG32 stream is sequential by construction)", prm->rng_mode);
    const int W = scene->dev.width, H = scene->dev.height;
    if (n_pixels >> 32) return fail(ctx, MSK_ERR_INVALID_ARG, "too many pixels");
    // The path state names a sample by its FILM pixel (PathState::id), so a pixel listed twice is rendered once and its
    // samples are copied to every place it is listed at.
    std::vector<uint4> pix;
    std::vector<uint32_t> inv((size_t) W * H, 0xffffffffu), place(n_pixels);
    for (uint64_t i = 0; i < n_pixels; ++i) {
        const int x = pixels[2 * i], y = pixels[2 * i + 1];
        if (x < 0 || y < 0 || x >= W || y >= H) return fail(ctx, MSK_ERR_INVALID_ARG, "pixel (%d,%d) outside the %dx%d film", x, y, W, H);
        const uint32_t f = (uint32_t) (y * W + x);
        if (inv[f] == 0xffffffffu) {
            inv[f] = (uint32_t) pix.size();
            // a "block" of one pixel at (x, y): the kernels take the film coordinates from the block offset (pixel_x / pixel_y)
            pix.push_back(make_uint4(f, 0u, (uint32_t) (x - scene->dev.filter_border), (uint32_t) (y - scene->dev.filter_border)));
        }
        place[i] = inv[f];
    }
    if (n_pixels == 0) return MSK_OK;
    const uint64_t n_listed = n_pixels;
    n_pixels = pix.size();
    msk_render_params p = *prm; p.sample_first = 0; p.sample_stride = 1;
    const uint64_t n_rec = n_pixels * p.spp;
    uint32_t region_size, n_regions;
    pool_shape(scene, n_rec, &region_size, &n_regions);
    StateBufs sb; DevBuf d_pix, d_inv, ra, rb, ox, op;
    HIP_TRY(ctx, sb.alloc((size_t) region_size * n_regions, n_regions));
    HIP_TRY(ctx, d_pix.upload(pix)); HIP_TRY(ctx, d_inv.upload(inv)); HIP_TRY(ctx, ra.alloc(n_rec * 16)); HIP_TRY(ctx, rb.alloc(n_rec * 4));
    HIP_TRY(ctx, ox.alloc(n_rec * 12)); HIP_TRY(ctx, op.alloc(n_rec * 8));
    EventPool ev{ctx, 0};
    std::vector<std::pair<hipEvent_t, hipEvent_t>> e1, e2;
    // a lost context (the watchdog gave up): the local buffers are dropped without hipFree, which would wait for the device
    auto sb_leak = [&]() { if (ctx->lost) { sb.leak(); for (DevBuf *b : {&d_pix, &d_inv, &ra, &rb, &ox, &op}) b->leak(); } };
    rc = run_wavefront(scene, ctx->stream, &p, p.spp, d_pix.as<uint4>(), d_inv.as<uint32_t>(), n_pixels, ra.as<float4>(), rb.as<float>(), sb,
                       region_size, n_regions, nullptr, ev, e1, e2);
    if (rc) { sb_leak(); return rc; }
    hipLaunchKernelGGL(k_export_records, dim3((uint32_t) ((n_rec + 255) / 256)), dim3(256), 0, ctx->stream, ra.as<float4>(),
                       rb.as<float>(), n_pixels, p.spp, ox.as<float>(), op.as<float>());
    if ((rc = ctx_sync(ctx, ctx->stream, "k_export_records"))) { sb_leak(); return rc; }
    std::vector<float> hx((size_t) n_rec * 3), hp(out_pos ? (size_t) n_rec * 2 : 0);
    HIP_TRY(ctx, hipMemcpy(hx.data(), ox.p, n_rec * 12, hipMemcpyDeviceToHost));
    if (out_pos) HIP_TRY(ctx, hipMemcpy(hp.data(), op.p, n_rec * 8, hipMemcpyDeviceToHost));
    for (uint64_t i = 0; i < n_listed; ++i) {
        std::memcpy(out_xyz + i * p.spp * 3, hx.data() + (size_t) place[i] * p.spp * 3, (size_t) p.spp * 12);
        if (out_pos) std::memcpy(out_pos + i * p.spp * 2, hp.data() + (size_t) place[i] * p.spp * 2, (size_t) p.spp * 8);
    }
    return MSK_OK;
}

static int trace_batch(msk_scene *scene, uint64_t n, const float *rays, float *out_hit, uint8_t *out_any) {
    msk_ctx *ctx = scene->ctx;
    MSK_REFUSE_LOST(ctx);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (n == 0) return MSK_OK;
    DevBuf d_rays, d_out;
    HIP_TRY(ctx, d_rays.alloc(n * 32)); HIP_TRY(ctx, d_out.alloc(out_any ? n : n * 16));
    HIP_TRY(ctx, hipMemcpy(d_rays.p, rays, n * 32, hipMemcpyHostToDevice));
    const uint32_t grid = (uint32_t) std::min<uint64_t>((n + MSK_BLOCK - 1) / MSK_BLOCK, 4096);
    float4 *oh = out_any ? nullptr : d_out.as<float4>();
    uint8_t *oa = out_any ? d_out.as<uint8_t>() : nullptr;
    DevBuf d_ovf;
    if (scene->dev.stack_total > scene->dev.stack_entries)
        HIP_TRY(ctx, d_ovf.alloc((size_t) (scene->dev.stack_total - scene->dev.stack_entries) * grid * MSK_BLOCK * 4));
    if (scene->trace_mode == 0)
        hipLaunchKernelGGL(k_trace_batch<0>, dim3(grid), dim3(MSK_BLOCK), scene->trace_lds_bytes, ctx->stream, scene->dev,
                           d_rays.as<float4>(), n, oh, oa, d_ovf.as<uint32_t>());
    else if (scene->trace_mode == 1)
        hipLaunchKernelGGL(k_trace_batch<1>, dim3(grid), dim3(MSK_BLOCK), scene->trace_lds_bytes, ctx->stream, scene->dev,
                           d_rays.as<float4>(), n, oh, oa, d_ovf.as<uint32_t>());
    else if (scene->trace_mode == 2)
        hipLaunchKernelGGL(k_trace_batch<2>, dim3(grid), dim3(MSK_BLOCK), scene->trace_lds_bytes, ctx->stream, scene->dev,
                           d_rays.as<float4>(), n, oh, oa, d_ovf.as<uint32_t>());
    else if (scene->trace_mode == 4)
        hipLaunchKernelGGL(k_trace_batch<4>, dim3(grid), dim3(MSK_BLOCK), scene->trace_lds_bytes, ctx->stream, scene->dev,
                           d_rays.as<float4>(), n, oh, oa, d_ovf.as<uint32_t>());
    else if (scene->trace_mode == 5)
        hipLaunchKernelGGL(k_trace_batch<5>, dim3(grid), dim3(MSK_BLOCK), scene->trace_lds_bytes, ctx->stream, scene->dev,
                           d_rays.as<float4>(), n, oh, oa, d_ovf.as<uint32_t>());
    else if (scene->trace_mode == 6)
        hipLaunchKernelGGL(k_trace_batch<6>, dim3(grid), dim3(MSK_BLOCK), scene->trace_lds_bytes, ctx->stream, scene->dev,
                           d_rays.as<float4>(), n, oh, oa, d_ovf.as<uint32_t>());
    else
        hipLaunchKernelGGL(k_trace_batch<3>, dim3(grid), dim3(MSK_BLOCK), scene->trace_lds_bytes, ctx->stream, scene->dev,
                           d_rays.as<float4>(), n, oh, oa, d_ovf.as<uint32_t>());
    HIP_TRY(ctx, hipGetLastError());
    if (int rcw = ctx_sync(ctx, ctx->stream, "k_trace_batch")) { d_rays.leak(); d_out.leak(); d_ovf.leak(); return rcw; }
    HIP_TRY(ctx, hipMemcpy(out_any ? (void *) out_any : (void *) out_hit, d_out.p, out_any ? n : n * 16, hipMemcpyDeviceToHost));
    return MSK_OK;
}

extern "C" int msk_gpu_trace_closest(msk_scene *scene, uint64_t n, const float *rays, float *out_hit) {
    if (!scene || (n && (!rays || !out_hit))) return fail(scene ? scene->ctx : nullptr, MSK_ERR_INVALID_ARG, "msk_gpu_trace_closest: NULL argument");
    if (scene->ctx->group) { const int rc = trace_batch(scene->parts[0], n, rays, out_hit, nullptr); return rc ? group_fail(scene->ctx, 0, rc) : MSK_OK; }
    return trace_batch(scene, n, rays, out_hit, nullptr);
}
extern "C" int msk_gpu_trace_any(msk_scene *scene, uint64_t n, const float *rays, uint8_t *out_occluded) {
    if (!scene || (n && (!rays || !out_occluded))) return fail(scene ? scene->ctx : nullptr, MSK_ERR_INVALID_ARG, "msk_gpu_trace_any: NULL argument");
    if (scene->ctx->group) { const int rc = trace_batch(scene->parts[0], n, rays, nullptr, out_occluded); return rc ? group_fail(scene->ctx, 0, rc) : MSK_OK; }
    return trace_batch(scene, n, rays, nullptr, out_occluded);
}
