// msk_multi.h — several devices behind one msk_ctx (included by msk_gpu.hip, which defines msk_ctx / msk_scene).
//
// msk_gpu_init(ids, n > 1) returns a GROUP context: one ordinary single-device context per entry of `ids` (an ordinal may
// repeat: two contexts on one GPU is how the path is rehearsed on a one-GPU box).  A scene created on a group is created on
// every member; a render is split by SAMPLE INDEX — member k renders the indices sample_first + (k + j n) sample_stride of the
// call (pixels x spp shard trivially, in counter RNG mode the union is exactly the single-device sample set) —
// one host thread per member, and the members' films {X,Y,Z,A,W} are summed onto the first device by k_film_sum, which reads
// the other devices' buffers directly over xGMI peer access (a staged hipMemcpyPeer where peer access cannot be enabled), in
// member order: ((f0 + f1) + f2) + ...  The result differs from a single-device film only by that re-association.
// This is the multi-GPU path for a caller that owns ONE process (the "path" plugin: gpu_devices="0,1,..."); the benchmark's
// one-process-per-GPU path (bench.py, RCCL reduce) sits above the C ABI and uses single-device contexts.
#pragma once

struct msk_group {
    std::vector<msk_ctx *> ctxs;            // members, in the order of `ids`
    std::vector<char> peer;                 // member k's memory is addressable from the first device
};

struct FilmSources { const float *src[MSK_MAX_GROUP]; uint32_t n; };
__global__ void __launch_bounds__(MSK_BLOCK) k_film_sum(float *dst, FilmSources s, size_t count) {
    for (size_t i = (size_t) blockIdx.x * MSK_BLOCK + threadIdx.x; i < count; i += (size_t) gridDim.x * MSK_BLOCK) {
        float v = s.src[0][i];
        for (uint32_t k = 1; k < s.n; ++k) v += s.src[k][i];
        dst[i] = v;
    }
}

static int group_init(const int *ids, int n, msk_ctx **out) {
    if (n > MSK_MAX_GROUP) return fail(nullptr, MSK_ERR_INVALID_ARG, "msk_gpu_init: at most %d devices per context (got %d)", MSK_MAX_GROUP, n);
    msk_ctx *g = new msk_ctx();
    g->group = new msk_group();
    for (int k = 0; k < n; ++k) {
        msk_ctx *c = nullptr;
        const int rc = msk_gpu_init(&ids[k], 1, &c);
        if (rc != MSK_OK) { msk_gpu_shutdown(g); return rc; }       // g_last_error holds the member's message
        g->group->ctxs.push_back(c);
    }
    for (int k = 0; k < n; ++k) {
        uint32_t same = 0;
        for (int j = 0; j < n; ++j) same += ids[j] == ids[k];
        g->group->ctxs[k]->device_sharers = same;
    }
    g->device = ids[0];
    g->prop = g->group->ctxs[0]->prop;
    g->group->peer.assign(n, 1);
    (void) hipSetDevice(ids[0]);
    // MSK_GROUP_FORCE_STAGED=1: every member's film but the first reaches the first device through hipMemcpyPeerAsync into a
    // staging buffer, as it does for a member whose memory cannot be mapped — the branch a node with full peer access (and a
    // one-GPU rehearsal, where every member is "the same device") never takes otherwise.  MSK_GROUP_LOG=1 says per member
    // which of the two it got (on stderr, once per msk_gpu_init).
    const bool force_staged = env_u32("MSK_GROUP_FORCE_STAGED", 0) != 0, log = env_u32("MSK_GROUP_LOG", 0) != 0;
    for (int k = 1; k < n; ++k) {
        int can = ids[k] == ids[0] ? 1 : 0;
        const char *why = ids[k] == ids[0] ? "same device as member 0" : "peer access enabled";
        if (ids[k] != ids[0]) {
            if (hipDeviceCanAccessPeer(&can, ids[0], ids[k]) != hipSuccess) can = 0;
            if (!can) why = "hipDeviceCanAccessPeer says no";
            else {
                const hipError_t e = hipDeviceEnablePeerAccess(ids[k], 0);
                if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) { can = 0; why = "hipDeviceEnablePeerAccess failed"; }
            }
            (void) hipGetLastError();
        }
        if (force_staged) { can = 0; why = "MSK_GROUP_FORCE_STAGED"; }
        g->group->peer[k] = (char) can;
        if (log) std::fprintf(stderr, "[msk_gpu] group member %d (device %d): film summed %s (%s)\n", k, ids[k],
                              can ? "in place over peer access" : "from a staged hipMemcpyPeerAsync copy on device 0", why);
    }
    *out = g;
    return MSK_OK;
}

// A member the watchdog gave up on takes its DEVICE out of the group's teardown: every member context on that ordinal is marked
// lost as well (their hipFree / hipStreamDestroy would wait for the same hung device), and so is the group.
static void group_spread_lost(msk_ctx *g) {
    for (msk_ctx *c : g->group->ctxs)
        if (c->lost) {
            g->lost = true;
            for (msk_ctx *o : g->group->ctxs) if (o->device == c->device) o->lost = true;
        }
}

static void group_shutdown(msk_ctx *g) {
    group_spread_lost(g);
    for (msk_ctx *c : g->group->ctxs) msk_gpu_shutdown(c);
    delete g->group;
    g->group = nullptr;
    delete g;
}

static int group_fail(msk_ctx *g, size_t k, int rc) {
    // the watchdog gave up on a member — the one whose error is reported or any other (the callers stop at the FIRST failing
    // member; a later one may be the lost one): the group is done as well, and its teardown frees nothing on a device
    group_spread_lost(g);
    return fail(g, rc, "device %d (member %zu of %zu): %s", g->group->ctxs[k]->device, k, g->group->ctxs.size(),
                g->group->ctxs[k]->last_error.c_str());
}

static int group_scene_create(msk_ctx *g, const msk_scene_desc *d, msk_scene **out) {
    const size_t n = g->group->ctxs.size();
    msk_scene *s = new msk_scene();
    s->ctx = g;
    s->parts.assign(n, nullptr);
    std::vector<int> rcs(n, MSK_OK);
    auto make = [&](size_t k) { rcs[k] = msk_gpu_scene_create(g->group->ctxs[k], d, &s->parts[k]); };
    std::vector<std::thread> th;
    for (size_t k = 1; k < n; ++k) th.emplace_back(make, k);          // the BVH build is host work: one thread per member
    make(0);
    for (auto &t : th) t.join();
    for (size_t k = 0; k < n; ++k)
        if (rcs[k] != MSK_OK) {
            const int rc = group_fail(g, k, rcs[k]);
            msk_gpu_scene_destroy(s);
            return rc;
        }
    s->dev.width = s->parts[0]->dev.width; s->dev.height = s->parts[0]->dev.height;
    s->dev.crop_w = s->parts[0]->dev.crop_w; s->dev.crop_h = s->parts[0]->dev.crop_h;
    *out = s;
    return MSK_OK;
}

static void group_scene_destroy(msk_scene *s) {
    group_spread_lost(s->ctx);
    for (msk_scene *p : s->parts) if (p) msk_gpu_scene_destroy(p);
    if (s->ctx->lost) return;                             // (as for a single context: device memory is not freed under a hung kernel)
    (void) hipSetDevice(s->ctx->device);
    if (s->group_host_film) (void) hipHostFree(s->group_host_film);
    delete s;
}

// member k's share of the call: of its sample indices (MSK_RNG_COUNTER: every sample has its own counter key) or of its blocks
// (MSK_RNG_PCG_BLOCK: a block's samples share one sequential stream, only the blocks' streams are independent — check_params
// refuses a sample shard of that mode); either composes with a shard the caller asked for
static msk_render_params group_shard(const msk_render_params &p, uint32_t k, uint32_t n) {
    msk_render_params q = p;
    if (p.rng_mode == MSK_RNG_PCG_BLOCK) {
        const uint32_t stride = p.block_stride ? p.block_stride : 1;
        q.block_first = p.block_first + k * stride;
        q.block_stride = stride * n;
        return q;
    }
    const uint32_t stride = p.sample_stride ? p.sample_stride : 1;
    q.sample_first = p.sample_first + k * stride;
    q.sample_stride = stride * n;
    return q;
}

static void group_merge_stats(msk_stats *out, const std::vector<msk_stats> &st, float ms_wall) {
    if (!out) return;
    std::memset(out, 0, sizeof *out);
    for (const msk_stats &s : st) {
        out->samples += s.samples; out->segments += s.segments; out->shadow_rays += s.shadow_rays; out->invalid_samples += s.invalid_samples;
        out->iterations = std::max(out->iterations, s.iterations); out->passes = std::max(out->passes, s.passes);
        out->ms_trace += s.ms_trace; out->ms_shade += s.ms_shade; out->ms_resolve = std::max(out->ms_resolve, s.ms_resolve);
        out->n_trace_launches += s.n_trace_launches; out->n_shade_launches += s.n_shade_launches;
        out->launches_trace += s.launches_trace; out->launches_shade += s.launches_shade; out->launches_wavefront += s.launches_wavefront;
        out->bytes_shade += s.bytes_shade; out->bytes_trace += s.bytes_trace;
    }
    out->ms_total = ms_wall;                 // host wall time of the whole call: the members' renders side by side + the film sum
}

// renders every member's shard into its own device film (channels floats per pixel), then sums them into d_dst (first device)
static int group_render_device(msk_scene *s, const msk_render_params *params, float *d_dst, hipStream_t stream, msk_stats *stats) {
    msk_ctx *g = s->ctx;
    if (!params) return fail(g, MSK_ERR_INVALID_ARG, "render params are NULL");
    const size_t n = s->parts.size();
    const size_t count = (size_t) s->dev.crop_w * s->dev.crop_h * 5, bytes = count * 4;
    const auto t0 = std::chrono::steady_clock::now();
    std::vector<int> rcs(n, MSK_OK);
    std::vector<msk_stats> st(n);
    auto run = [&](size_t k) {
        msk_ctx *c = g->group->ctxs[k];
        if (hipSetDevice(c->device) != hipSuccess || s->part_films[k].reserve(bytes) != hipSuccess) {
            rcs[k] = fail_to(&c->last_error, MSK_ERR_OOM, "film buffer of %zu bytes", bytes);
            return;
        }
        const msk_render_params q = group_shard(*params, (uint32_t) k, (uint32_t) n);
        rcs[k] = msk_gpu_render_device(s->parts[k], &q, s->part_films[k].as<float>(), nullptr, &st[k]);
    };
    std::vector<std::thread> th;
    for (size_t k = 1; k < n; ++k) th.emplace_back(run, k);
    run(0);
    for (auto &t : th) t.join();
    for (size_t k = 0; k < n; ++k) if (rcs[k] != MSK_OK) return group_fail(g, k, rcs[k]);
    // ---- the film sum on the first device (every member's render is complete: msk_gpu_render_device returns after its sync)
    HIP_TRY(g, hipSetDevice(g->device));
    hipStream_t q = stream ? stream : g->group->ctxs[0]->stream;
    FilmSources src;
    src.n = (uint32_t) n;
    for (size_t k = 0; k < n; ++k) {
        src.src[k] = s->part_films[k].as<float>();
        if (!g->group->peer[k]) {                // no peer mapping: copy the member's film over first
            HIP_TRY(g, s->staged[k].reserve(bytes));
            HIP_TRY(g, hipMemcpyPeerAsync(s->staged[k].p, g->device, s->part_films[k].p, g->group->ctxs[k]->device, bytes, q));
            src.src[k] = s->staged[k].as<float>();
        }
    }
    const uint32_t grid = (uint32_t) std::min<size_t>((count + MSK_BLOCK - 1) / MSK_BLOCK, 4096);
    hipLaunchKernelGGL(k_film_sum, dim3(grid), dim3(MSK_BLOCK), 0, q, d_dst, src, count);
    HIP_TRY(g, hipGetLastError());
    // (a timed wait like every other of a render: a first device that stopped answering loses its member context, and with it the group)
    if (const int rcw = ctx_sync(g->group->ctxs[0], q, "the group's film sum (k_film_sum)")) return group_fail(g, 0, rcw);
    group_merge_stats(stats, st, std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count());
    return MSK_OK;
}

static int group_render(msk_scene *s, const msk_render_params *params, float *film_xyzaw, msk_stats *stats) {
    msk_ctx *g = s->ctx;
    const size_t bytes = (size_t) s->dev.crop_w * s->dev.crop_h * 5 * 4;
    HIP_TRY(g, hipSetDevice(g->device));
    HIP_TRY(g, s->group_film.reserve(bytes));
    const int rc = group_render_device(s, params, s->group_film.as<float>(), nullptr, stats);
    if (rc) return rc;
    HIP_TRY(g, hipSetDevice(g->device));
    msk_ctx *c0 = g->group->ctxs[0];
    const int rcc = film_to_host(c0, c0->stream, s->group_film.p, film_xyzaw, bytes, &s->group_host_film, &s->group_host_film_bytes);
    return rcc ? group_fail(g, 0, rcc) : MSK_OK;
}

// The "aov" integrator is not the hot path: every member renders its shard to the host, the films are added there in member order.
static int group_render_aov(msk_scene *s, const msk_render_params *params, const int32_t *aov_types, uint32_t n_aovs, float *film, msk_stats *stats) {
    msk_ctx *g = s->ctx;
    const size_t n = s->parts.size();
    const uint32_t ch = 5 + msk_gpu_aov_channels(aov_types, n_aovs);
    const size_t count = (size_t) s->dev.crop_w * s->dev.crop_h * ch;
    const auto t0 = std::chrono::steady_clock::now();
    std::vector<std::vector<float>> films(n);
    std::vector<int> rcs(n, MSK_OK);
    std::vector<msk_stats> st(n);
    auto run = [&](size_t k) {
        films[k].assign(count, 0.f);
        const msk_render_params q = group_shard(*params, (uint32_t) k, (uint32_t) n);
        rcs[k] = msk_gpu_render_aov(s->parts[k], &q, aov_types, n_aovs, films[k].data(), &st[k]);
    };
    std::vector<std::thread> th;
    for (size_t k = 1; k < n; ++k) th.emplace_back(run, k);
    run(0);
    for (auto &t : th) t.join();
    for (size_t k = 0; k < n; ++k) if (rcs[k] != MSK_OK) return group_fail(g, k, rcs[k]);
    for (size_t i = 0; i < count; ++i) { float v = films[0][i]; for (size_t k = 1; k < n; ++k) v += films[k][i]; film[i] = v; }
    group_merge_stats(stats, st, std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count());
    return MSK_OK;
}
