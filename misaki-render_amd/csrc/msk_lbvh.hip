// msk_lbvh.hip — BVH construction on the device (MSK_BVH_BUILD=gpu): a linear BVH over Morton-sorted triangle centres
// (Karras 2012), for scenes that change between renders.  Replaces rtcCommitScene (reference: src/librender/scene.cpp:201-212)
// like the host's binned-SAH builder (msk_bvh.h) does, and writes the same records: 64-byte binary nodes with padded child
// boxes, leaf-ordered 64-byte triangles in Embree's precomputed form, their D10 bounds.  Hit selection is by (t, prim), so
// the tree decides how fast a ray finds its hit, never which hit it finds: films are bit-identical with either builder.
//
//   k_prims      per triangle: bounding box, 30-bit Morton code of its centre in the scene box -> key = code << 32 | index
//   k_rs_*       stable LSD radix sort of the keys on their Morton half, 8 bits a pass (the index half is ascending to begin with, so
//                this is the order of the whole 64-bit key): per 2048-key tile a digit histogram, one exclusive scan over
//                (digit, tile), and a scatter in which ONE WAVE walks its tile 64 keys at a time — a key's rank among the equal
//                digits of its 64 comes from eight ballots, the running position of every digit sits in LDS
//   k_hierarchy  per internal node: its key range and split from common-prefix lengths (one thread per node, no atomics)
//   k_refit      per leaf, bottom-up: the second thread to arrive at a node unions its children's boxes (one counter per node)
//   scan         exclusive scan of the "kept" flags (k_scan_tile / k_scan_add, three levels: 2048 per block): ranges of <= leaf_size
//                keys become leaves, the rest are renumbered densely
//   k_emit       per kept node: the 64-byte record;  k_tris  per sorted triangle: record + bounds
#include "msk_lbvh.h"
#include <algorithm>
#include <cmath>

namespace msklbvh {

#define LB_BLOCK 256
#define LB_LEAF_BIT 0x80000000u

struct Box6 { float lo[3], hi[3]; };

__device__ __forceinline__ uint32_t expand10(uint32_t v) {        // 10 bits -> every third bit
    v = (v * 0x00010001u) & 0xFF0000FFu;
    v = (v * 0x00000101u) & 0x0F00F00Fu;
    v = (v * 0x00000011u) & 0xC30C30C3u;
    v = (v * 0x00000005u) & 0x49249249u;
    return v;
}

__global__ void __launch_bounds__(LB_BLOCK)
k_prims(const float4 *tri_verts, uint32_t n, float3 lo, float3 inv_ext, Box6 *tri_box, unsigned long long *keys) {
    const uint32_t i = blockIdx.x * LB_BLOCK + threadIdx.x;
    if (i >= n) return;
    const float4 a = tri_verts[(size_t) i * 3], b = tri_verts[(size_t) i * 3 + 1], c = tri_verts[(size_t) i * 3 + 2];
    Box6 bx;
    bx.lo[0] = fminf(a.x, fminf(b.x, c.x)); bx.lo[1] = fminf(a.y, fminf(b.y, c.y)); bx.lo[2] = fminf(a.z, fminf(b.z, c.z));
    bx.hi[0] = fmaxf(a.x, fmaxf(b.x, c.x)); bx.hi[1] = fmaxf(a.y, fmaxf(b.y, c.y)); bx.hi[2] = fmaxf(a.z, fmaxf(b.z, c.z));
    tri_box[i] = bx;
    const float cx = (0.5f * (bx.lo[0] + bx.hi[0]) - lo.x) * inv_ext.x, cy = (0.5f * (bx.lo[1] + bx.hi[1]) - lo.y) * inv_ext.y,
                cz = (0.5f * (bx.lo[2] + bx.hi[2]) - lo.z) * inv_ext.z;
    const uint32_t qx = (uint32_t) fminf(fmaxf(cx * 1024.f, 0.f), 1023.f), qy = (uint32_t) fminf(fmaxf(cy * 1024.f, 0.f), 1023.f),
                   qz = (uint32_t) fminf(fmaxf(cz * 1024.f, 0.f), 1023.f);
    const uint32_t code = (expand10(qx) << 2) | (expand10(qy) << 1) | expand10(qz);
    keys[i] = ((unsigned long long) code << 32) | i;
}

struct Hier {                    // per internal node i (0 .. n-2)
    uint32_t *first, *last;      // key range covered
    uint32_t *left, *right;      // child: index | LB_LEAF_BIT for a single key (leaf index = position in sorted order)
    uint32_t *parent;            // of internal node i (root: 0xffffffff)
    uint32_t *leaf_parent;       // of sorted key k
};

__device__ __forceinline__ int delta(const unsigned long long *keys, int n, int i, int j) {
    if (j < 0 || j >= n) return -1;
    return __clzll((long long) (keys[i] ^ keys[j]));
}

__global__ void __launch_bounds__(LB_BLOCK)
k_hierarchy(const unsigned long long *keys, int n, Hier h) {
    const int i = (int) (blockIdx.x * LB_BLOCK + threadIdx.x);
    if (i >= n - 1) return;
    const int d = (delta(keys, n, i, i + 1) - delta(keys, n, i, i - 1)) >= 0 ? 1 : -1;
    const int dmin = delta(keys, n, i, i - d);
    int lmax = 2;
    while (delta(keys, n, i, i + lmax * d) > dmin) lmax *= 2;
    int l = 0;
    for (int t = lmax / 2; t >= 1; t /= 2)
        if (delta(keys, n, i, i + (l + t) * d) > dmin) l += t;
    const int j = i + l * d;
    const int dnode = delta(keys, n, i, j);
    int s = 0, t = l;
    do {
        t = (t + 1) / 2;
        if (delta(keys, n, i, i + (s + t) * d) > dnode) s += t;
    } while (t > 1);
    const int gamma = i + s * d + (d < 0 ? -1 : 0);
    const int lo = i < j ? i : j, hi = i < j ? j : i;
    h.first[i] = (uint32_t) lo; h.last[i] = (uint32_t) hi;
    if (lo == gamma) { h.left[i] = (uint32_t) gamma | LB_LEAF_BIT; h.leaf_parent[gamma] = (uint32_t) i; }
    else { h.left[i] = (uint32_t) gamma; h.parent[gamma] = (uint32_t) i; }
    if (hi == gamma + 1) { h.right[i] = (uint32_t) (gamma + 1) | LB_LEAF_BIT; h.leaf_parent[gamma + 1] = (uint32_t) i; }
    else { h.right[i] = (uint32_t) (gamma + 1); h.parent[gamma + 1] = (uint32_t) i; }
    if (i == 0) h.parent[0] = 0xffffffffu;
}

// loads that must see another CU's stores (the sibling subtree's box): device-scope atomics, past this CU's L1
__device__ __forceinline__ float ld_dev(const float *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ uint32_t ld_dev(const uint32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__device__ __forceinline__ Box6 child_box(uint32_t ref, const Box6 *tri_box, const unsigned long long *keys, const Box6 *node_box, bool fresh) {
    Box6 b;
    if (ref & LB_LEAF_BIT) return tri_box[(uint32_t) keys[ref & ~LB_LEAF_BIT]];
    const Box6 *p = node_box + ref;
    if (!fresh) return *p;
#pragma unroll
    for (int a = 0; a < 3; ++a) { b.lo[a] = ld_dev(&p->lo[a]); b.hi[a] = ld_dev(&p->hi[a]); }
    return b;
}

__global__ void __launch_bounds__(LB_BLOCK)
k_refit(const unsigned long long *keys, int n, Hier h, const Box6 *tri_box, Box6 *node_box, uint32_t *height, uint32_t *arrived) {
    const int k = (int) (blockIdx.x * LB_BLOCK + threadIdx.x);
    if (k >= n) return;
    uint32_t cur = h.leaf_parent[k];
    for (;;) {
        __threadfence();                                          // this thread's box stores before its arrival is counted
        if (atomicAdd(&arrived[cur], 1u) == 0u) return;           // the first to arrive leaves the node to the second
        __threadfence();
        const uint32_t l = h.left[cur], r = h.right[cur];
        const Box6 a = child_box(l, tri_box, keys, node_box, true), b = child_box(r, tri_box, keys, node_box, true);
        Box6 u;
#pragma unroll
        for (int x = 0; x < 3; ++x) { u.lo[x] = fminf(a.lo[x], b.lo[x]); u.hi[x] = fmaxf(a.hi[x], b.hi[x]); }
        node_box[cur] = u;
        const uint32_t hl = (l & LB_LEAF_BIT) ? 0u : ld_dev(&height[l]), hr = (r & LB_LEAF_BIT) ? 0u : ld_dev(&height[r]);
        height[cur] = 1u + (hl > hr ? hl : hr);
        const uint32_t p = h.parent[cur];
        if (p == 0xffffffffu) return;
        cur = p;
    }
}

__global__ void __launch_bounds__(LB_BLOCK)
k_keep(int n, Hier h, uint32_t leaf_size, uint32_t *keep) {
    const int i = (int) (blockIdx.x * LB_BLOCK + threadIdx.x);
    if (i >= n - 1) return;
    keep[i] = (h.last[i] - h.first[i] + 1u > leaf_size) ? 1u : 0u;
}

// child ref of the final tree: a leaf (first key << 5 | count) for a single key or a range of <= leaf_size keys, else the node's new index
__device__ __forceinline__ uint32_t final_ref(uint32_t ref, const Hier &h, uint32_t leaf_size, const uint32_t *new_index) {
    if (ref & LB_LEAF_BIT) return LB_LEAF_BIT | ((ref & ~LB_LEAF_BIT) << 5) | 1u;
    const uint32_t cnt = h.last[ref] - h.first[ref] + 1u;
    if (cnt <= leaf_size) return LB_LEAF_BIT | (h.first[ref] << 5) | cnt;
    return new_index[ref];
}

__global__ void __launch_bounds__(LB_BLOCK)
k_emit(const unsigned long long *keys, int n, Hier h, uint32_t leaf_size, const uint32_t *keep, const uint32_t *new_index,
       const Box6 *tri_box, const Box6 *node_box, float pad, float4 *nodes) {
    const int i = (int) (blockIdx.x * LB_BLOCK + threadIdx.x);
    if (i >= n - 1 || !keep[i]) return;
    const uint32_t l = h.left[i], r = h.right[i];
    Box6 a = child_box(l, tri_box, keys, node_box, false), b = child_box(r, tri_box, keys, node_box, false);
#pragma unroll
    for (int x = 0; x < 3; ++x) { a.lo[x] -= pad; a.hi[x] += pad; b.lo[x] -= pad; b.hi[x] += pad; }
    float4 *o = nodes + (size_t) new_index[i] * 4;                    // msk_bvh.h: the two children's bounds interleaved
    o[0] = make_float4(a.lo[0], b.lo[0], a.lo[1], b.lo[1]);
    o[1] = make_float4(a.lo[2], b.lo[2], a.hi[0], b.hi[0]);
    o[2] = make_float4(a.hi[1], b.hi[1], a.hi[2], b.hi[2]);
    o[3] = make_float4(__uint_as_float(final_ref(l, h, leaf_size, new_index)), __uint_as_float(final_ref(r, h, leaf_size, new_index)), 0.f, 0.f);
}

// leaf-ordered triangle records + D10 bounds: the fp32 operations of msk_bvh.h's build() (this file is compiled with
// -ffp-contract=off like everything else), so both builders hand the traversal the same numbers
__global__ void __launch_bounds__(LB_BLOCK)
k_tris(const unsigned long long *keys, uint32_t n, const float4 *tri_verts, const int4 *mesh_info, const float4 *bsdfs, uint32_t n_bsdfs,
       uint32_t bsdf_f4, uint32_t class_shift, float tri_pad, float4 *tris, float4 *bounds) {
    const uint32_t k = blockIdx.x * LB_BLOCK + threadIdx.x;
    if (k >= n) return;
    const uint32_t prim = (uint32_t) keys[k];
    const float4 a = tri_verts[(size_t) prim * 3], b = tri_verts[(size_t) prim * 3 + 1], c = tri_verts[(size_t) prim * 3 + 2];
    const float p[9] = {a.x, a.y, a.z, b.x, b.y, b.z, c.x, c.y, c.z};
    const float e1[3] = {p[0] - p[3], p[1] - p[4], p[2] - p[5]};
    const float e2[3] = {p[6] - p[0], p[7] - p[1], p[8] - p[2]};
    const float ng[3] = {e2[1] * e1[2] - e2[2] * e1[1], e2[2] * e1[0] - e2[0] * e1[2], e2[0] * e1[1] - e2[1] * e1[0]};
    uint32_t word = prim;
    if (class_shift) {                      // material class of the triangle's BSDF (msk_kernels.h: MSK_CLASS_SHIFT)
        const int bsdf = mesh_info[__float_as_uint(a.w)].x;
        const uint32_t cls = (bsdf >= 0 && (uint32_t) bsdf < n_bsdfs) ? (uint32_t) __float_as_int(bsdfs[(size_t) bsdf * bsdf_f4].x) : 0u;
        word |= (cls & 3u) << class_shift;
    }
    float4 *t = tris + (size_t) k * 4;
    t[0] = make_float4(p[0], p[1], p[2], __uint_as_float(word));
    t[1] = make_float4(e1[0], e1[1], e1[2], 0.f);
    t[2] = make_float4(e2[0], e2[1], e2[2], 0.f);
    t[3] = make_float4(ng[0], ng[1], ng[2], 0.f);
    float lo[3], hi[3];
#pragma unroll
    for (int x = 0; x < 3; ++x) {
        const float w0 = p[x], w1 = p[x] - e1[x], w2 = p[x] + e2[x];
        const float mn = w1 < w2 ? w1 : w2, mx = w1 < w2 ? w2 : w1;       // std::min / std::max of msk_bvh.h
        lo[x] = (mn < w0 ? mn : w0) - tri_pad;
        hi[x] = (w0 < mx ? mx : w0) + tri_pad;
    }
    bounds[(size_t) k * 2] = make_float4(lo[0], lo[1], lo[2], 0.f);
    bounds[(size_t) k * 2 + 1] = make_float4(hi[0], hi[1], hi[2], 0.f);
}

// ------------------------------------------------------------------------------------------
// exclusive scan of uint32: a block scans a tile of LB_SCAN_TILE elements (eight per thread) and leaves the tile's sum; the sums
// are scanned the same way (three levels reach 2048^3 elements) and added back
// ------------------------------------------------------------------------------------------
#define LB_SCAN_ITEMS 8
#define LB_SCAN_TILE (LB_BLOCK * LB_SCAN_ITEMS)
__global__ void __launch_bounds__(LB_BLOCK)
k_scan_tile(const uint32_t *in, uint32_t *out, uint32_t n, uint32_t *tile_sums) {
    __shared__ uint32_t wave_sum[LB_BLOCK / 64];
    const uint32_t base = blockIdx.x * LB_SCAN_TILE + threadIdx.x * LB_SCAN_ITEMS, lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
    uint32_t v[LB_SCAN_ITEMS], sum = 0;
#pragma unroll
    for (int k = 0; k < LB_SCAN_ITEMS; ++k) { v[k] = base + k < n ? in[base + k] : 0u; sum += v[k]; }
    uint32_t incl = sum;                                  // inclusive scan of the threads' sums across the wave
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const uint32_t t = __shfl_up(incl, d, 64); if (lane >= (uint32_t) d) incl += t; }
    if (lane == 63) wave_sum[w] = incl;
    __syncthreads();
    uint32_t before = 0, total = 0;
#pragma unroll
    for (uint32_t k = 0; k < LB_BLOCK / 64; ++k) { if (k < w) before += wave_sum[k]; total += wave_sum[k]; }
    uint32_t run = before + incl - sum;
#pragma unroll
    for (int k = 0; k < LB_SCAN_ITEMS; ++k) { if (base + k < n) out[base + k] = run; run += v[k]; }
    if (threadIdx.x == 0 && tile_sums) tile_sums[blockIdx.x] = total;
}
__global__ void __launch_bounds__(LB_BLOCK)
k_scan_add(uint32_t *out, uint32_t n, const uint32_t *tile_offsets) {
    const uint32_t add = tile_offsets[blockIdx.x];
    const uint32_t base = blockIdx.x * LB_SCAN_TILE + threadIdx.x * LB_SCAN_ITEMS;
#pragma unroll
    for (int k = 0; k < LB_SCAN_ITEMS; ++k) if (base + k < n) out[base + k] += add;
}
static size_t scan_tmp_words(size_t n) {                  // words of scratch exclusive_scan needs for n elements
    const size_t t1 = (n + LB_SCAN_TILE - 1) / LB_SCAN_TILE, t2 = (t1 + LB_SCAN_TILE - 1) / LB_SCAN_TILE;
    return 2 * t1 + 2 * t2 + 8;
}
static void exclusive_scan(hipStream_t stream, const uint32_t *in, uint32_t *out, uint32_t n, uint32_t *tmp) {
    const uint32_t t1 = (n + LB_SCAN_TILE - 1) / LB_SCAN_TILE;
    if (t1 <= 1) { hipLaunchKernelGGL(k_scan_tile, dim3(1), dim3(LB_BLOCK), 0, stream, in, out, n, (uint32_t *) nullptr); return; }
    uint32_t *sums = tmp, *offs = tmp + t1, *rest = tmp + 2 * (size_t) t1;
    hipLaunchKernelGGL(k_scan_tile, dim3(t1), dim3(LB_BLOCK), 0, stream, in, out, n, sums);
    exclusive_scan(stream, sums, offs, t1, rest);
    hipLaunchKernelGGL(k_scan_add, dim3(t1), dim3(LB_BLOCK), 0, stream, out, n, offs);
}

// ------------------------------------------------------------------------------------------
// stable LSD radix sort, one 8-bit digit per pass.  Tile = LB_RS_TILE consecutive keys; hist is digit-major ([digit][tile]) so that
// its exclusive scan is, for every (digit, tile), where that tile's first key of that digit goes.
// ------------------------------------------------------------------------------------------
#define LB_RS_TILE 2048
__global__ void __launch_bounds__(LB_BLOCK)
k_rs_hist(const unsigned long long *keys, uint32_t n, uint32_t shift, uint32_t n_tiles, uint32_t *hist) {
    __shared__ uint32_t h[256];
    h[threadIdx.x] = 0u;
    __syncthreads();
    const uint32_t base = blockIdx.x * LB_RS_TILE;
    for (uint32_t k = threadIdx.x; k < LB_RS_TILE && base + k < n; k += LB_BLOCK)
        atomicAdd(&h[(uint32_t) (keys[base + k] >> shift) & 255u], 1u);
    __syncthreads();
    hist[(size_t) threadIdx.x * n_tiles + blockIdx.x] = h[threadIdx.x];
}
__global__ void __launch_bounds__(64)
k_rs_scatter(const unsigned long long *keys, unsigned long long *out, uint32_t n, uint32_t shift, uint32_t n_tiles, const uint32_t *offsets) {
    __shared__ uint32_t pos[256];
    const uint32_t lane = threadIdx.x;
    for (uint32_t d = lane; d < 256u; d += 64u) pos[d] = offsets[(size_t) d * n_tiles + blockIdx.x];
    __syncthreads();
    const uint32_t base = blockIdx.x * LB_RS_TILE;
    for (uint32_t c = 0; c < LB_RS_TILE && base + c < n; c += 64u) {           // (the condition is uniform: the whole wave leaves together)
        const bool live = base + c + lane < n;
        const unsigned long long key = live ? keys[base + c + lane] : 0ull;
        const uint32_t d = (uint32_t) (key >> shift) & 255u;
        unsigned long long peers = __ballot(live);                             // the live lanes that hold the same digit
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const bool bit = (d >> b) & 1u;
            const unsigned long long m = __ballot(bit);
            peers &= bit ? m : ~m;
        }
        const uint32_t rank = (uint32_t) __popcll(peers & ((1ull << lane) - 1ull)), cnt = (uint32_t) __popcll(peers);
        const uint32_t first = live ? pos[d] : 0u;
        __syncthreads();                                                       // every lane has read before the digit's first lane moves it on
        if (live) {
            out[first + rank] = key;
            if (rank == 0u) pos[d] = first + cnt;
        }
        __syncthreads();
    }
}

// keys (code << 32 | index, index ascending) -> k1 sorted; k0 is scratch afterwards.  hist / hist_off: 256 * ceil(n / LB_RS_TILE) words
// each, scan_tmp: scan_tmp_words(max(that, n)) words
static hipError_t sort_by_morton(hipStream_t stream, unsigned long long *k0, unsigned long long *k1, uint32_t n, uint32_t *hist, uint32_t *hist_off,
                                 uint32_t *scan_tmp) {
    const uint32_t n_tiles = (n + LB_RS_TILE - 1) / LB_RS_TILE;
    for (uint32_t pass = 0; pass < 4; ++pass) {              // 30 bits of Morton code: four 8-bit passes, k0 -> k1 -> k0 -> k1 -> k0
        const unsigned long long *src = (pass & 1u) ? k1 : k0;
        unsigned long long *dst = (pass & 1u) ? k0 : k1;
        const uint32_t shift = 32u + 8u * pass;
        hipLaunchKernelGGL(k_rs_hist, dim3(n_tiles), dim3(LB_BLOCK), 0, stream, src, n, shift, n_tiles, hist);
        exclusive_scan(stream, hist, hist_off, 256u * n_tiles, scan_tmp);
        hipLaunchKernelGGL(k_rs_scatter, dim3(n_tiles), dim3(64), 0, stream, src, dst, n, shift, n_tiles, hist_off);
    }
    return hipMemcpyAsync(k1, k0, (size_t) n * 8, hipMemcpyDeviceToDevice, stream);
}

#define LB_TRY(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { snprintf(err, err_len, "%s: %s", #x, hipGetErrorString(e_)); rc = -1; goto done; } } while (0)

int build(hipStream_t stream, const Input &in, float4 *nodes, float4 *tris, float4 *bounds, Result *out, char *err, size_t err_len) {
    int rc = 0;
    const uint32_t n = in.n_tris;
    out->root_ref = 0; out->depth = 0; out->n_nodes = 0;
    if (n == 0) return 0;
    const uint32_t leaf_size = in.leaf_size < 1 ? 1 : in.leaf_size;
    char *pool = nullptr;
    // one allocation for the temporaries
    size_t off = 0;
    auto take = [&](size_t bytes) { const size_t o = off; off += (bytes + 255) & ~(size_t) 255; return o; };
    const size_t o_box = take((size_t) n * sizeof(Box6)), o_k0 = take((size_t) n * 8), o_k1 = take((size_t) n * 8);
    const size_t o_first = take((size_t) n * 4), o_last = take((size_t) n * 4), o_left = take((size_t) n * 4), o_right = take((size_t) n * 4),
                 o_par = take((size_t) n * 4), o_lpar = take((size_t) n * 4), o_nbox = take((size_t) n * sizeof(Box6)),
                 o_h = take((size_t) n * 4), o_arr = take((size_t) n * 4), o_keep = take((size_t) n * 4), o_new = take((size_t) n * 4);
    const uint32_t n_tiles = (n + LB_RS_TILE - 1) / LB_RS_TILE;
    const size_t n_hist = (size_t) 256 * n_tiles;
    const size_t o_hist = take(n_hist * 4), o_hoff = take(n_hist * 4), o_stmp = take(scan_tmp_words(std::max<size_t>(n_hist, n)) * 4);
    const uint32_t grid = (n + LB_BLOCK - 1) / LB_BLOCK;
    LB_TRY(hipMalloc((void **) &pool, off));
    {
        Box6 *tri_box = (Box6 *) (pool + o_box);
        unsigned long long *k0 = (unsigned long long *) (pool + o_k0), *k1 = (unsigned long long *) (pool + o_k1);
        Hier h{(uint32_t *) (pool + o_first), (uint32_t *) (pool + o_last), (uint32_t *) (pool + o_left), (uint32_t *) (pool + o_right),
               (uint32_t *) (pool + o_par), (uint32_t *) (pool + o_lpar)};
        Box6 *node_box = (Box6 *) (pool + o_nbox);
        uint32_t *height = (uint32_t *) (pool + o_h), *arrived = (uint32_t *) (pool + o_arr), *keep = (uint32_t *) (pool + o_keep),
                 *new_index = (uint32_t *) (pool + o_new);
        uint32_t *hist = (uint32_t *) (pool + o_hist), *hist_off = (uint32_t *) (pool + o_hoff), *scan_tmp = (uint32_t *) (pool + o_stmp);
        const float ex = in.hi[0] - in.lo[0], ey = in.hi[1] - in.lo[1], ez = in.hi[2] - in.lo[2];
        const float3 lo = make_float3(in.lo[0], in.lo[1], in.lo[2]);
        const float3 inv = make_float3(ex > 0.f ? 1.f / ex : 0.f, ey > 0.f ? 1.f / ey : 0.f, ez > 0.f ? 1.f / ez : 0.f);
        hipLaunchKernelGGL(k_prims, dim3(grid), dim3(LB_BLOCK), 0, stream, in.tri_verts, n, lo, inv, tri_box, k0);
        LB_TRY(sort_by_morton(stream, k0, k1, n, hist, hist_off, scan_tmp));
        if (n == 1 || n <= leaf_size) {
            out->root_ref = LB_LEAF_BIT | (0u << 5) | n;
        } else {
            LB_TRY(hipMemsetAsync(arrived, 0, (size_t) n * 4, stream));
            LB_TRY(hipMemsetAsync(height, 0, (size_t) n * 4, stream));
            hipLaunchKernelGGL(k_hierarchy, dim3(grid), dim3(LB_BLOCK), 0, stream, k1, (int) n, h);
            hipLaunchKernelGGL(k_refit, dim3(grid), dim3(LB_BLOCK), 0, stream, k1, (int) n, h, tri_box, node_box, height, arrived);
            hipLaunchKernelGGL(k_keep, dim3(grid), dim3(LB_BLOCK), 0, stream, (int) n, h, leaf_size, keep);
            exclusive_scan(stream, keep, new_index, n - 1, scan_tmp);
            hipLaunchKernelGGL(k_emit, dim3(grid), dim3(LB_BLOCK), 0, stream, k1, (int) n, h, leaf_size, keep, new_index, tri_box, node_box,
                               in.box_pad, nodes);
            uint32_t last_keep = 0, last_new = 0, root_height = 0;
            LB_TRY(hipMemcpyAsync(&last_keep, keep + (n - 2), 4, hipMemcpyDeviceToHost, stream));
            LB_TRY(hipMemcpyAsync(&last_new, new_index + (n - 2), 4, hipMemcpyDeviceToHost, stream));
            LB_TRY(hipMemcpyAsync(&root_height, height, 4, hipMemcpyDeviceToHost, stream));
            LB_TRY(hipStreamSynchronize(stream));
            out->n_nodes = last_new + last_keep;
            out->root_ref = 0;                           // node 0 is the root and keeps index 0 (n > leaf_size)
            out->depth = (int) root_height;
        }
        hipLaunchKernelGGL(k_tris, dim3(grid), dim3(LB_BLOCK), 0, stream, k1, n, in.tri_verts, in.mesh_info, in.bsdfs, in.n_bsdfs, in.bsdf_f4,
                           in.class_shift, in.tri_pad, tris, bounds);
        LB_TRY(hipGetLastError());
        LB_TRY(hipStreamSynchronize(stream));
    }
done:
    if (pool) (void) hipFree(pool);
    return rc;
}

}  // namespace msklbvh
