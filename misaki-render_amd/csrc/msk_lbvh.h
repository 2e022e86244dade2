// msk_lbvh.h — interface of the device-side BVH builder (msk_lbvh.hip); see there.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

namespace msklbvh {

struct Input {
    const float4 *tri_verts;     // device, 3 x float4 per triangle in scene-global order: p0 | mesh (uint bits), p1, p2
    uint32_t n_tris;
    float lo[3], hi[3];          // scene bounding box (the Morton grid)
    float box_pad;               // padding of the child boxes (msk_bvh.h: 2 tri_pad)
    float tri_pad;               // padding of the triangles' D10 bounds
    uint32_t leaf_size;          // key ranges of at most this many triangles become leaves
    // material class bits of the prim word (0 = none): bsdf type of the triangle's mesh, shifted by class_shift
    const int4 *mesh_info; const float4 *bsdfs; uint32_t n_bsdfs, bsdf_f4, class_shift;
};
struct Result { uint32_t root_ref; int depth; uint32_t n_nodes; };

// Writes n_nodes 64-byte nodes (capacity: n_tris), n_tris triangle records and bounds (device pointers), synchronises the
// stream.  Returns 0, or -1 with a message in err.
int build(hipStream_t stream, const Input &in, float4 *nodes, float4 *tris, float4 *bounds, Result *out, char *err, size_t err_len);

}  // namespace msklbvh
