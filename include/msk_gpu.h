/*
 * msk_gpu.h — C ABI of the MI355X path-tracing back end for misaki-render.
 *
 * This is the drop-in boundary for ONE hot path of the reference:
 *
 *   SamplingIntegrator::render          src/librender/integrator.cpp:31-80
 *     -> render_block / render_sample   src/librender/integrator.cpp:82-126
 *     -> PathTracer::sample             src/librender/integrators/path.cpp:23-125
 *     -> Scene::ray_intersect/ray_test  src/librender/scene.cpp:216-273 (Embree3)
 *     -> ImageBlock::put / Film::put    src/librender/imageblock.cpp:36-114
 *
 * The reference has no FFI of its own (it is one C++ shared library); the entry
 * points below are what its `"path"` Integrator plugin
 * (include/misaki/render/integrator.h:9-17, integrators/path.cpp:139-140)
 * binds instead of running the Embree3+TBB tile loop.  INTEGRATION.md shows the
 * plugin-side stub.  Plain pointers and sizes only — no C++ or torch types.
 *
 * Conventions
 *   - every function returns MSK_OK (0) or a negative MSK_ERR_* code; the text
 *     of the last error is available from msk_gpu_last_error().  No exception
 *     crosses this boundary (the plugin turns a non-zero code into the
 *     reference's `Throw`, include/misaki/core/logger.h:81-88).
 *   - the caller owns every host array for the duration of the call that takes
 *     it; the library copies what it needs into HBM.
 *   - handles are opaque and destroyed explicitly.
 *   - one calling thread per msk_ctx at a time (the reference calls render()
 *     from a single worker thread, src/apps/main.cpp:37).  A render call drives
 *     four HIP streams of its own from the calling thread (MSK_HOST_THREADS=4:
 *     from three short-lived helper threads as well) and sleeps while the
 *     device works; every wait has a wall limit (MSK_WATCHDOG_S): the call
 *     returns, with MSK_ERR_HIP and a lost context if the device stopped
 *     answering.  The streams are created at the device's highest stream
 *     priority (MSK_STREAM_PRIORITY) so that they do not share hardware queues
 *     with the application's other streams.
 */
#ifndef MSK_GPU_H
#define MSK_GPU_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MSK_ABI_VERSION 7

/* ---- status codes ------------------------------------------------------- */
#define MSK_OK                 0
#define MSK_ERR_INVALID_ARG   (-1)
#define MSK_ERR_NO_DEVICE     (-2)
#define MSK_ERR_HIP           (-3)
#define MSK_ERR_OOM           (-4)
#define MSK_ERR_UNSUPPORTED   (-5)

/* ---- plugin type tags (the reference's registered plugin names) --------- */
#define MSK_BSDF_DIFFUSE        0  /* "diffuse"        bsdfs/diffuse.cpp:73          */
#define MSK_BSDF_ROUGHCONDUCTOR 1  /* "roughconductor" bsdfs/roughconductor.cpp:139 (GGX only) */
#define MSK_BSDF_ROUGHDIELECTRIC 2 /* "roughdielectric" bsdfs/roughdielectric.cpp:209 (GGX only) */
#define MSK_EMITTER_AREA       0   /* "area"     emitters/area.cpp:61          */
#define MSK_TEXTURE_CHECKERBOARD 1 /* "checkerboard" textures/checkerboard.cpp:52 */

#define MSK_EMITTER_CONSTANT   1   /* "constant" emitters/constant.cpp:95 (environment; mesh_id = -1, at most one) */

/* AOV channel groups of the "aov" integrator (integrators/aov.cpp:21-28,87-144); channels per type: 1 3 2 3 3 4 */
#define MSK_AOV_DEPTH          0   /* si.t, 0 on a miss                                  (aov.cpp:97-99)   */
#define MSK_AOV_POSITION       1   /* si.p                                               (aov.cpp:101-105) */
#define MSK_AOV_UV             2   /* si.uv                                              (aov.cpp:107-110) */
#define MSK_AOV_GEO_NORMAL     3   /* si.n                                               (aov.cpp:112-116) */
#define MSK_AOV_SH_NORMAL      4   /* si.sh_frame.n                                      (aov.cpp:118-122) */
#define MSK_AOV_PATH_RGBA      5   /* nested "path" integrator: xyz_to_srgb(XYZ of its sample), 1 (aov.cpp:124-141) */

/* RNG semantics (the reference's own seeding is not reproducible) */
#define MSK_RNG_PCG_BLOCK      0   /* samplers/independent.cpp as written: one PCG32 stream per image block, drawn from in the scalar
                                      loops' order.  Sequential by construction: the device renders one block per LANE (msk_serial.h) —
                                      a fidelity mode (BASELINE config 1 in about 1.2 seconds, one block per wave), bit-identical to the oracle's; msk_gpu_render /
                                      _render_device only (not the "aov" integrator, not msk_gpu_sample_pixels) */
#define MSK_RNG_COUNTER        1   /* stateless hash of (seed,pixel,sample,dim): the hot path (wavefront kernels); GPU + oracle */

#define MSK_CIE_SAMPLES        95  /* include/misaki/core/spectrum.h:75        */
#define MSK_FILTER_RESOLUTION  32  /* include/misaki/render/rfilter.h:6        */

/*
 * One triangle mesh = one reference `Mesh` (include/misaki/render/mesh.h:86-90).
 * Vertices use the reference's interleaved layout, 8 floats per vertex
 * [px py pz nx ny nz u v] (shapes/obj.cpp:137-142), already in world space
 * (obj.cpp:90,102).  Faces are 3 uint32 per triangle, local to the mesh.
 * Mesh order == the reference's m_shapes order == Embree geomID (scene.cpp:235-240).
 */
typedef struct msk_mesh_desc {
    uint32_t first_vertex;   /* offset into msk_scene_desc.vertices, in vertices */
    uint32_t vertex_count;
    uint32_t first_face;     /* offset into msk_scene_desc.faces, in triangles  */
    uint32_t face_count;
    int32_t  bsdf_id;        /* index into bsdfs[]                              */
    int32_t  emitter_id;     /* index into emitters[], -1 = not an emitter      */
    uint32_t has_normals;    /* Mesh::has_vertex_normals()  (mesh.h:66)         */
    uint32_t has_texcoords;  /* Mesh::has_vertex_texcoords() (mesh.h:67)        */
} msk_mesh_desc;

/*
 * A tabulated spectrum on a regular wavelength grid = the reference's `regular` texture plugin
 * (spectra/regular.cpp:27-91,148 — what <spectrum value="l0:v0, l1:v1, ..."/> with equidistant
 * wavelengths becomes, xml.cpp:300-341; the loader has already multiplied the values of a spectrum inside an
 * <emitter> by MSK_CIE_Y_NORMALIZATION, xml.cpp:306-314).  value(l) = lerp of values[first_value + i], i = 0 .. size-1,
 * at x = (l - lambda_min) * inv_interval with inv_interval = float(1 / ((double) lambda_max - lambda_min) / (size - 1)))
 * and the segment index clamped to [0, size - 2] (RegularSpectrum::eval -> eval_pdf, regular.cpp:73-91: outside the
 * table the end segments are continued linearly).  ABI v7.  2 <= size <= MSK_REGULAR_MAX (the D65 and CIE tables'
 * 95; regular.cpp:30-31 needs two entries), lambda_min < lambda_max, values finite and non-negative (regular.cpp:59-60).
 * An `irregular` spectrum (unequal steps) is not part of this ABI: the host loader refuses it with its own message.
 */
#define MSK_REGULAR_MAX 95
typedef struct msk_regular_spectrum_desc {
    float    lambda_min, lambda_max;
    uint32_t size;
    uint32_t first_value;    /* offset into msk_scene_desc.regular_values       */
} msk_regular_spectrum_desc;

/* A spectrum the device can evaluate.  regular == 0: value(l) = scale * S(coeff, l) with the sigmoid polynomial
 * S of render/srgb.h:8-19; RGB triples above 1 use the normalisation of srgb_d65.cpp:18-22
 * (scale = 2 * max(rgb), coeff = fetch(rgb / scale)) without the D65 factor; scale >= 0.
 * regular == k > 0 (ABI v7): the tabulated spectrum regular_spectra[k - 1] of the scene; coeff / scale are ignored. */
typedef struct msk_spectrum_desc {
    float coeff[3];
    float scale;
    uint32_t regular;
} msk_spectrum_desc;

/*
 * BSDF parameters.
 * MSK_BSDF_DIFFUSE: `reflectance` = the three sigmoid-polynomial coefficients srgb_model_fetch()
 * returned for the RGB reflectance (spectra/srgb.cpp:13-19, srgb.cpp:11-28).
 * MSK_BSDF_ROUGHCONDUCTOR (bsdfs/roughconductor.cpp:12-50): GGX microfacet conductor; alpha_u/alpha_v,
 * sample_visible, and eta / k / specular_reflectance evaluated at the path's four wavelengths (the
 * reference's RGB-typed code does not compile in its spectral build; DESIGN.md §rough conductor).
 * MSK_BSDF_ROUGHDIELECTRIC (bsdfs/roughdielectric.cpp:14-55): GGX microfacet dielectric with reflection and
 * transmission lobes; ior_eta = int_ior / ext_ior, ior_inv_eta = ext_ior / int_ior, alpha_*, sample_visible,
 * specular_reflectance / specular_transmittance.
 * back_bsdf implements the "twosided" adapter (bsdfs/twosided.cpp:38-101): the BSDF evaluated with
 * flipped wi/wo when cos(theta_i) < 0; the entry's own index for twosided(A), -1 for a one-sided BSDF.
 * reflectance_texture: 0 = the diffuse reflectance is the constant `reflectance`; k > 0 = it is
 * textures[k-1] evaluated at the hit's uv (SmoothDiffuse::m_reflectance->eval(si), diffuse.cpp:31,44).
 */
typedef struct msk_bsdf_desc {
    int32_t type;
    int32_t back_bsdf;
    float   reflectance[3];
    float   alpha_u, alpha_v;
    int32_t sample_visible;
    msk_spectrum_desc eta, k, specular_reflectance, specular_transmittance;
    float   ior_eta, ior_inv_eta;
    uint32_t reflectance_texture;
    float   reflectance_scale;   /* diffuse reflectance = S(reflectance, l) * reflectance_scale: 1 for an `srgb` spectrum; a
                                    `uniform` spectrum of value c (spectra/uniform.cpp:16-27, what <spectrum value="c"/> makes
                                    outside an emitter, xml.cpp:285-292) is {0, 0, +inf} with scale c */
    uint32_t reflectance_regular; /* ABI v7: k > 0 = the diffuse reflectance is the tabulated spectrum regular_spectra[k - 1]
                                    (reflectance / reflectance_scale / reflectance_texture are then unused); 0 = not tabulated */
} msk_bsdf_desc;

/*
 * A texture that varies over the surface.  MSK_TEXTURE_CHECKERBOARD (textures/checkerboard.cpp:10-33):
 * uv' = m_transform.transform_affine_point(si.uv) with m_transform = the top-left 3x3 of the `to_uv`
 * 4x4 (Transform4f::extract, core/transform.h:142-148; `to_uv` holds its rows 0 and 1, so the uv offset
 * is the 4x4's z column), u = uv'.x - floor(uv'.x), v likewise; color0 where (u > .5) == (v > .5), else
 * color1.  color0 / color1 are sigmoid-polynomial coefficients of two constant `srgb` spectra (nested
 * textures other than that are not flattened).  si.uv = the interpolated vertex texcoords, or the
 * hit's barycentrics for a mesh without them (mesh.cpp:66,68-72).
 */
typedef struct msk_texture_desc {
    int32_t type;
    float   color0[3];
    float   color1[3];
    float   to_uv[6];        /* m00 m01 m02 / m10 m11 m12 */
    float   reserved[3];
} msk_texture_desc;

/*
 * Area emitter (emitters/area.cpp) with an `srgb_d65` radiance
 * (spectra/srgb_d65.cpp:13-36): radiance(l) = d65(l) * d65_scale * S(coeff, l),
 * d65_scale = scale * 2*max(rgb) / 10568 (srgb_d65.cpp:18-26, d65.cpp:33-34).
 * The constant environment emitter (emitters/constant.cpp) uses the same radiance form (its default is
 * Texture::D65(1): coefficients {0, 0, +inf}); it has no mesh (mesh_id = -1), and its place in `emitters`
 * is its place in Scene::m_emitters (scene.cpp:27-41: XML order, shapes' area lights and top-level emitters mixed).
 */
typedef struct msk_emitter_desc {
    int32_t type;
    int32_t mesh_id;         /* the shape this emitter is attached to          */
    float   radiance[3];     /* sigmoid-polynomial coefficients                */
    float   d65_scale;
    uint32_t radiance_regular; /* ABI v7: k > 0 = radiance(l) is the tabulated spectrum regular_spectra[k - 1] as it stands
                                  (AreaLight / ConstantBackgroundEmitter with a `regular` radiance: no D65 factor, no
                                  sigmoid; radiance / d65_scale unused); 0 = the srgb_d65 form above */
} msk_emitter_desc;

/* PerspectiveCamera (sensors/perspective.cpp:8-42); matrices are row-major 4x4. */
typedef struct msk_camera_desc {
    float sample_to_camera[16];  /* m_sample_to_camera, sample space in PIXELS */
    float to_world[16];          /* m_world_transform                          */
    float near_clip, far_clip;
} msk_camera_desc;

/* Film::size() (film.cpp:10) + ReconstructionFilter discretisation (rfilter.cpp:12-27) + the crop window
 * (film.cpp:12-21,51-63: crop_offset_x / _y, crop_width / _height; HDRFilm's storage is an ImageBlock of the
 * crop size placed at the crop offset, films/hdrfilm.cpp:37-38).  The sensor — sample_to_camera, the pixel
 * grid, the spiral block ids — is that of the FULL film (integrator.cpp:45: BlockGenerator(film->size(), 0,
 * block_size)); the film the render calls write is the crop window, crop_size[1] x crop_size[0] pixels, and
 * holds exactly what Film::put leaves there: every block's contribution clipped to the window
 * (imageblock.cpp:36-53,133-173).  Blocks whose bordered area misses the window add nothing and are not
 * rendered (and not counted in msk_stats).  crop_size = {0, 0} means the whole film. */
typedef struct msk_film_desc {
    int32_t width, height;
    float   filter_radius;                       /* m_radius                    */
    float   filter_lut[MSK_FILTER_RESOLUTION + 1]; /* m_values, [32] == 0       */
    int32_t crop_offset[2];                      /* m_crop_offset (x, y)        */
    int32_t crop_size[2];                        /* m_crop_size (width, height) */
} msk_film_desc;

typedef struct msk_scene_desc {
    uint32_t abi_version;       /* MSK_ABI_VERSION                             */
    uint32_t n_meshes, n_bsdfs, n_emitters;
    const msk_mesh_desc    *meshes;
    const msk_bsdf_desc    *bsdfs;
    const msk_emitter_desc *emitters;
    const float    *vertices;   /* n_vertices * 8 floats                       */
    const uint32_t *faces;      /* n_faces * 3                                 */
    uint32_t n_vertices, n_faces;
    msk_camera_desc camera;
    msk_film_desc   film;
    /* spectral tables owned by the host side of the reference
       (spectrum.cpp:8-111: x,y,z each 95 samples 360..830 nm; d65.cpp:12-27) */
    const float *cie1931_xyz;   /* 3 * MSK_CIE_SAMPLES                         */
    const float *d65;           /* MSK_CIE_SAMPLES                             */
    uint32_t n_textures;        /* may be 0 (textures then unused)             */
    const msk_texture_desc *textures;
    /* ABI v7: tabulated spectra (may be 0 / NULL) */
    uint32_t n_regular_spectra, n_regular_values;
    const msk_regular_spectrum_desc *regular_spectra;
    const float *regular_values;
} msk_scene_desc;

/*
 * Render parameters = the integrator/sampler properties of the reference
 * (integrator.cpp:20-23,130-136; sampler.cpp:8-9) plus the shard selectors.
 * Defaults that reproduce the reference's effective behaviour:
 * rr_depth 5, max_depth -1, hide_emitters 0, block_size 32.
 */
typedef struct msk_render_params {
    uint32_t spp;            /* sampler sample_count (<= 2^20 per call: shard more with sample_first / sample_stride) */
    uint64_t seed;           /* sampler base_seed                              */
    int32_t  rng_mode;       /* MSK_RNG_*                                      */
    int32_t  rr_depth;
    int32_t  max_depth;
    int32_t  hide_emitters;
    int32_t  block_size;     /* MSK_BLOCK_SIZE = 32 (imageblock.h:8)           */
    /* shard: this call renders the spiral blocks id with
       id % block_stride == block_first (imageblock.cpp:187-247 order) ...     */
    uint32_t block_first, block_stride;
    /* ... and of every pixel the sample indices s = sample_first + k * sample_stride (k = 0, 1, ...) below spp.
       (0,1) = everything; (r,G) = rank r's interleaved share of G; (a,1) with spp = b = the contiguous range [a, b),
       which lets shares of unequal size be handed out (misaki-render_amd/multigpu.py).
       MSK_RNG_PCG_BLOCK takes (0,1) only (MSK_ERR_UNSUPPORTED otherwise): a block's samples share one sequential PCG32 stream
       (samplers/independent.cpp:9-35), so shards of its sample indices would all draw the same numbers; that mode shards
       by blocks.                                                                                                 */
    uint32_t sample_first, sample_stride;
} msk_render_params;

typedef struct msk_stats {
    uint64_t samples;        /* camera samples traced                          */
    uint64_t segments;       /* closest-hit rays (path segments)               */
    uint64_t shadow_rays;    /* occlusion rays                                 */
    uint32_t iterations;     /* wavefront iterations                           */
    uint32_t passes;         /* block groups (record-buffer passes)            */
    float    ms_total;       /* device time of the whole call                  */
    float    ms_generate;    /* always 0: camera-ray generation is fused into the shading kernel (its time
                                is part of ms_shade); the field keeps the struct layout of ABI v4 */
    /* Sums of the kernel durations of the launches that carried timing events, and how many did
       (n_*_launches).  With the environment variable MSK_TIMING_EVERY=n (default 1) only the launches of
       every n-th group of wavefront iterations are timed: ms_trace / ms_shade are then SAMPLES — divide by
       n_*_launches for an average launch, scale by iterations / n_*_launches for an estimate of the total.
       When the wavefront loop runs on several streams launches overlap, so these sums can
       exceed ms_total.  ms_resolve (film replay + Film::put) is always complete. */
    float    ms_trace, ms_shade, ms_resolve;
    uint32_t n_trace_launches, n_shade_launches;
    /* kernel launches the call actually made (timed or not): traversal kernels, k_shade_gen, k_wavefront (the device-side
       loop of the thin end of a pass, booked as up to 16 `iterations` each) */
    uint32_t launches_trace, launches_shade, launches_wavefront;
    /* ABI v6: samples ImageBlock::put would have warned about (imageblock.cpp:57-81, "Invalid sample value: [...]"): a value of
       the sample's X, Y, Z (or of the nested integrator's R, G, B under "aov") that is not finite, or — "path" only: the
       reference switches the test off for blocks with AOV channels, integrator.cpp:59-60 — below -1e-5.  Such a sample is
       splatted all the same, as the reference does; the plugin logs the count at Warn level. */
    uint64_t invalid_samples;
    /* ABI v7: the bytes of SoA path state the call's launches of the shading / traversal kernels were ASKED to move — counted
       by the library from its own layout (what a live slot makes its kernels read and write), not measured:
       shading 176 B per segment (+ 16 in the general variant) - 64 B per sample + 48 B per shadow ray + 20 B per sample record;
       traversal 48 B per segment + 32 B (16 B for trees in HBM) per shadow ray.  bytes / launches_* = the algorithmic bytes per
       launch a roofline needs; HBM traffic is a separate, measured figure (rocprofv3 PMC).  0 for MSK_RNG_PCG_BLOCK renders. */
    uint64_t bytes_shade, bytes_trace;
} msk_stats;

typedef struct msk_ctx   msk_ctx;
typedef struct msk_scene msk_scene;

/* ---- lifetime ------------------------------------------------------------ */
/* device_ids: n >= 1 HIP ordinals (at most 8).  n == 1: an ordinary context on that device.  n > 1: a GROUP context
   (SURVEY §8b writes the entry point with a device list; csrc/msk_multi.h): one member context per entry — an ordinal may
   repeat —, every scene created on it lives on every member, msk_gpu_render / _render_device / _render_aov shard the call's
   samples over the members by index (member k: sample_first + (k + j n) sample_stride; a MSK_RNG_PCG_BLOCK call, whose
   samples cannot be sharded, by spiral block: block_first + (k + j n) block_stride), one host thread each, and sum the
   films on device_ids[0] over peer access in member order; d_film_xyzaw of msk_gpu_render_device is memory of device_ids[0];
   the sub-stage entry points run on the first member.  msk_stats of a group call: samples / segments / shadow_rays / launches_* and
   the kernel-time samples ms_trace / ms_shade / n_*_launches are SUMS over the members (device time of n devices: not comparable
   with ms_total), iterations / passes / ms_resolve the members' maxima, ms_total the HOST WALL TIME of the whole call (the members'
   renders side by side + the film sum).  (The one-process-per-GPU scheme — one single-device msk_ctx per
   rank + an RCCL film reduce — sits above this ABI: DESIGN.md §7.) */
int  msk_gpu_init(const int *device_ids, int n, msk_ctx **out_ctx);
void msk_gpu_shutdown(msk_ctx *ctx);
const char *msk_gpu_last_error(const msk_ctx *ctx); /* ctx may be NULL */

/* replaces Scene::accel_init (scene.cpp:201-212): upload geometry, build BVH,
   area-light tables (mesh.cpp:39-48).
   Node boxes are padded by 1e-5 (the D10 triangle bounds by 0.5e-5) of the scene's scale = max(diagonal, largest |coordinate|),
   so that the slab test never culls what the triangle test accepts.  Measured margin (tests/test_padding_margin.py,
   test_gpu_trees_equal_brute_force_with_a_tenth_of_the_padding): tree and brute force agree down to 1e-7 of the scale and
   part at 1e-8 — the rule keeps a factor of ~100.  The environment variable MSK_PAD_SCALE (margin tests only; the oracle
   reads it too) replaces the 1e-5; a value that is not a number in [1e-7, 1e-3] fails the call with MSK_ERR_INVALID_ARG. */
int  msk_gpu_scene_create(msk_ctx *ctx, const msk_scene_desc *desc, msk_scene **out_scene);
void msk_gpu_scene_destroy(msk_scene *scene);

/* ---- the hot path --------------------------------------------------------- */
/*
 * Replaces the body of SamplingIntegrator::render between film->prepare() and
 * the last film->put() (integrator.cpp:48-76).  Writes the film's weighted sums
 * {X,Y,Z,A,W} per pixel of the crop window (the whole film by default), row-major
 * crop_height*crop_width*5 floats, exactly what HDRFilm's storage ImageBlock holds
 * after the last put (hdrfilm.cpp:37-38,43-46).
 * film_xyzaw: host memory, caller-owned.  If it is pinned (hipHostMalloc / hipHostRegister) the film is copied into it
 * directly by DMA; a pageable array is filled from the library's own pinned staging buffer.  stats may be NULL.
 */
int  msk_gpu_render(msk_scene *scene, const msk_render_params *params,
                    float *film_xyzaw, msk_stats *stats);

/* Same, but the film stays in HBM: d_film_xyzaw is a device pointer on the
   ctx's device (e.g. a torch tensor's data_ptr, reduced over RCCL afterwards).
   hip_stream: a hipStream_t or NULL for the library's own stream; the call
   returns after the work is complete on that stream. */
int  msk_gpu_render_device(msk_scene *scene, const msk_render_params *params,
                           float *d_film_xyzaw, void *hip_stream, msk_stats *stats);

/*
 * AOVIntegrator::render (integrators/aov.cpp:87-144 through SamplingIntegrator::render, integrator.cpp:31-80):
 * the same job with extra film channels.  aov_types: n_aovs values MSK_AOV_*, in the order of the plugin's
 * m_aov_types; at most one MSK_AOV_PATH_RGBA (the nested "path" integrator whose sample also becomes the
 * XYZ result, aov.cpp:138-139; without one XYZ is 0 — the reference returns an uninitialised Spectrum there).
 * Every AOV value of a camera ray that misses the scene is 0 (the reference reads an uninitialised
 * SceneInteraction for all but depth).  film: crop_height*crop_width*(5 + msk_gpu_aov_channels()) floats, host,
 * the weighted sums HDRFilm's storage holds (channels X,Y,Z,A,W, then the AOV channels in order).
 */
uint32_t msk_gpu_aov_channels(const int32_t *aov_types, uint32_t n_aovs);   /* 0 if a type is invalid */
int  msk_gpu_render_aov(msk_scene *scene, const msk_render_params *params,
                        const int32_t *aov_types, uint32_t n_aovs, float *film, msk_stats *stats);

/* ---- sub-stage entry points (parity tests bind these) ---------------------- */
/*
 * Scene::ray_intersect / Scene::ray_test (scene.cpp:216-273) on a batch of
 * rays.  rays: n * 8 floats {ox,oy,oz,tmin, dx,dy,dz,tmax}.
 * closest: out_hit n * 4 floats {t,u,v, bitcast(prim)} with t = +inf on a
 * miss; prim is the scene-global triangle index (mesh.first_face + primID).
 * any: out_occluded n bytes (0/1).  Host pointers.
 */
int  msk_gpu_trace_closest(msk_scene *scene, uint64_t n, const float *rays, float *out_hit);
int  msk_gpu_trace_any(msk_scene *scene, uint64_t n, const float *rays, uint8_t *out_occluded);

/*
 * render_sample (integrator.cpp:103-126) for every sample of the listed
 * pixels, WITHOUT the film splat: out_xyz = n_pixels * spp * 3 floats, the
 * {X,Y,Z} that render_sample hands to ImageBlock::put, out_pos (may be NULL)
 * = n_pixels * spp * 2 floats position_sample.  pixels: n_pixels * 2 int32
 * (x,y).  counter RNG only.  Host pointers.
 */
int  msk_gpu_sample_pixels(msk_scene *scene, const msk_render_params *params,
                           uint64_t n_pixels, const int32_t *pixels,
                           float *out_xyz, float *out_pos);

/* device + build information for logs: fills a NUL-terminated string */
int  msk_gpu_describe(const msk_ctx *ctx, char *buf, uint64_t buf_size);

#ifdef __cplusplus
}
#endif
#endif /* MSK_GPU_H */
