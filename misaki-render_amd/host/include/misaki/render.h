// misaki/render.h — render-side interfaces of the host library, mirroring the reference's
// include/misaki/render/*.h for the classes the sampling hot path touches.  Per-sample work does
// not happen here: the "path" integrator flattens these objects into msk_scene_desc and calls the
// C ABI (include/msk_gpu.h).
#pragma once
#include <cstring>
#include "core.h"
#include "msk_gpu.h"

namespace misaki {

class Scene; class Shape; class Mesh; class BSDF; class Emitter; class Sensor; class Film; class Sampler;
class ReconstructionFilter; class ImageBlock; class Integrator;

// include/misaki/core/mathutils.h:17 / core/spectrum.h:75
namespace math { constexpr float Epsilon = 5.9604644775390625e-08f; }       // std::numeric_limits<float>::epsilon() / 2
#define MSK_CIE_Y_NORMALIZATION float(1.0 / 106.7502593994140625)

// include/misaki/render/texture.h — spectra as the back end sees them
class Texture : public Object {
public:
    // plugins that the GPU back end can evaluate describe themselves as a sigmoid-polynomial
    // (render/srgb.h:8-19) optionally multiplied by a scaled D65 table (spectra/srgb_d65.cpp:34-36)
    // ... or as a table on a regular wavelength grid (spectra/regular.cpp): `regular` with lambda_min / lambda_max / values
    struct Flat { float coeff[3] = {0, 0, 0}; float scale = 1.f; float d65_scale = 0.f; bool uses_d65 = false;
                  bool regular = false; float lambda_min = 0.f, lambda_max = 0.f; std::vector<float> values; };
    virtual bool flatten(Flat &out) const { (void) out; return false; }
    // plugins whose value varies with si.uv describe themselves as an msk_texture_desc instead
    virtual bool flatten_texture(msk_texture_desc &out) const { (void) out; return false; }
    virtual float mean() const { return 0.f; }
    static ref<Texture> D65(float scale);
    MSK_DECLARE_CLASS()
protected:
    Texture(const Properties &props) : m_id(props.id()) {}
    std::string m_id;
};

// the scene-level tables a BSDF / emitter adds to while it flattens itself: textures that vary over the surface and
// tabulated spectra (msk_scene_desc::textures / regular_spectra / regular_values)
struct FlatTables {
    std::vector<msk_texture_desc> &textures;
    std::vector<msk_regular_spectrum_desc> &regular;
    std::vector<float> &regular_values;
    uint32_t add_regular(const Texture::Flat &f) {          // -> the 1-based index msk_spectrum_desc::regular holds
        regular.push_back(msk_regular_spectrum_desc{f.lambda_min, f.lambda_max, (uint32_t) f.values.size(), (uint32_t) regular_values.size()});
        regular_values.insert(regular_values.end(), f.values.begin(), f.values.end());
        return (uint32_t) regular.size();
    }
    // a spectrum the device evaluates as scale * S(coeff, l) or as a table; false for the illuminant forms
    bool put(msk_spectrum_desc &d, const Texture::Flat &f) {
        if (f.uses_d65) return false;
        std::memcpy(d.coeff, f.coeff, sizeof f.coeff); d.scale = f.scale; d.regular = f.regular ? add_regular(f) : 0u;
        if (f.regular) { d.coeff[0] = d.coeff[1] = d.coeff[2] = 0.f; d.scale = 1.f; }
        return true;
    }
};

// include/misaki/render/rfilter.h
class ReconstructionFilter : public Object {
public:
    virtual float eval(float x) const = 0;
    float radius() const { return m_radius; }
    uint32_t border_size() const { return m_border_size; }
    const std::vector<float> &values() const { return m_values; }
    MSK_DECLARE_CLASS()
protected:
    ReconstructionFilter(const Properties &) {}
    void init_discretization();            // rfilter.cpp:12-27
    std::vector<float> m_values;
    float m_radius = 0, m_scale_factor = 0;
    uint32_t m_border_size = 0;
};

// include/misaki/render/imageblock.h — only what the boundary needs: a borderless H x W x C buffer
class ImageBlock : public Object {
public:
    ImageBlock(const Vector2i &size, size_t channel_count);
    const Vector2i &size() const { return m_size; }
    const Vector2i &offset() const { return m_offset; }
    void set_offset(const Vector2i &o) { m_offset = o; }
    size_t channel_count() const { return m_channel_count; }
    std::vector<float> &data() { return m_data; }
    const std::vector<float> &data() const { return m_data; }
    void clear();
    void put(const ImageBlock *block);     // imageblock.cpp:36-53 (borderless source and target)
    MSK_DECLARE_CLASS()
private:
    Vector2i m_offset, m_size;
    size_t m_channel_count;
    std::vector<float> m_data;
};

// include/misaki/render/film.h
class Film : public Object {
public:
    virtual void prepare(const std::vector<std::string> &channels) = 0;
    virtual void put(const ImageBlock *block) = 0;
    virtual void set_destination_file(const std::string &filename) = 0;
    virtual void develop() = 0;
    virtual std::vector<float> image() = 0;            // R,G,B,A (+ AOV channels) float, row-major
    virtual std::vector<std::string> image_channels() const { return {"R", "G", "B", "A"}; }
    virtual const ImageBlock *storage() const = 0;
    const Vector2i &size() const { return m_size; }
    const Vector2i &crop_size() const { return m_crop_size; }
    const Vector2i &crop_offset() const { return m_crop_offset; }
    const ReconstructionFilter *filter() const { return m_filter.get(); }
    MSK_DECLARE_CLASS()
protected:
    Film(const Properties &props);          // film.cpp:9-44
    Vector2i m_size, m_crop_size, m_crop_offset;
    ref<ReconstructionFilter> m_filter;
};

// include/misaki/render/sampler.h
class Sampler : public Object {
public:
    size_t sample_count() const { return m_sample_count; }
    uint64_t base_seed() const { return m_base_seed; }
    MSK_DECLARE_CLASS()
protected:
    Sampler(const Properties &props);       // sampler.cpp:7-10
    size_t m_sample_count;
    uint64_t m_base_seed;
};

// include/misaki/render/bsdf.h
class BSDF : public Object {
public:
    // `textures`: the scene's table of surface-varying textures; a BSDF that uses one appends it (msk_bsdf_desc::reflectance_texture)
    virtual bool flatten(msk_bsdf_desc &out, FlatTables &tables) const { (void) out; (void) tables; return false; }
    virtual const BSDF *nested(int side) const { (void) side; return nullptr; }   // twosided: the BSDF of side 0 / 1
    std::string id() const override { return m_id; }
    MSK_DECLARE_CLASS()
protected:
    BSDF(const Properties &props) : m_id(props.id()) {}
    std::string m_id;
};

// include/misaki/render/emitter.h
class Emitter : public Object {
public:
    virtual bool flatten(msk_emitter_desc &out, FlatTables &tables) const { (void) out; (void) tables; return false; }
    virtual bool is_environment() const { return false; }
    virtual bool is_surface() const { return false; }
    void set_shape(Shape *shape);
    Shape *shape() const { return m_shape; }
    MSK_DECLARE_CLASS()
protected:
    Emitter(const Properties &) {}
    Shape *m_shape = nullptr;
};

// include/misaki/render/shape.h, mesh.h
class Shape : public Object {
public:
    const BSDF *bsdf() const { return m_bsdf.get(); }
    const Emitter *emitter() const { return m_emitter.get(); }
    bool is_emitter() const { return (bool) m_emitter; }
    virtual bool is_mesh() const { return false; }
    std::string id() const override { return m_id; }
    MSK_DECLARE_CLASS()
protected:
    Shape(const Properties &props);         // shape.cpp:14-57
    void set_children();
    std::string m_id;
    ref<BSDF> m_bsdf;
    ref<Emitter> m_emitter;
};

class Mesh : public Shape {
public:
    uint32_t vertex_count() const { return m_vertex_count; }
    uint32_t face_count() const { return m_face_count; }
    const float *vertices() const { return m_vertices.data(); }     // 8 floats per vertex [p n uv]
    const uint32_t *faces() const { return m_faces.data(); }
    bool has_vertex_normals() const { return m_normal_offset != 0; }
    bool has_vertex_texcoords() const { return m_texcoord_offset != 0; }
    bool is_mesh() const override { return true; }
    MSK_DECLARE_CLASS()
protected:
    Mesh(const Properties &props);
    std::vector<float> m_vertices;
    std::vector<uint32_t> m_faces;
    uint32_t m_vertex_count = 0, m_face_count = 0, m_normal_offset = 0, m_texcoord_offset = 0;
    Transform4f m_to_world;
    std::string m_name;
};

// include/misaki/render/sensor.h
class Sensor : public Object {
public:
    Film *film() const { return m_film.get(); }
    Sampler *sampler() const { return m_sampler.get(); }
    const Transform4f &world_transform() const { return m_world_transform; }
    virtual bool flatten(msk_camera_desc &out) const { (void) out; return false; }
    MSK_DECLARE_CLASS()
protected:
    Sensor(const Properties &props);        // sensor.cpp:9-45
    Transform4f m_world_transform;
    ref<Film> m_film;
    ref<Sampler> m_sampler;
    float m_aspect = 1.f;
};

class ProjectiveCamera : public Sensor {
public:
    float near_clip() const { return m_near_clip; }
    float far_clip() const { return m_far_clip; }
    MSK_DECLARE_CLASS()
protected:
    ProjectiveCamera(const Properties &props);   // sensor.cpp:136-141
    float m_near_clip, m_far_clip, m_focus_distance;
};

// include/misaki/render/integrator.h:9-62
class Integrator : public Object {
public:
    virtual bool render(Scene *scene, Sensor *sensor) = 0;
    MSK_DECLARE_CLASS()
protected:
    Integrator(const Properties &) {}
};

class SamplingIntegrator : public Integrator {
public:
    virtual std::vector<std::string> aov_names() const { return {}; }   // integrator.cpp:28
    MSK_DECLARE_CLASS()
protected:
    SamplingIntegrator(const Properties &props);   // integrator.cpp:18-24
    uint32_t m_block_size;
    bool m_hide_emitters;
};

class MonteCarloIntegrator : public SamplingIntegrator {
public:
    MSK_DECLARE_CLASS()
protected:
    MonteCarloIntegrator(const Properties &props); // integrator.cpp:128-137
    int m_max_depth, m_rr_depth;
};

// include/misaki/render/scene.h
class Scene : public Object {
public:
    Scene(const Properties &props);         // scene.cpp:26-64
    const std::vector<ref<Shape>> &shapes() const { return m_shapes; }
    const std::vector<ref<Emitter>> &emitters() const { return m_emitters; }
    Sensor *sensor() const { return m_sensor.get(); }
    Integrator *integrator() const { return m_integrator.get(); }
    const Emitter *environment() const { return m_environment.get(); }
    MSK_DECLARE_CLASS()
private:
    std::vector<ref<Shape>> m_shapes;
    std::vector<ref<Emitter>> m_emitters;
    ref<Emitter> m_environment;
    ref<Sensor> m_sensor;
    ref<Integrator> m_integrator;
};

// ---- flattening (the step the "path" plugin performs before calling the C ABI)
struct FlatScene {
    msk_scene_desc desc;
    std::vector<msk_mesh_desc> meshes;
    std::vector<msk_bsdf_desc> bsdfs;
    std::vector<msk_emitter_desc> emitters;
    std::vector<msk_texture_desc> textures;
    std::vector<msk_regular_spectrum_desc> regular;      // tabulated spectra (ABI v7) and their values
    std::vector<float> regular_values;
    std::vector<float> vertices;
    std::vector<uint32_t> faces;
    msk_render_params params;
};
void flatten_scene(const Scene *scene, const Sensor *sensor, FlatScene &out);

// spectral data owned by the host (core/spectrum.h:78-80, spectra/d65.cpp:12-27)
const float *cie1931_xyz_table();   // 3 * 95
const float *d65_table();           // 95
Color3 srgb_model_fetch(const Color3 &rgb);     // src/librender/srgb.cpp:11-28: trilinear fetch in the res-64 sRGB table (rgb2spec.cpp)
const std::string &srgb_model_source();         // where that table came from (a file, or "(computed)")
// the table itself (rgb2spec_table.cpp; replaces ext/rgb2spec): optimiser, file I/O ("SPEC" layout of data/srgb.coeff), fetch
void rgb2spec_build_table(int res, std::vector<float> &scale, std::vector<float> &data, int threads);
void rgb2spec_write_table(const std::string &path, const std::vector<float> &scale, const std::vector<float> &data);
bool rgb2spec_read_table(const std::string &path, std::vector<float> &scale, std::vector<float> &data);
void rgb2spec_fetch_table(int res, const float *scale, const float *data, const float rgb_in[3], float out[3]);

// image output (core/image.h): float RGBA
void write_pfm(const std::string &path, int w, int h, int channels, const float *data);
void write_exr(const std::string &path, int w, int h, const std::vector<std::string> &channels, const float *data);

}  // namespace misaki
