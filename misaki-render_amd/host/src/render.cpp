// render.cpp — render-side base classes, the plugins on the sampling hot path, and the flatten step.
// Plugin names, property keys, defaults and error texts follow the reference (files cited inline);
// per-sample work is NOT here: PathTracer::render() hands the flattened scene to the C ABI.
#include <misaki/render.h>

#include <algorithm>
#include <cctype>
#include <chrono>
#include <cstring>
#include <fstream>
#include <unordered_map>

namespace misaki {

// =========================================================================== base classes
MSK_IMPLEMENT_CLASS(Texture, Object, "texture")
MSK_IMPLEMENT_CLASS(ReconstructionFilter, Object, "rfilter")
MSK_IMPLEMENT_CLASS(ImageBlock, Object)
MSK_IMPLEMENT_CLASS(Film, Object, "film")
MSK_IMPLEMENT_CLASS(Sampler, Object, "sampler")
MSK_IMPLEMENT_CLASS(BSDF, Object, "bsdf")
MSK_IMPLEMENT_CLASS(Emitter, Object, "emitter")
MSK_IMPLEMENT_CLASS(Shape, Object, "shape")
MSK_IMPLEMENT_CLASS(Mesh, Shape)
MSK_IMPLEMENT_CLASS(Sensor, Object, "sensor")
MSK_IMPLEMENT_CLASS(ProjectiveCamera, Sensor)
MSK_IMPLEMENT_CLASS(Integrator, Object, "integrator")
MSK_IMPLEMENT_CLASS(SamplingIntegrator, Integrator)
MSK_IMPLEMENT_CLASS(MonteCarloIntegrator, SamplingIntegrator)
MSK_IMPLEMENT_CLASS(Scene, Object, "scene")

// rfilter.cpp:12-27
void ReconstructionFilter::init_discretization() {
    m_values.resize(MSK_FILTER_RESOLUTION + 1);
    float sum = 0.f;
    for (size_t i = 0; i < MSK_FILTER_RESOLUTION; ++i) {
        m_values[i] = eval(float(m_radius * i) / MSK_FILTER_RESOLUTION);
        sum += m_values[i];
    }
    m_values[MSK_FILTER_RESOLUTION] = 0;
    m_scale_factor = float(MSK_FILTER_RESOLUTION) / m_radius;
    m_border_size = (uint32_t) std::ceil(m_radius - .5f);
    sum *= 2 * m_radius / MSK_FILTER_RESOLUTION;
    const float normalization = 1.0f / sum;
    for (size_t i = 0; i < MSK_FILTER_RESOLUTION; ++i) m_values[i] *= normalization;
}

ImageBlock::ImageBlock(const Vector2i &size, size_t channel_count) : m_size(size), m_channel_count(channel_count) {
    m_data.assign((size_t) size.x * size.y * channel_count, 0.f);
}
void ImageBlock::clear() { std::fill(m_data.begin(), m_data.end(), 0.f); }
void ImageBlock::put(const ImageBlock *block) {         // accumulate_2d with clipping (imageblock.cpp:133-173)
    if (block->channel_count() != channel_count()) Throw("ImageBlock::put(): mismatched channel counts!");
    const int ox = block->offset().x - m_offset.x, oy = block->offset().y - m_offset.y;
    for (int y = 0; y < block->size().y; ++y) {
        const int ty = y + oy;
        if (ty < 0 || ty >= m_size.y) continue;
        for (int x = 0; x < block->size().x; ++x) {
            const int tx = x + ox;
            if (tx < 0 || tx >= m_size.x) continue;
            for (size_t c = 0; c < m_channel_count; ++c)
                m_data[((size_t) ty * m_size.x + tx) * m_channel_count + c] += block->data()[((size_t) y * block->size().x + x) * m_channel_count + c];
        }
    }
}

// film.cpp:9-44
Film::Film(const Properties &props) {
    m_size = {props.int_("width", 640), props.int_("height", 320)};
    Vector2i crop_offset{props.int_("crop_offset_x", 0), props.int_("crop_offset_y", 0)};
    Vector2i crop_size{props.int_("crop_width", m_size.x), props.int_("crop_height", m_size.y)};
    if (crop_offset.x < 0 || crop_offset.y < 0 || crop_size.x <= 0 || crop_size.y <= 0 || crop_offset.x + crop_size.x > m_size.x ||
        crop_offset.y + crop_size.y > m_size.y)
        Throw("Invalid crop window specification!\noffset {} + crop size {} vs full size {}", crop_offset.x, crop_size.x, m_size.x);
    m_crop_size = crop_size; m_crop_offset = crop_offset;
    for (auto &kv : props.objects()) {
        auto *rf = dynamic_cast<ReconstructionFilter *>(kv.second.get());
        if (!rf) Throw("Tried to add an unsupported component of type {}", kv.second->to_string());
        if (m_filter) Throw("A film can only have one filter");
        m_filter = rf;
    }
    if (!m_filter) m_filter = InstanceManager::get()->create_instance<ReconstructionFilter>(Properties("gaussian"));
}

Sampler::Sampler(const Properties &props) {        // sampler.cpp:7-10
    m_sample_count = (size_t) props.int_("sample_count", 1);
    m_base_seed = (uint64_t) props.int_("base_seed", 0);
}

void Emitter::set_shape(Shape *shape) {
    if (m_shape) Throw("An emitter can be only be attached to a single shape.");
    m_shape = shape;
}

// shape.cpp:14-57
Shape::Shape(const Properties &props) : m_id(props.id()) {
    for (auto &kv : props.objects()) {
        auto *emitter = dynamic_cast<Emitter *>(kv.second.get());
        auto *bsdf = dynamic_cast<BSDF *>(kv.second.get());
        if (emitter) {
            if (m_emitter) Throw("Only one light can be specified by a shape.");
            m_emitter = emitter;
        } else if (bsdf) {
            if (m_bsdf) Throw("Only one bsdf can be specified by a shape.");
            m_bsdf = bsdf;
        } else {
            Throw("Tired to add unsuppored object of type \"{}\"", kv.second->to_string());
        }
    }
    if (!m_bsdf) m_bsdf = InstanceManager::get()->create_instance<BSDF>(Properties("diffuse"));
}
void Shape::set_children() { if (m_emitter) m_emitter->set_shape(this); }

Mesh::Mesh(const Properties &props) : Shape(props) {
    m_to_world = props.transform("to_world", Transform4f());
    set_children();
}

// sensor.cpp:9-45
Sensor::Sensor(const Properties &props) {
    m_world_transform = props.transform("to_world", Transform4f());
    for (auto &kv : props.objects()) {
        auto *film = dynamic_cast<Film *>(kv.second.get());
        auto *sampler = dynamic_cast<Sampler *>(kv.second.get());
        if (film) { if (m_film) Throw("Camera can only have one film."); m_film = film; }
        else if (sampler) { if (m_sampler) Throw("Can only have one samplelr."); m_sampler = sampler; }
    }
    if (!m_film) m_film = InstanceManager::get()->create_instance<Film>(Properties("rgbfilm"));
    if (!m_sampler) m_sampler = InstanceManager::get()->create_instance<Sampler>(Properties("independent"));
    m_aspect = m_film->size().x / (float) m_film->size().y;
}
ProjectiveCamera::ProjectiveCamera(const Properties &props) : Sensor(props) {   // sensor.cpp:136-141
    m_near_clip = props.float_("near_clip", 1e-2f);
    m_far_clip = props.float_("far_clip", 1e4f);
    m_focus_distance = props.float_("focus_distance", m_far_clip);
}

SamplingIntegrator::SamplingIntegrator(const Properties &props) : Integrator(props) {   // integrator.cpp:18-24
    m_block_size = (uint32_t) props.int_("block_size", 32);
    m_hide_emitters = props.bool_("hide_emitters", false);
}
MonteCarloIntegrator::MonteCarloIntegrator(const Properties &props) : SamplingIntegrator(props) {   // integrator.cpp:128-137
    m_rr_depth = props.int_("rr_depth", 5);
    if (m_rr_depth <= 0) Throw("\"rr_depth\" must be set to a value greater than zero!");
    m_max_depth = props.int_("max_depth", -1);
    if (m_max_depth < 0 && m_max_depth != -1) Throw("\"max_depth\" must be set to -1 (infinite) or a value >= 0");
}

// scene.cpp:26-64
Scene::Scene(const Properties &props) {
    for (auto &kv : props.objects()) {
        Object *obj = kv.second.get();
        if (auto *shape = dynamic_cast<Shape *>(obj)) {
            if (shape->is_emitter()) m_emitters.emplace_back(const_cast<Emitter *>(shape->emitter()));
            m_shapes.push_back(shape);
        } else if (auto *emitter = dynamic_cast<Emitter *>(obj)) {
            if (!emitter->is_surface()) m_emitters.emplace_back(emitter);
            if (emitter->is_environment()) {
                if (m_environment) Throw("Can only have one environment light");
                m_environment = emitter;
            }
        } else if (auto *sensor = dynamic_cast<Sensor *>(obj)) {
            if (m_sensor) Throw("Can only have one camera.");
            m_sensor = sensor;
        } else if (auto *integrator = dynamic_cast<Integrator *>(obj)) {
            if (m_integrator) Throw("Can only have one integrator.");
            m_integrator = integrator;
        }
    }
    if (!m_integrator) {
        Log(Warn, "No integrator found! Instantiating a path tracer..");
        m_integrator = InstanceManager::get()->create_instance<Integrator>(Properties("path"));
    }
}

MSK_REGISTER_INSTANCE(Scene, "scene")      // <scene> is instantiated like any plugin (xml.cpp:402-403)

// =========================================================================== spectra plugins
// spectra/srgb.cpp:13-23
class SRGBReflectanceSpectrum final : public Texture {
public:
    SRGBReflectanceSpectrum(const Properties &props) : Texture(props) { m_value = srgb_model_fetch(props.color("color")); }
    bool flatten(Flat &out) const override { out.coeff[0] = m_value.r; out.coeff[1] = m_value.g; out.coeff[2] = m_value.b; out.uses_d65 = false; return true; }
    MSK_DECLARE_CLASS()
private:
    Color3 m_value;
};
MSK_IMPLEMENT_CLASS(SRGBReflectanceSpectrum, Texture)
MSK_REGISTER_INSTANCE(SRGBReflectanceSpectrum, "srgb")

// spectra/uniform.cpp:12-27: the same value at every wavelength (what <spectrum value="c"/> creates outside an emitter)
class UniformSpectrum final : public Texture {
public:
    UniformSpectrum(const Properties &props) : Texture(props) { m_value = props.float_("value"); }
    bool flatten(Flat &out) const override { out.coeff[0] = out.coeff[1] = 0.f; out.coeff[2] = INFINITY; out.scale = m_value; out.uses_d65 = false; return true; }
    float mean() const override { return m_value; }
    MSK_DECLARE_CLASS()
private:
    float m_value;
};
MSK_IMPLEMENT_CLASS(UniformSpectrum, Texture)
MSK_REGISTER_INSTANCE(UniformSpectrum, "uniform")

// spectra/regular.cpp:27-91,148: a spectrum tabulated on a regular wavelength grid (what <spectrum value="l0:v0, l1:v1, ..."/>
// with equidistant wavelengths makes, xml.cpp:300-341).  Same properties: size, lambda_min, lambda_max, values (a pointer to
// `size` floats, read here); same checks and messages as SpectrumContinuousDistribution::update (regular.cpp:27-70).
class RegularSpectrum final : public Texture {
public:
    RegularSpectrum(const Properties &props) : Texture(props) {
        m_lambda_min = props.float_("lambda_min"); m_lambda_max = props.float_("lambda_max");
        const int size = props.int_("size");
        const float *values = (const float *) props.pointer("values");
        if (size < 2) Throw("ContinuousDistribution: needs at least two entries!");
        if (!(m_lambda_min < m_lambda_max)) Throw("ContinuousDistribution: invalid range!");
        m_values.assign(values, values + size);
        const double interval_size = ((double) m_lambda_max - (double) m_lambda_min) / (size - 1);
        double integral = 0.;
        bool mass = false;
        for (int i = 0; i + 1 < size; ++i) {
            const double y0 = m_values[i], y1 = m_values[i + 1], value = 0.5 * interval_size * (y0 + y1);
            integral += value;
            if (y0 < 0. || y1 < 0.) Throw("ContinuousDistribution: entries must be non-negative!");
            mass |= value > 0.;
        }
        if (!mass) Throw("ContinuousDistribution: no probability mass found!");
        m_integral = (float) integral;
    }
    bool flatten(Flat &out) const override {
        out.regular = true; out.uses_d65 = false; out.lambda_min = m_lambda_min; out.lambda_max = m_lambda_max; out.values = m_values;
        return true;
    }
    float mean() const override { return m_integral; }              // regular.cpp:150 returns the integral
    MSK_DECLARE_CLASS()
private:
    float m_lambda_min, m_lambda_max, m_integral = 0.f;
    std::vector<float> m_values;
};
MSK_IMPLEMENT_CLASS(RegularSpectrum, Texture)
MSK_REGISTER_INSTANCE(RegularSpectrum, "regular")

// srgb with the normalisation of spectra/srgb_d65.cpp:18-22 but no illuminant: value = scale * S(fetch(rgb / scale))
// for colours above 1 (conductor eta / k given as <rgb>), plain srgb otherwise
class SRGBUnboundedSpectrum final : public Texture {
public:
    SRGBUnboundedSpectrum(const Properties &props) : Texture(props) {
        Color3 c = props.color("color");
        const float mx = std::max(c.r, std::max(c.g, c.b));
        if (mx > 1.f) { m_scale = mx * 2.f; c.r /= m_scale; c.g /= m_scale; c.b /= m_scale; }
        m_value = srgb_model_fetch(c);
    }
    bool flatten(Flat &out) const override { out.coeff[0] = m_value.r; out.coeff[1] = m_value.g; out.coeff[2] = m_value.b; out.scale = m_scale; out.uses_d65 = false; return true; }
    MSK_DECLARE_CLASS()
private:
    Color3 m_value;
    float m_scale = 1.f;
};
MSK_IMPLEMENT_CLASS(SRGBUnboundedSpectrum, Texture)
MSK_REGISTER_INSTANCE(SRGBUnboundedSpectrum, "srgb_unbounded")

// spectra/d65.cpp:29-47: a D65 table scaled by scale / 10568
class D65Spectrum final : public Texture {
public:
    D65Spectrum(const Properties &props) : Texture(props) { m_scale = props.float_("scale", 1.f); m_scale *= 1.f / 10568.f; }
    bool flatten(Flat &out) const override { out.coeff[0] = out.coeff[1] = 0.f; out.coeff[2] = INFINITY; out.d65_scale = m_scale; out.uses_d65 = true; return true; }
    float scale() const { return m_scale; }
    MSK_DECLARE_CLASS()
private:
    float m_scale;
};
MSK_IMPLEMENT_CLASS(D65Spectrum, Texture)
MSK_REGISTER_INSTANCE(D65Spectrum, "d65")
ref<Texture> Texture::D65(float scale) { Properties p("d65"); p.set_float("scale", scale); return InstanceManager::get()->create_instance<Texture>(p); }

// spectra/srgb_d65.cpp:13-36
class SRGBEmitterSpectrum final : public Texture {
public:
    SRGBEmitterSpectrum(const Properties &props) : Texture(props) {
        Color3 color = props.color("color");
        float scale = std::max(color.r, std::max(color.g, color.b)) * 2.f;
        if (scale != 0.f) { color.r /= scale; color.g /= scale; color.b /= scale; }
        m_value = srgb_model_fetch(color);
        Properties p2("d65");
        p2.set_float("scale", props.float_("scale", 1.f) * scale);
        m_d65 = InstanceManager::get()->create_instance<Texture>(p2);
    }
    bool flatten(Flat &out) const override {
        Flat d; m_d65->flatten(d);
        out.coeff[0] = m_value.r; out.coeff[1] = m_value.g; out.coeff[2] = m_value.b; out.d65_scale = d.d65_scale; out.uses_d65 = true;
        return true;
    }
    MSK_DECLARE_CLASS()
private:
    Color3 m_value;
    ref<Texture> m_d65;
};
MSK_IMPLEMENT_CLASS(SRGBEmitterSpectrum, Texture)
MSK_REGISTER_INSTANCE(SRGBEmitterSpectrum, "srgb_d65")

// textures/checkerboard.cpp:10-15 — same plugin name and properties (color0 default .4, color1 default .2, to_uv);
// evaluated on the MI355X from its flat description
class CheckerboardTexture final : public Texture {
public:
    CheckerboardTexture(const Properties &props) : Texture(props) {
        m_color0 = props.texture("color0", .4f);
        m_color1 = props.texture("color1", .2f);
        m_to_uv = props.transform("to_uv", Transform4f());
    }
    bool flatten_texture(msk_texture_desc &out) const override {
        Flat c0, c1;
        if (!m_color0->flatten(c0) || !m_color1->flatten(c1) || c0.uses_d65 || c1.uses_d65 || c0.regular || c1.regular || c0.scale != 1.f || c1.scale != 1.f) return false;
        std::memset(&out, 0, sizeof out);
        out.type = MSK_TEXTURE_CHECKERBOARD;
        std::memcpy(out.color0, c0.coeff, sizeof c0.coeff);
        std::memcpy(out.color1, c1.coeff, sizeof c1.coeff);
        // Transform4f::extract (core/transform.h:142-148): the top-left 3x3 acts on (u, v, 1)
        for (int r = 0; r < 2; ++r) for (int c = 0; c < 3; ++c) out.to_uv[r * 3 + c] = (float) m_to_uv.matrix().m[r][c];
        return true;
    }
    float mean() const override { return m_color0->mean() + m_color1->mean(); }   // checkerboard.cpp:46 (a sum, as there)
    MSK_DECLARE_CLASS()
private:
    ref<Texture> m_color0, m_color1;
    Transform4f m_to_uv;
};
MSK_IMPLEMENT_CLASS(CheckerboardTexture, Texture)
MSK_REGISTER_INSTANCE(CheckerboardTexture, "checkerboard")

// =========================================================================== filter, sampler, film
// filters/gaussian.cpp:10-20
class GaussianFilter final : public ReconstructionFilter {
public:
    GaussianFilter(const Properties &props) : ReconstructionFilter(props) {
        m_stddev = props.float_("stddev", 0.5f);
        m_radius = 4 * m_stddev;
        m_alpha = -1.f / (2.f * m_stddev * m_stddev);
        m_bias = std::exp(m_alpha * m_radius * m_radius);
        init_discretization();
    }
    float eval(float x) const override { return std::max(0.f, std::exp(m_alpha * x * x) - m_bias); }
    MSK_DECLARE_CLASS()
private:
    float m_stddev, m_alpha, m_bias;
};
MSK_IMPLEMENT_CLASS(GaussianFilter, ReconstructionFilter)
MSK_REGISTER_INSTANCE(GaussianFilter, "gaussian")

// samplers/independent.cpp — only the parameters matter on this path: draws happen on the device
class IndependentSampler final : public Sampler {
public:
    IndependentSampler(const Properties &props) : Sampler(props) {}
    MSK_DECLARE_CLASS()
};
MSK_IMPLEMENT_CLASS(IndependentSampler, Sampler)
MSK_REGISTER_INSTANCE(IndependentSampler, "independent")

// films/hdrfilm.cpp:14-112
class HDRFilm : public Film {
public:
    HDRFilm(const Properties &props) : Film(props) {
        m_file_format = string::to_lower(props.string("file_format", "openexr"));
        m_dest_file = props.string("filename", "");
    }
    void set_destination_file(const std::string &f) override { m_dest_file = f; }
    void prepare(const std::vector<std::string> &channels) override {
        for (size_t i = 1; i < channels.size(); ++i)
            if (channels[i] == channels[i - 1]) Throw("Film::prepare(): duplicate channel name \"{}\"", channels[i]);
        m_storage = new ImageBlock(m_crop_size, channels.size());
        m_storage->set_offset(m_crop_offset);
        m_storage->clear();
        m_channels = channels;
    }
    void put(const ImageBlock *block) override { m_storage->put(block); }
    const ImageBlock *storage() const override { return m_storage.get(); }
    std::vector<float> image() override {              // hdrfilm.cpp:48-90: xyz_to_srgb / weight
        if (!m_storage) Throw("HDRFilm::image(): the film was never prepared");
        const int w = m_storage->size().x, h = m_storage->size().y;
        const size_t cc = m_channels.size();
        const size_t oc = cc - 1;                        // R,G,B,A + the AOV channels (hdrfilm.cpp:52-59)
        std::vector<float> out((size_t) w * h * oc);
        static const float M[9] = {3.240479f, -1.537150f, -0.498535f, -0.969256f, 1.875991f, 0.041556f, 0.055648f, -0.204043f, 1.057311f};
        for (int y = 0; y < h; ++y)
            for (int x = 0; x < w; ++x) {
                const float *p = &m_storage->data()[((size_t) y * w + x) * cc];
                const float weight = p[4], inv = weight != 0 ? 1.f / weight : 0.f;
                float *o = &out[((size_t) y * w + x) * oc];
                for (int r = 0; r < 3; ++r) o[r] = (M[r * 3] * p[0] + (M[r * 3 + 1] * p[1] + M[r * 3 + 2] * p[2])) * inv;
                o[3] = p[3] * inv;
                for (size_t ch = 5; ch < cc; ++ch) o[ch - 1] = p[ch] * inv;       // hdrfilm.cpp:82-85
            }
        return out;
    }
    std::vector<std::string> image_channels() const override {
        std::vector<std::string> names = {"R", "G", "B", "A"};
        for (size_t i = 5; i < m_channels.size(); ++i) names.push_back(m_channels[i]);
        return names;
    }
    void develop() override {                           // hdrfilm.cpp:92-112
        if (m_dest_file.empty()) Throw("Destination file not specified, cannot develop.");
        std::string ext = m_file_format == "openexr" ? ".exr" : m_file_format == "rgbe" ? ".rgbe" : ".pfm";
        std::string filename = m_dest_file;
        size_t dot = filename.find_last_of('.'), slash = filename.find_last_of('/');
        if (dot != std::string::npos && (slash == std::string::npos || dot > slash)) filename = filename.substr(0, dot);
        filename += ext;
        Log(Info, "Developing \"{}\" ..", filename);
        std::vector<float> img = image();
        const int w = m_storage->size().x, h = m_storage->size().y;
        const size_t oc = m_channels.size() - 1;
        if (ext == ".exr") write_exr(filename, w, h, image_channels(), img.data());
        else if (ext == ".pfm") {
            std::vector<float> rgb((size_t) w * h * 3);
            for (size_t i = 0; i < (size_t) w * h; ++i) for (int c = 0; c < 3; ++c) rgb[i * 3 + c] = img[i * oc + c];
            write_pfm(filename, w, h, 3, rgb.data());
        } else Throw("file_format \"{}\" is not supported by this build (use openexr or pfm)", m_file_format);
    }
    MSK_DECLARE_CLASS()
protected:
    std::string m_file_format, m_dest_file;
    ref<ImageBlock> m_storage;
    std::vector<std::string> m_channels;
};
MSK_IMPLEMENT_CLASS(HDRFilm, Film)
MSK_REGISTER_INSTANCE(HDRFilm, "hdrfilm")
// assets/cbox/scene.xml:15 asks for "rgbfilm", which the reference does not compile: alias
class RGBFilm final : public HDRFilm { public: RGBFilm(const Properties &p) : HDRFilm(p) {} MSK_DECLARE_CLASS() };
MSK_IMPLEMENT_CLASS(RGBFilm, HDRFilm)
MSK_REGISTER_INSTANCE(RGBFilm, "rgbfilm")

static void init_bsdf_desc(msk_bsdf_desc &out) {
    std::memset(&out, 0, sizeof out);
    out.back_bsdf = -1;
    const msk_spectrum_desc one{{0.f, 0.f, INFINITY}, 1.f};
    out.eta = out.k = out.specular_reflectance = out.specular_transmittance = one;
    out.ior_eta = out.ior_inv_eta = 1.f;
    out.reflectance_scale = 1.f;
}

// =========================================================================== bsdf, emitter, sensor, shape
// bsdfs/diffuse.cpp:12-16
class SmoothDiffuse final : public BSDF {
public:
    SmoothDiffuse(const Properties &props) : BSDF(props) { m_reflectance = props.texture("reflectance", 0.5f); }
    bool flatten(msk_bsdf_desc &out, FlatTables &tables) const override {
        Texture::Flat f;
        msk_texture_desc td;
        init_bsdf_desc(out);
        out.type = MSK_BSDF_DIFFUSE;
        if (m_reflectance->flatten(f) && f.regular) {                // a tabulated reflectance (spectra/regular.cpp)
            out.reflectance_regular = tables.add_regular(f);
        } else if (m_reflectance->flatten(f) && !f.uses_d65) {
            std::memcpy(out.reflectance, f.coeff, sizeof f.coeff);
            out.reflectance_scale = f.scale;
        } else if (m_reflectance->flatten_texture(td)) {            // a reflectance that varies over the surface
            tables.textures.push_back(td);
            out.reflectance_texture = (uint32_t) tables.textures.size();
        } else {
            return false;
        }
        return true;
    }
    MSK_DECLARE_CLASS()
private:
    ref<Texture> m_reflectance;
};
MSK_IMPLEMENT_CLASS(SmoothDiffuse, BSDF)
MSK_REGISTER_INSTANCE(SmoothDiffuse, "diffuse")

// bsdfs/roughconductor.cpp:12-50.  Adaptation (DESIGN.md §rough conductor; the reference's version is not
// compiled and typed against RGB): GGX only — "beckmann", the reference's default, evaluates to
// zero there (microfacet.h:113-115) and is rejected here; alpha is a float property; eta / k /
// specular_reflectance are spectra.  An <rgb> above 1 (metal eta, k) is loaded as an srgb_unbounded texture.
class RoughConductor final : public BSDF {
public:
    RoughConductor(const Properties &props) : BSDF(props) {
        if (!props.has_property("eta") || !props.has_property("k"))
            Throw("roughconductor: both \"eta\" and \"k\" must be specified");
        m_eta = unbounded(props, "eta"); m_k = unbounded(props, "k");
        const std::string distr = props.string("distribution", "ggx");
        if (distr != "ggx") {
            if (distr == "beckmann") Throw("roughconductor: the \"beckmann\" distribution is not implemented (use \"ggx\")");
            Throw("Specified an invalid distribution \"{}\", must be \"beckmann\" or \"ggx\"!", distr);
        }
        m_sample_visible = props.bool_("sample_visible", false);
        if (props.has_property("alpha_u") || props.has_property("alpha_v")) {
            if (!props.has_property("alpha_u") || !props.has_property("alpha_v"))
                Throw("Microfacet model: both 'alpha_u' and 'alpha_v' must be specified.");
            if (props.has_property("alpha")) Throw("Microfacet model: please specifyeither 'alpha' or 'alpha_u'/'alpha_v'.");
            m_alpha_u = props.float_("alpha_u"); m_alpha_v = props.float_("alpha_v");
        } else {
            m_alpha_u = m_alpha_v = props.float_("alpha", 0.1f);
        }
        m_specular_reflectance = props.texture("specular_reflectance", 1.f);
    }
    // props.texture() builds a bounded "srgb" texture from <rgb>; conductors need values above 1
    static ref<Texture> unbounded(const Properties &props, const std::string &name) {
        if (props.type(name) == Properties::Type::Object) return props.texture(name);
        Properties p("srgb_unbounded");
        if (props.type(name) == Properties::Type::Color) p.set_color("color", props.color(name));
        else { const float v = props.float_(name); p.set_color("color", Color3{v, v, v}); }
        return InstanceManager::get()->create_instance<Texture>(p);
    }
    bool flatten(msk_bsdf_desc &out, FlatTables &tables) const override {
        Texture::Flat e, k, s;
        if (!m_eta->flatten(e) || !m_k->flatten(k) || !m_specular_reflectance->flatten(s) || e.uses_d65 || k.uses_d65 || s.uses_d65) return false;
        init_bsdf_desc(out);
        out.type = MSK_BSDF_ROUGHCONDUCTOR;
        out.alpha_u = m_alpha_u; out.alpha_v = m_alpha_v; out.sample_visible = m_sample_visible ? 1 : 0;
        return tables.put(out.eta, e) && tables.put(out.k, k) && tables.put(out.specular_reflectance, s);
    }
    MSK_DECLARE_CLASS()
private:
    ref<Texture> m_eta, m_k, m_specular_reflectance;
    float m_alpha_u, m_alpha_v;
    bool m_sample_visible;
};
MSK_IMPLEMENT_CLASS(RoughConductor, BSDF)
MSK_REGISTER_INSTANCE(RoughConductor, "roughconductor")

// bsdfs/roughdielectric.cpp:12-55.  The reference defaults to the Beckmann distribution here, whose branches are
// empty in render/microfacet.h:115-118,135-137; only "ggx" renders there, and only "ggx" is accepted here.
class RoughDielectric final : public BSDF {
public:
    RoughDielectric(const Properties &props) : BSDF(props) {
        m_specular_reflectance = props.texture("specular_reflectance", 1.f);
        m_specular_transmittance = props.texture("specular_transmittance", 1.f);
        const float int_ior = props.float_("int_ior", 1.5046f), ext_ior = props.float_("ext_ior", 1.00028f);
        if (int_ior < 0.f || ext_ior < 0.f || int_ior == ext_ior)
            Throw("The interior and exterior indices of refraction must be positive and differ!");
        m_eta = int_ior / ext_ior; m_inv_eta = ext_ior / int_ior;
        std::string distr = props.string("distribution", "beckmann");
        for (auto &c : distr) c = (char) std::tolower((unsigned char) c);
        if (distr == "beckmann") Throw("roughdielectric: the \"beckmann\" distribution is not implemented (use \"ggx\")");
        if (distr != "ggx") Throw("Specified an invalid distribution \"{}\", must be \"beckmann\" or \"ggx\"!", distr);
        m_sample_visible = props.bool_("sample_visible", false);
        if (props.has_property("alpha_u") || props.has_property("alpha_v")) {
            if (!props.has_property("alpha_u") || !props.has_property("alpha_v"))
                Throw("Microfacet model: both 'alpha_u' and 'alpha_v' must be specified.");
            if (props.has_property("alpha")) Throw("Microfacet model: please specifyeither 'alpha' or 'alpha_u'/'alpha_v'.");
            m_alpha_u = props.float_("alpha_u"); m_alpha_v = props.float_("alpha_v");
        } else {
            m_alpha_u = m_alpha_v = props.float_("alpha", 0.1f);
        }
    }
    bool flatten(msk_bsdf_desc &out, FlatTables &tables) const override {
        Texture::Flat r, t;
        if (!m_specular_reflectance->flatten(r) || !m_specular_transmittance->flatten(t) || r.uses_d65 || t.uses_d65) return false;
        init_bsdf_desc(out);
        out.type = MSK_BSDF_ROUGHDIELECTRIC;
        out.alpha_u = m_alpha_u; out.alpha_v = m_alpha_v; out.sample_visible = m_sample_visible ? 1 : 0;
        out.ior_eta = m_eta; out.ior_inv_eta = m_inv_eta;
        return tables.put(out.specular_reflectance, r) && tables.put(out.specular_transmittance, t);
    }
    MSK_DECLARE_CLASS()
private:
    ref<Texture> m_specular_reflectance, m_specular_transmittance;
    float m_eta, m_inv_eta, m_alpha_u, m_alpha_v;
    bool m_sample_visible;
};
MSK_IMPLEMENT_CLASS(RoughDielectric, BSDF)
MSK_REGISTER_INSTANCE(RoughDielectric, "roughdielectric")

// bsdfs/twosided.cpp:12-36
class TwoSidedBRDF final : public BSDF {
public:
    TwoSidedBRDF(const Properties &props) : BSDF(props) {
        auto bsdfs = props.objects();
        for (auto &kv : bsdfs) if (!dynamic_cast<BSDF *>(kv.second.get())) Throw("twosided: nested object \"{}\" is not a BSDF", kv.first);
        if (!bsdfs.empty()) m_brdf[0] = static_cast<BSDF *>(bsdfs[0].second.get());
        if (bsdfs.size() == 2) m_brdf[1] = static_cast<BSDF *>(bsdfs[1].second.get());
        else if (bsdfs.size() > 2) Throw("At most two nested BSDFs can be specified!");
        if (!m_brdf[0]) Throw("A nested one-sided material is required!");
        if (!m_brdf[1]) m_brdf[1] = m_brdf[0];
    }
    const BSDF *nested(int i) const override { return m_brdf[i].get(); }
    bool flatten(msk_bsdf_desc &out, FlatTables &tables) const override { return m_brdf[0]->flatten(out, tables); }   // front side; flatten_scene adds the back
    MSK_DECLARE_CLASS()
private:
    ref<BSDF> m_brdf[2];
};
MSK_IMPLEMENT_CLASS(TwoSidedBRDF, BSDF)
MSK_REGISTER_INSTANCE(TwoSidedBRDF, "twosided")

// emitters/area.cpp:13-18
class AreaLight final : public Emitter {
public:
    AreaLight(const Properties &props) : Emitter(props) { m_radiance = props.texture("radiance", Texture::D65(1.f)); }
    bool is_surface() const override { return true; }
    bool flatten(msk_emitter_desc &out, FlatTables &tables) const override {
        Texture::Flat f;
        if (!m_radiance->flatten(f) || !(f.uses_d65 || f.regular)) return false;
        std::memset(&out, 0, sizeof out);
        out.type = MSK_EMITTER_AREA;
        if (f.regular) { out.radiance_regular = tables.add_regular(f); return true; }      // a `regular` radiance as it stands
        std::memcpy(out.radiance, f.coeff, sizeof f.coeff);
        out.d65_scale = f.d65_scale;
        return true;
    }
    MSK_DECLARE_CLASS()
private:
    ref<Texture> m_radiance;
};
MSK_IMPLEMENT_CLASS(AreaLight, Emitter)
MSK_REGISTER_INSTANCE(AreaLight, "area")

// emitters/constant.cpp:12-19.  set_scene()'s bounding sphere is derived by the back end from the uploaded vertices.
class ConstantBackgroundEmitter final : public Emitter {
public:
    ConstantBackgroundEmitter(const Properties &props) : Emitter(props) { m_radiance = props.texture("radiance", Texture::D65(1.f)); }
    bool is_environment() const override { return true; }
    bool flatten(msk_emitter_desc &out, FlatTables &tables) const override {
        Texture::Flat f;
        if (!m_radiance->flatten(f) || !(f.uses_d65 || f.regular)) return false;
        std::memset(&out, 0, sizeof out);
        out.type = MSK_EMITTER_CONSTANT;
        out.mesh_id = -1;
        if (f.regular) { out.radiance_regular = tables.add_regular(f); return true; }      // a `regular` radiance as it stands
        std::memcpy(out.radiance, f.coeff, sizeof f.coeff);
        out.d65_scale = f.d65_scale;
        return true;
    }
    MSK_DECLARE_CLASS()
private:
    ref<Texture> m_radiance;
};
MSK_IMPLEMENT_CLASS(ConstantBackgroundEmitter, Emitter)
MSK_REGISTER_INSTANCE(ConstantBackgroundEmitter, "constant")

// sensors/perspective.cpp:8-42
class PerspectiveCamera final : public ProjectiveCamera {
public:
    PerspectiveCamera(const Properties &props) : ProjectiveCamera(props) {
        m_fov = props.float_("fov", 30);
        m_camera_to_sample = Transform4f::scale(Vector3f{(float) m_film->size().x, (float) m_film->size().y, 1.f}) *
                             Transform4f::scale(Vector3f{-0.5f, -0.5f * m_aspect, 1.f}) *
                             Transform4f::translate(Vector3f{-1.f, -1.f / m_aspect, 0.f}) *
                             Transform4f::perspective(m_fov, m_near_clip, m_far_clip);
        m_sample_to_camera = m_camera_to_sample.inverse();
    }
    bool flatten(msk_camera_desc &out) const override {
        m_sample_to_camera.to_float16(out.sample_to_camera);
        m_world_transform.to_float16(out.to_world);
        out.near_clip = m_near_clip; out.far_clip = m_far_clip;
        return true;
    }
    MSK_DECLARE_CLASS()
private:
    Transform4f m_camera_to_sample, m_sample_to_camera;
    float m_fov;
};
MSK_IMPLEMENT_CLASS(PerspectiveCamera, ProjectiveCamera)
MSK_REGISTER_INSTANCE(PerspectiveCamera, "perspective")

// shapes/obj.cpp:58-181
class OBJMesh final : public Mesh {
    struct OBJVertex {
        int p = -1, n = -1, uv = -1;
        bool operator==(const OBJVertex &o) const { return p == o.p && n == o.n && uv == o.uv; }
    };
    struct Hash { size_t operator()(const OBJVertex &v) const { return (size_t) v.p * 73856093u ^ (size_t) (v.n + 1) * 19349663u ^ (size_t) (v.uv + 1) * 83492791u; } };
public:
    OBJMesh(const Properties &props) : Mesh(props) {
        const bool flip_tex_coords = props.bool_("filp_tex_coords", true);   // sic: the reference's key (obj.cpp:59)
        const std::string path = get_file_resolver()->resolve(props.string("filename"));
        size_t slash = path.find_last_of('/');
        m_name = slash == std::string::npos ? path : path.substr(slash + 1);
        // The reference parses with one istringstream per line (obj.cpp:83-135); same grammar here over the whole file in
        // memory with strtof / a hand-rolled index parser (about 10x the throughput on a 70 k-triangle mesh).
        std::string text;
        {
            std::ifstream is(path, std::ios::binary);
            if (!is) Throw("Error while loading OBJ file \"{}\": file not found", m_name);
            is.seekg(0, std::ios::end);
            text.resize((size_t) is.tellg());
            is.seekg(0);
            is.read(&text[0], (std::streamsize) text.size());
        }
        Log(Info, "Loading mesh from \"{}\"", m_name);
        std::vector<Vector3f> positions, normals;
        std::vector<std::array<float, 2>> texcoords;
        std::vector<uint32_t> triangles;
        std::vector<OBJVertex> obj_vertices;
        std::unordered_map<OBJVertex, uint32_t, Hash> vertex_map;
        vertex_map.reserve(text.size() / 48);
        const char *p = text.c_str(), *end = p + text.size();
        auto skip_ws = [&](const char *q, const char *le) { while (q < le && (*q == ' ' || *q == '\t' || *q == '\r')) ++q; return q; };
        auto next_float = [&](const char *&q, const char *le, float &out) {       // like `line >> f`: leaves `out` alone on failure
            q = skip_ws(q, le);
            if (q >= le) return false;
            char *e = nullptr;
            const float v = std::strtof(q, &e);
            if (e == q || e > le) return false;
            out = v; q = e;
            return true;
        };
        auto next_token = [&](const char *&q, const char *le, const char *&tb, const char *&te) {
            q = skip_ws(q, le);
            tb = q;
            while (q < le && *q != ' ' && *q != '\t' && *q != '\r') ++q;
            te = q;
            return te > tb;
        };
        auto parse_index = [&](const char *&q, const char *te, int &out) {         // optional sign + digits; false if none
            const char *b = q; bool neg = false;
            if (q < te && (*q == '-' || *q == '+')) { neg = *q == '-'; ++q; }
            long v = 0; const char *d0 = q;
            while (q < te && *q >= '0' && *q <= '9') { v = v * 10 + (*q - '0'); ++q; }
            if (q == d0) { q = b; return false; }
            out = (int) (neg ? -v : v);
            return true;
        };
        auto parse_vertex_at = [&](const char *tb, const char *te) {               // "p", "p/uv", "p//n", "p/uv/n"
            OBJVertex v;
            const char *q = tb;
            if (!parse_index(q, te, v.p)) Throw("Error while loading OBJ file \"{}\": malformed face vertex \"{}\"", m_name, std::string(tb, te));
            if (q < te && *q == '/') {
                ++q;
                parse_index(q, te, v.uv);
                if (q < te && *q == '/') { ++q; parse_index(q, te, v.n); }
            }
            return v;
        };
        while (p < end) {
            const char *le = (const char *) std::memchr(p, '\n', (size_t) (end - p));
            if (!le) le = end;
            const char *q = p, *tb, *te;
            if (next_token(q, le, tb, te)) {
                const size_t tl = (size_t) (te - tb);
                if (tl == 1 && tb[0] == 'v') {
                    Vector3f v{0.f, 0.f, 0.f};
                    next_float(q, le, v.x) && next_float(q, le, v.y) && next_float(q, le, v.z);
                    positions.push_back(m_to_world.apply_point(v));                       // obj.cpp:90
                } else if (tl == 2 && tb[0] == 'v' && tb[1] == 't') {
                    std::array<float, 2> tc{0, 0};
                    next_float(q, le, tc[0]) && next_float(q, le, tc[1]);
                    if (flip_tex_coords) tc[1] = 1.f - tc[1];
                    texcoords.push_back(tc);
                } else if (tl == 2 && tb[0] == 'v' && tb[1] == 'n') {
                    Vector3f n{0.f, 0.f, 0.f};
                    next_float(q, le, n.x) && next_float(q, le, n.y) && next_float(q, le, n.z);
                    n = m_to_world.apply_normal(n);
                    const float z = n.x * n.x + (n.y * n.y + n.z * n.z);
                    if (z > 0) { const float l = std::sqrt(z); n = Vector3f{n.x / l, n.y / l, n.z / l}; }
                    normals.push_back(n);
                } else if (tl == 1 && tb[0] == 'f') {
                    OBJVertex verts[6]; int nv = 0;
                    const char *vb, *ve;
                    while (nv < 4 && next_token(q, le, vb, ve)) verts[nv++] = parse_vertex_at(vb, ve);
                    if (nv < 3) Throw("Error while loading OBJ file \"{}\": face with fewer than three vertices", m_name);
                    int n_vertices = 3;
                    if (nv == 4) { verts[4] = verts[0]; verts[5] = verts[2]; n_vertices = 6; }   // quad -> (v0,v1,v2), (v3,v0,v2)  (obj.cpp:109-119)
                    for (int i = 0; i < n_vertices; ++i) {
                        auto it = vertex_map.find(verts[i]);
                        if (it == vertex_map.end()) {
                            vertex_map[verts[i]] = (uint32_t) obj_vertices.size();
                            triangles.push_back((uint32_t) obj_vertices.size());
                            obj_vertices.push_back(verts[i]);
                        } else triangles.push_back(it->second);
                    }
                }
            }
            p = le < end ? le + 1 : end;
        }
        m_vertex_count = (uint32_t) obj_vertices.size();
        m_face_count = (uint32_t) (triangles.size() / 3);
        m_normal_offset = normals.empty() ? 0 : 3;
        m_texcoord_offset = texcoords.empty() ? 0 : 6;
        m_faces = triangles;
        m_vertices.assign((size_t) m_vertex_count * 8, 0.f);
        for (size_t i = 0; i < obj_vertices.size(); ++i) {
            const OBJVertex &v = obj_vertices[i];
            if (v.p < 1 || (size_t) v.p > positions.size()) Throw("Error while loading OBJ file \"{}\": vertex index {} out of range", m_name, v.p);
            float *o = &m_vertices[i * 8];
            o[0] = positions[v.p - 1].x; o[1] = positions[v.p - 1].y; o[2] = positions[v.p - 1].z;
            if (v.n != -1) {
                if ((size_t) v.n > normals.size() || v.n < 1) Throw("Error while loading OBJ file \"{}\": normal index {} out of range", m_name, v.n);
                o[3] = normals[v.n - 1].x; o[4] = normals[v.n - 1].y; o[5] = normals[v.n - 1].z;
            }
            if (v.uv != -1) {
                if ((size_t) v.uv > texcoords.size() || v.uv < 1) Throw("Error while loading OBJ file \"{}\": texcoord index {} out of range", m_name, v.uv);
                o[6] = texcoords[v.uv - 1][0]; o[7] = texcoords[v.uv - 1][1];
            }
        }
        Log(Info, "\"{}\": read {} faces, {} vertices", m_name, m_face_count, m_vertex_count);
    }
    MSK_DECLARE_CLASS()
};
MSK_IMPLEMENT_CLASS(OBJMesh, Mesh)
MSK_REGISTER_INSTANCE(OBJMesh, "obj")

// =========================================================================== flatten
void flatten_scene(const Scene *scene, const Sensor *sensor, FlatScene &out) {
    out.meshes.clear(); out.bsdfs.clear(); out.emitters.clear(); out.textures.clear(); out.vertices.clear(); out.faces.clear();
    out.regular.clear(); out.regular_values.clear();
    FlatTables tables{out.textures, out.regular, out.regular_values};
    // Scene::m_emitters order (scene.cpp:27-41) decides which emitter sample_emitter_direct picks (scene.cpp:80-84)
    std::map<const Emitter *, int> emitter_index;
    for (auto &e : scene->emitters()) {
        msk_emitter_desc ed;
        if (!e->flatten(ed, tables))
            Throw("Emitter \"{}\" cannot be evaluated by the GPU path integrator", e->clazz()->name());
        if (!e->is_surface() && !e->is_environment())
            Throw("Emitter \"{}\" is not attached to a shape: not supported by the GPU path integrator", e->clazz()->name());
        emitter_index[e.get()] = (int) out.emitters.size();
        out.emitters.push_back(ed);
    }
    uint32_t nv = 0, nf = 0;
    for (size_t i = 0; i < scene->shapes().size(); ++i) {
        const Shape *shape = scene->shapes()[i].get();
        if (!shape->is_mesh()) Throw("Shape {} (\"{}\") is not a triangle mesh: not supported by the GPU path integrator", i, shape->id());
        const Mesh *mesh = static_cast<const Mesh *>(shape);
        msk_bsdf_desc bd;
        if (!shape->bsdf()->flatten(bd, tables))
            Throw("BSDF \"{}\" of shape {} cannot be evaluated by the GPU path integrator", shape->bsdf()->clazz()->name(), i);
        if (const BSDF *back = shape->bsdf()->nested(1)) {          // twosided adapter (twosided.cpp:38-101)
            if (back == shape->bsdf()->nested(0)) {
                bd.back_bsdf = (int32_t) out.bsdfs.size();
            } else {
                msk_bsdf_desc bb;
                if (!back->flatten(bb, tables)) Throw("BSDF \"{}\" of shape {} cannot be evaluated by the GPU path integrator", back->clazz()->name(), i);
                out.bsdfs.push_back(bb);
                bd.back_bsdf = (int32_t) out.bsdfs.size() - 1;
            }
        }
        out.bsdfs.push_back(bd);
        int eid = -1;
        if (shape->is_emitter()) {
            eid = emitter_index.at(shape->emitter());
            out.emitters[eid].mesh_id = (int32_t) i;
        }
        msk_mesh_desc md{nv, mesh->vertex_count(), nf, mesh->face_count(), (int32_t) (out.bsdfs.size() - 1), eid,
                         mesh->has_vertex_normals() ? 1u : 0u, mesh->has_vertex_texcoords() ? 1u : 0u};
        out.meshes.push_back(md);
        out.vertices.insert(out.vertices.end(), mesh->vertices(), mesh->vertices() + (size_t) mesh->vertex_count() * 8);
        out.faces.insert(out.faces.end(), mesh->faces(), mesh->faces() + (size_t) mesh->face_count() * 3);
        nv += mesh->vertex_count(); nf += mesh->face_count();
    }
    msk_scene_desc &d = out.desc;
    std::memset(&d, 0, sizeof d);
    d.abi_version = MSK_ABI_VERSION;
    d.n_meshes = (uint32_t) out.meshes.size(); d.n_bsdfs = (uint32_t) out.bsdfs.size(); d.n_emitters = (uint32_t) out.emitters.size();
    d.meshes = out.meshes.data(); d.bsdfs = out.bsdfs.data(); d.emitters = out.emitters.data();
    d.vertices = out.vertices.data(); d.faces = out.faces.data(); d.n_vertices = nv; d.n_faces = nf;
    if (!sensor->flatten(d.camera)) Throw("Sensor \"{}\" is not supported by the GPU path integrator", sensor->clazz()->name());
    const Film *film = sensor->film();
    d.film.width = film->size().x; d.film.height = film->size().y;
    // film.cpp:12-21: the window HDRFilm's storage covers (hdrfilm.cpp:37-38) — what the render calls write
    d.film.crop_offset[0] = film->crop_offset().x; d.film.crop_offset[1] = film->crop_offset().y;
    d.film.crop_size[0] = film->crop_size().x; d.film.crop_size[1] = film->crop_size().y;
    d.film.filter_radius = film->filter()->radius();
    std::memcpy(d.film.filter_lut, film->filter()->values().data(), sizeof d.film.filter_lut);
    d.cie1931_xyz = cie1931_xyz_table(); d.d65 = d65_table();
    d.n_textures = (uint32_t) out.textures.size(); d.textures = out.textures.data();
    d.n_regular_spectra = (uint32_t) out.regular.size(); d.n_regular_values = (uint32_t) out.regular_values.size();
    d.regular_spectra = out.regular.data(); d.regular_values = out.regular_values.data();
}

// "0,1,2" -> {0,1,2}; empty -> {single}
static std::vector<int> parse_gpu_devices(const std::string &list, int single) {
    std::vector<int> ids;
    for (auto &tok : string::tokenize(list, ", ")) {
        char *end = nullptr;
        const long v = std::strtol(tok.c_str(), &end, 10);
        if (end == tok.c_str() || *end || v < 0) Throw("\"gpu_devices\": \"{}\" is not a list of device ordinals", list);
        ids.push_back((int) v);
    }
    if (ids.empty()) ids.push_back(single);
    return ids;
}

// =========================================================================== the "path" integrator
// integrators/path.cpp:19-21,133-140 — same plugin name and properties; render() runs on the MI355X.
class PathTracer final : public MonteCarloIntegrator {
public:
    PathTracer(const Properties &props) : MonteCarloIntegrator(props) {
        m_device = props.int_("gpu_device", 0);
        // gpu_devices="0,1,2": several GPUs behind this one integrator — the C ABI renders the samples sharded
        // over them and sums the films on the first (msk_gpu_init with n > 1); overrides gpu_device
        m_devices = parse_gpu_devices(props.string("gpu_devices", ""), m_device);
        // The reference's PathTracer shadows max_depth / rr_depth / hide_emitters with private members
        // fixed at -1 / 5 / false (path.cpp:135-136); `honor_properties` opts into the values
        // the XML specifies instead.
        m_honor = props.bool_("honor_properties", false);
        // "counter" (default): the stateless counter RNG of the wavefront kernels.  "pcg_block": the reference's `independent`
        // sampler as written — one PCG32 stream per image block (samplers/independent.cpp:9-35) — rendered one block per lane
        // (msk_serial.h): a fidelity mode for parity runs against the CPU path, not a fast one.
        const std::string rng = props.string("rng", "counter");
        if (rng == "counter") m_rng_mode = MSK_RNG_COUNTER;
        else if (rng == "pcg_block") m_rng_mode = MSK_RNG_PCG_BLOCK;
        else Throw("\"rng\" must be \"counter\" or \"pcg_block\" (got \"{}\")", rng);
    }
    ~PathTracer() { if (m_ctx) msk_gpu_shutdown(m_ctx); }

    void fill_params(const Sensor *sensor, msk_render_params &p) const {
        std::memset(&p, 0, sizeof p);
        p.spp = (uint32_t) sensor->sampler()->sample_count();
        p.seed = sensor->sampler()->base_seed();
        p.rng_mode = m_rng_mode;
        p.rr_depth = m_honor ? m_rr_depth : 5;
        p.max_depth = m_honor ? m_max_depth : -1;
        p.hide_emitters = m_honor ? (m_hide_emitters ? 1 : 0) : 0;
        p.block_size = (int32_t) m_block_size;
        p.block_first = 0; p.block_stride = 1; p.sample_first = 0; p.sample_stride = 1;
    }

    bool render(Scene *scene, Sensor *sensor) override {           // integrator.cpp:31-80
        ref<Film> film = sensor->film();
        const Vector2i size = film->size();
        film->prepare({"X", "Y", "Z", "A", "W"});
        Log(Info, "Starting render job ({}x{}, {} sample)", size.x, size.y, sensor->sampler()->sample_count());
        auto t0 = std::chrono::steady_clock::now();
        FlatScene flat;
        flatten_scene(scene, sensor, flat);
        fill_params(sensor, flat.params);
        if (!m_ctx && msk_gpu_init(m_devices.data(), (int) m_devices.size(), &m_ctx) != MSK_OK) Throw("{}", msk_gpu_last_error(nullptr));
        msk_scene *gs = nullptr;
        if (msk_gpu_scene_create(m_ctx, &flat.desc, &gs) != MSK_OK) Throw("{}", msk_gpu_last_error(m_ctx));
        ref<ImageBlock> whole = new ImageBlock(film->crop_size(), 5);        // the crop window (the whole film by default), as the storage holds it
        whole->set_offset(film->crop_offset());
        msk_stats st;
        const int rc = msk_gpu_render(gs, &flat.params, whole->data().data(), &st);
        msk_gpu_scene_destroy(gs);
        if (rc != MSK_OK) Throw("{}", msk_gpu_last_error(m_ctx));
        film->put(whole);
        m_last_stats = st;
        // ImageBlock::put logs every such sample as it arrives (imageblock.cpp:57-81); the samples are splatted on the device, so
        // the plugin reports how many there were
        if (st.invalid_samples) Log(Warn, "Invalid sample value: {} of {} samples were negative or not finite", st.invalid_samples, st.samples);
        const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        Log(Info, "Rendering finished. (took {}s, device {} ms, {} Msamples/s)", secs, st.ms_total,
            st.ms_total > 0 ? st.samples / (st.ms_total * 1e3) : 0.0);
        return true;
    }
    const msk_stats &last_stats() const { return m_last_stats; }
    MSK_DECLARE_CLASS()
private:
    int m_device = 0;
    std::vector<int> m_devices;
    bool m_honor = false;
    int m_rng_mode = MSK_RNG_COUNTER;
    msk_ctx *m_ctx = nullptr;
    msk_stats m_last_stats{};
};
MSK_IMPLEMENT_CLASS(PathTracer, MonteCarloIntegrator)
MSK_REGISTER_INSTANCE(PathTracer, "path")

// integrators/aov.cpp:19-144 — same plugin name, "aovs" string and nested-integrator convention; render() runs the
// primary-hit channels and the nested path integrator on the MI355X in one job.
class AOVIntegrator final : public MonteCarloIntegrator {
public:
    AOVIntegrator(const Properties &props) : MonteCarloIntegrator(props) {
        m_device = props.int_("gpu_device", 0);
        m_devices = parse_gpu_devices(props.string("gpu_devices", ""), m_device);
        for (const std::string &token : string::tokenize(props.string("aovs", ""))) {
            std::vector<std::string> item = string::tokenize(token, ":");
            if (item.size() != 2 || item[0].empty() || item[1].empty()) {
                Log(Warn, "Invalid AOV specification: require <name>:<type> pair");
                continue;
            }
            auto add = [&](int32_t type, std::initializer_list<const char *> suffixes) {
                m_types.push_back(type);
                for (const char *sfx : suffixes) m_names.push_back(item[0] + sfx);
            };
            if (item[1] == "depth") add(MSK_AOV_DEPTH, {""});
            else if (item[1] == "position") add(MSK_AOV_POSITION, {".X", ".Y", ".Z"});
            else if (item[1] == "uv") add(MSK_AOV_UV, {".U", ".V"});
            else if (item[1] == "geo_normal") add(MSK_AOV_GEO_NORMAL, {".X", ".Y", ".Z"});
            else if (item[1] == "sh_normal") add(MSK_AOV_SH_NORMAL, {".X", ".Y", ".Z"});
            else Throw("Invalid AOV type \"{}\"!", item[1]);
        }
        for (auto &kv : props.objects()) {
            auto *integrator = dynamic_cast<SamplingIntegrator *>(kv.second.get());
            if (!integrator) Throw("Child objects must be of type 'SamplingIntegrator'!");
            auto *path = dynamic_cast<PathTracer *>(integrator);
            if (!path) Throw("aov: nested integrator \"{}\" is not the \"path\" integrator: not supported by the GPU back end", kv.first);
            if (m_path) Throw("aov: at most one nested integrator is supported by the GPU back end");
            m_path = path;
            m_types.push_back(MSK_AOV_PATH_RGBA);
            for (auto &name : integrator->aov_names()) m_names.push_back(kv.first + "." + name);
            for (const char *sfx : {".R", ".G", ".B", ".A"}) m_names.push_back(kv.first + sfx);
        }
        if (m_names.empty()) Log(Warn, "No AOVs were specified!");
    }
    ~AOVIntegrator() { if (m_ctx) msk_gpu_shutdown(m_ctx); }
    std::vector<std::string> aov_names() const override { return m_names; }
    const std::vector<int32_t> &aov_types() const { return m_types; }
    void fill_params(const Sensor *sensor, msk_render_params &p) const {
        if (m_path) m_path->fill_params(sensor, p);
        else {
            std::memset(&p, 0, sizeof p);
            p.spp = (uint32_t) sensor->sampler()->sample_count(); p.seed = sensor->sampler()->base_seed();
            p.rng_mode = MSK_RNG_COUNTER; p.rr_depth = 5; p.max_depth = -1; p.block_stride = 1; p.sample_stride = 1;
        }
        p.block_size = (int32_t) m_block_size;      // the tile loop is the outer integrator's (integrator.cpp:45)
    }

    bool render(Scene *scene, Sensor *sensor) override {           // integrator.cpp:31-80
        ref<Film> film = sensor->film();
        const Vector2i size = film->size();
        std::vector<std::string> channels = {"X", "Y", "Z", "A", "W"};
        channels.insert(channels.end(), m_names.begin(), m_names.end());
        film->prepare(channels);
        Log(Info, "Starting render job ({}x{}, {} sample)", size.x, size.y, sensor->sampler()->sample_count());
        FlatScene flat;
        flatten_scene(scene, sensor, flat);
        fill_params(sensor, flat.params);
        if (!m_ctx && msk_gpu_init(m_devices.data(), (int) m_devices.size(), &m_ctx) != MSK_OK) Throw("{}", msk_gpu_last_error(nullptr));
        msk_scene *gs = nullptr;
        if (msk_gpu_scene_create(m_ctx, &flat.desc, &gs) != MSK_OK) Throw("{}", msk_gpu_last_error(m_ctx));
        ref<ImageBlock> whole = new ImageBlock(film->crop_size(), channels.size());   // the crop window, as the storage holds it (as in "path")
        whole->set_offset(film->crop_offset());
        msk_stats st;
        const int rc = msk_gpu_render_aov(gs, &flat.params, m_types.data(), (uint32_t) m_types.size(), whole->data().data(), &st);
        msk_gpu_scene_destroy(gs);
        if (rc != MSK_OK) Throw("{}", msk_gpu_last_error(m_ctx));
        film->put(whole);
        m_last_stats = st;
        if (st.invalid_samples) Log(Warn, "Invalid sample value: {} of {} samples were not finite", st.invalid_samples, st.samples);
        Log(Info, "Rendering finished. (device {} ms)", st.ms_total);
        return true;
    }
    const msk_stats &last_stats() const { return m_last_stats; }
    MSK_DECLARE_CLASS()
private:
    std::vector<int32_t> m_types;
    std::vector<std::string> m_names;
    ref<PathTracer> m_path;
    int m_device = 0;
    std::vector<int> m_devices;
    msk_ctx *m_ctx = nullptr;
    msk_stats m_last_stats{};
};
MSK_IMPLEMENT_CLASS(AOVIntegrator, MonteCarloIntegrator)
MSK_REGISTER_INSTANCE(AOVIntegrator, "aov")

std::vector<int32_t> integrator_aov_types(const Integrator *integ) {
    auto *av = dynamic_cast<const AOVIntegrator *>(integ);
    return av ? av->aov_types() : std::vector<int32_t>{};
}
std::vector<std::string> integrator_aov_names(const Integrator *integ) {
    auto *si = dynamic_cast<const SamplingIntegrator *>(integ);
    return si ? si->aov_names() : std::vector<std::string>{};
}

// used by capi.cpp
void path_fill_params(const Integrator *integ, const Sensor *sensor, msk_render_params &p) {
    if (auto *av = dynamic_cast<const AOVIntegrator *>(integ)) { av->fill_params(sensor, p); return; }
    auto *pt = dynamic_cast<const PathTracer *>(integ);
    if (!pt) Throw("the scene's integrator is not a GPU integrator (\"path\" or \"aov\")");
    pt->fill_params(sensor, p);
}
bool path_last_stats(const Integrator *integ, msk_stats &st) {
    if (auto *pt = dynamic_cast<const PathTracer *>(integ)) { st = pt->last_stats(); return true; }
    if (auto *av = dynamic_cast<const AOVIntegrator *>(integ)) { st = av->last_stats(); return true; }
    return false;
}

}  // namespace misaki
