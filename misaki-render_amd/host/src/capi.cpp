// capi.cpp — C entry points over the host library for scripting hosts and tests (ctypes).
#include <misaki/render.h>

#include <cstring>

namespace misaki {
void path_fill_params(const Integrator *integ, const Sensor *sensor, msk_render_params &p);
bool path_last_stats(const Integrator *integ, msk_stats &st);
std::vector<std::string> integrator_aov_names(const Integrator *integ);
std::vector<int32_t> integrator_aov_types(const Integrator *integ);
}
using namespace misaki;

struct msk_host_scene {
    ref<Scene> scene;
    FlatScene flat;
};
static thread_local std::string g_err;
static int fail(const std::exception &e) { g_err = e.what(); return -1; }

extern "C" {

const char *msk_host_last_error() { return g_err.c_str(); }
void msk_host_set_log_level(int level) { set_log_level((LogLevel) level); }

// params: "key=value;key=value" substituted for $key in the XML (xml.cpp:350-359)
int msk_host_load_scene(const char *xml_path, const char *params, msk_host_scene **out) {
    try {
        xml::ParameterList pl;
        if (params)
            for (auto &kv : string::tokenize(params, ";")) {
                size_t eq = kv.find('=');
                if (eq != std::string::npos) pl.emplace_back(kv.substr(0, eq), kv.substr(eq + 1));
            }
        ref<Object> root = xml::load_file(xml_path, pl);
        auto *scene = dynamic_cast<Scene *>(root.get());
        if (!scene) Throw("Root element of \"{}\" is not a scene", xml_path);
        if (!scene->sensor()) Throw("The scene has no sensor");
        auto *h = new msk_host_scene();
        h->scene = scene;
        *out = h;
        return 0;
    } catch (const std::exception &e) { return fail(e); }
}
void msk_host_free_scene(msk_host_scene *h) { delete h; }

// the flatten step of the "path" plugin; pointers inside *desc stay valid until the scene is freed
int msk_host_flatten(msk_host_scene *h, msk_scene_desc *desc, msk_render_params *params) {
    try {
        flatten_scene(h->scene.get(), h->scene->sensor(), h->flat);
        path_fill_params(h->scene->integrator(), h->scene->sensor(), h->flat.params);
        *desc = h->flat.desc;
        if (params) *params = h->flat.params;
        return 0;
    } catch (const std::exception &e) { return fail(e); }
}

// scene->integrator()->render(scene, sensor) followed by HDRFilm::image(); optionally develop() to a file
int msk_host_render(msk_host_scene *h, float *film_xyzaw, float *rgba, const char *develop_to, msk_stats *stats) {
    try {
        Scene *scene = h->scene.get();
        Sensor *sensor = scene->sensor();
        if (!scene->integrator()->render(scene, sensor)) Throw("render() failed");
        Film *film = sensor->film();
        const ImageBlock *st = film->storage();
        if (film_xyzaw) std::memcpy(film_xyzaw, st->data().data(), st->data().size() * sizeof(float));
        if (rgba) { auto img = film->image(); std::memcpy(rgba, img.data(), img.size() * sizeof(float)); }
        if (develop_to && *develop_to) { film->set_destination_file(develop_to); film->develop(); }
        if (stats) path_last_stats(scene->integrator(), *stats);
        return 0;
    } catch (const std::exception &e) { return fail(e); }
}

// aov_names() of the scene's integrator, '\n'-separated; returns the number of names or -1
int msk_host_aov_names(msk_host_scene *h, char *buf, size_t cap) {
    try {
        auto names = integrator_aov_names(h->scene->integrator());
        std::string all;
        for (auto &n : names) { all += n; all += '\n'; }
        if (all.size() + 1 > cap) Throw("msk_host_aov_names: buffer too small");
        std::memcpy(buf, all.c_str(), all.size() + 1);
        return (int) names.size();
    } catch (const std::exception &e) { return fail(e); }
}

// MSK_AOV_* of the "aov" integrator in order (what it passes to msk_gpu_render_aov); returns the count or -1
int msk_host_aov_types(msk_host_scene *h, int32_t *out, size_t cap) {
    try {
        auto t = integrator_aov_types(h->scene->integrator());
        if (t.size() > cap) Throw("msk_host_aov_types: buffer too small");
        std::copy(t.begin(), t.end(), out);
        return (int) t.size();
    } catch (const std::exception &e) { return fail(e); }
}

int msk_host_film_size(msk_host_scene *h, int *w, int *hgt, int *spp) {
    try {
        Sensor *s = h->scene->sensor();
        *w = s->film()->size().x; *hgt = s->film()->size().y; *spp = (int) s->sampler()->sample_count();
        return 0;
    } catch (const std::exception &e) { return fail(e); }
}

// the film's crop window {offset x, y, width, height}: the size of the storage msk_host_render copies out (film.cpp:12-21)
int msk_host_film_crop(msk_host_scene *h, int *xywh) {
    try {
        const Film *f = h->scene->sensor()->film();
        xywh[0] = f->crop_offset().x; xywh[1] = f->crop_offset().y; xywh[2] = f->crop_size().x; xywh[3] = f->crop_size().y;
        return 0;
    } catch (const std::exception &e) { return fail(e); }
}

int msk_host_srgb_model_fetch(const float *rgb, float *out) {
    try { Color3 c = srgb_model_fetch(Color3{rgb[0], rgb[1], rgb[2]}); out[0] = c.r; out[1] = c.g; out[2] = c.b; return 0; }
    catch (const std::exception &e) { return fail(e); }
}
// the res-`res` sRGB upsampling table computed by this library's optimiser, written in the layout of data/srgb.coeff
int msk_host_rgb2spec_build(int res, const char *path, int threads) {
    try { std::vector<float> scale, data; rgb2spec_build_table(res, scale, data, threads); rgb2spec_write_table(path, scale, data); return 0; }
    catch (const std::exception &e) { return fail(e); }
}
// where srgb_model_fetch's table came from (loads or computes it if that has not happened yet)
int msk_host_srgb_model_source(char *buf, size_t cap) {
    try { const std::string &s = srgb_model_source(); if (s.size() + 1 > cap) Throw("buffer too small"); std::memcpy(buf, s.c_str(), s.size() + 1); return 0; }
    catch (const std::exception &e) { return fail(e); }
}
int msk_host_write_image(const char *path, int w, int h, int channels, const float *data) {
    try {
        std::string p(path);
        if (p.size() > 4 && p.substr(p.size() - 4) == ".exr") {
            std::vector<std::string> names = channels == 4 ? std::vector<std::string>{"R", "G", "B", "A"} : std::vector<std::string>{"R", "G", "B"};
            write_exr(p, w, h, names, data);
        } else write_pfm(p, w, h, channels, data);
        return 0;
    } catch (const std::exception &e) { return fail(e); }
}
}
