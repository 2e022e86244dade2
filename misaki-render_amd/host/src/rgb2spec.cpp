// rgb2spec.cpp — srgb_model_fetch (src/librender/srgb.cpp:11-28): RGB -> coefficients of the sigmoid-polynomial spectrum
// (render/srgb.h:8-19), read trilinearly from the res-64 sRGB table exactly as ext/rgb2spec/rgb2spec.c:77-119 does.
// The reference loads "data/srgb.coeff", written at build time by its rgb2spec_opt tool; this library looks for the same
// file (same layout) and, if there is none, computes the table with its own optimiser (rgb2spec_table.cpp, a few seconds
// on the host's cores) and caches it.  Search order:
//   $MSK_SRGB_COEFF (must load if set), the file resolver's "data/srgb.coeff" (the reference's one location, srgb.cpp:13: a
//   scene or an install that ships its own table is honoured), <directory of libmisaki-render.so>/srgb.coeff, and the per-user
//   cache ($XDG_CACHE_HOME or ~/.cache)/misaki-render/srgb.coeff.  A computed table is written beside the library when that
//   directory is writable (the in-tree build), else into the per-user cache — never into a read-only install.
// One deliberate deviation: pure black.  rgb2spec_fetch computes (res - 1) / 0 * 0 = NaN for it and returns NaN
// coefficients; here black is the constant-zero spectrum (0, 0, -inf) that srgb_model_eval (render/srgb.h:13-14) knows.
#include <misaki/render.h>

#include <dlfcn.h>

#include <cmath>
#include <filesystem>
#include <mutex>

namespace misaki {
namespace {
struct Model { std::vector<float> scale, data; std::string source; };
Model *g_model = nullptr;
std::mutex g_model_mutex;

std::string library_dir() {
    Dl_info info;
    if (dladdr((const void *) &library_dir, &info) && info.dli_fname) {
        std::string p = info.dli_fname;
        size_t slash = p.find_last_of('/');
        return slash == std::string::npos ? "." : p.substr(0, slash);
    }
    return ".";
}

const Model &model() {
    std::lock_guard<std::mutex> lock(g_model_mutex);
    if (g_model) return *g_model;
    auto *m = new Model();
    std::vector<std::string> candidates;
    if (const char *e = getenv("MSK_SRGB_COEFF")) candidates.push_back(e);
    const std::string beside = library_dir() + "/srgb.coeff";
    std::string cache;
    if (const char *x = getenv("XDG_CACHE_HOME")) cache = std::string(x) + "/misaki-render";
    else if (const char *h = getenv("HOME")) cache = std::string(h) + "/.cache/misaki-render";
    candidates.push_back(get_file_resolver()->resolve("data/srgb.coeff"));
    candidates.push_back(beside);
    if (!cache.empty()) candidates.push_back(cache + "/srgb.coeff");
    for (const auto &c : candidates)
        if (rgb2spec_read_table(c, m->scale, m->data)) { m->source = c; break; }
    if (m->source.empty()) {
        if (getenv("MSK_SRGB_COEFF")) { delete m; Throw("Could not load sRGB-to-spectrum upsampling model ('{}')", getenv("MSK_SRGB_COEFF")); }
        Log(Info, "Optimising the spectral upsampling model (sRGB, resolution 64) .. ");
        rgb2spec_build_table(64, m->scale, m->data, 0);
        m->source = "(computed)";
        try { rgb2spec_write_table(beside, m->scale, m->data); m->source = beside; }
        catch (const std::exception &) {                 // read-only install: the per-user cache, or just the in-memory table
            if (!cache.empty()) {
                try { std::filesystem::create_directories(cache); rgb2spec_write_table(cache + "/srgb.coeff", m->scale, m->data); m->source = cache + "/srgb.coeff"; }
                catch (const std::exception &) {}
            }
        }
    } else {
        Log(Info, "Loading spectral upsampling model \"{}\" .. ", m->source);
    }
    g_model = m;
    return *g_model;
}
}  // namespace

const std::string &srgb_model_source() { return model().source; }

Color3 srgb_model_fetch(const Color3 &c) {
    const float rgb[3] = {c.r, c.g, c.b};
    if (!std::isfinite(rgb[0]) || !std::isfinite(rgb[1]) || !std::isfinite(rgb[2])) Throw("srgb_model_fetch: colour is not finite");
    if (std::max(std::min(rgb[0], 1.f), 0.f) == 0.f && std::max(std::min(rgb[1], 1.f), 0.f) == 0.f && std::max(std::min(rgb[2], 1.f), 0.f) == 0.f)
        return Color3{0.f, 0.f, -INFINITY};
    const Model &m = model();
    float out[3];
    rgb2spec_fetch_table((int) m.scale.size(), m.scale.data(), m.data.data(), rgb, out);
    return Color3{out[0], out[1], out[2]};
}

}  // namespace misaki
