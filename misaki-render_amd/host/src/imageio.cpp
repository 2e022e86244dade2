// imageio.cpp — float image output for HDRFilm::develop (the reference goes through OpenImageIO,
// src/librender/image.cpp:20-43): PFM and uncompressed scanline OpenEXR (32-bit float channels).
#include <misaki/render.h>

#include <algorithm>
#include <cstring>
#include <fstream>

namespace misaki {

void write_pfm(const std::string &path, int w, int h, int channels, const float *data) {
    if (channels != 1 && channels != 3) Throw("write_pfm(): 1 or 3 channels expected");
    std::ofstream os(path, std::ios::binary);
    if (!os) Throw("Could not open \"{}\" for writing", path);
    os << (channels == 3 ? "PF" : "Pf") << "\n" << w << " " << h << "\n-1.0\n";   // little endian, bottom-up rows
    for (int y = h - 1; y >= 0; --y) os.write((const char *) (data + (size_t) y * w * channels), (std::streamsize) sizeof(float) * w * channels);
}

void write_exr(const std::string &path, int w, int h, const std::vector<std::string> &channels, const float *data) {
    std::ofstream os(path, std::ios::binary);
    if (!os) Throw("Could not open \"{}\" for writing", path);
    auto put32 = [&](uint32_t v) { os.write((const char *) &v, 4); };
    auto put64 = [&](uint64_t v) { os.write((const char *) &v, 8); };
    auto attr = [&](const char *name, const char *type, const std::string &payload) {
        os.write(name, std::strlen(name) + 1); os.write(type, std::strlen(type) + 1);
        put32((uint32_t) payload.size()); os.write(payload.data(), (std::streamsize) payload.size());
    };
    put32(20000630u); put32(2u);                           // magic, version 2, single-part scanline
    std::vector<size_t> order(channels.size());            // channels are stored alphabetically
    for (size_t i = 0; i < order.size(); ++i) order[i] = i;
    std::sort(order.begin(), order.end(), [&](size_t a, size_t b) { return channels[a] < channels[b]; });
    std::string ch;
    for (size_t i : order) {
        ch += channels[i]; ch += '\0';
        uint32_t v[4] = {2u /* FLOAT */, 0u, 1u, 1u};       // type, pLinear + reserved, xSampling, ySampling
        ch.append((const char *) v, 16);
    }
    ch += '\0';
    attr("channels", "chlist", ch);
    attr("compression", "compression", std::string(1, '\0'));
    int32_t box[4] = {0, 0, w - 1, h - 1};
    attr("dataWindow", "box2i", std::string((const char *) box, 16));
    attr("displayWindow", "box2i", std::string((const char *) box, 16));
    attr("lineOrder", "lineOrder", std::string(1, '\0'));
    float one = 1.f, zero2[2] = {0.f, 0.f};
    attr("pixelAspectRatio", "float", std::string((const char *) &one, 4));
    attr("screenWindowCenter", "v2f", std::string((const char *) zero2, 8));
    attr("screenWindowWidth", "float", std::string((const char *) &one, 4));
    os.put('\0');
    const size_t nc = channels.size();
    const uint64_t line_bytes = (uint64_t) w * nc * 4;
    uint64_t offset = (uint64_t) os.tellp() + (uint64_t) h * 8;
    for (int y = 0; y < h; ++y) { put64(offset); offset += 8 + line_bytes; }
    std::vector<float> row((size_t) w);
    for (int y = 0; y < h; ++y) {
        put32((uint32_t) y); put32((uint32_t) line_bytes);
        for (size_t i : order) {
            for (int x = 0; x < w; ++x) row[x] = data[((size_t) y * w + x) * nc + i];
            os.write((const char *) row.data(), (std::streamsize) w * 4);
        }
    }
}

}  // namespace misaki
