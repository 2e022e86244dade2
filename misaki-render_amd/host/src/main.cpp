// misaki-cli — command line front end (the reference's src/apps/main.cpp hard-codes its scene path
// and takes no arguments; this one takes them):
//   misaki-cli scene.xml [-o output.(exr|pfm)] [-D name=value ...] [-q]
//   misaki-cli --make-srgb-coeff out.coeff [res]     (what the reference's rgb2spec_opt tool does at build time)
#include <misaki/render.h>

#include <cstring>
#include <iostream>

using namespace misaki;

int main(int argc, char **argv) {
    std::string scene_path, out_path;
    xml::ParameterList params;
    if (argc >= 3 && std::strcmp(argv[1], "--make-srgb-coeff") == 0) {
        try {
            std::vector<float> scale, data;
            rgb2spec_build_table(argc > 3 ? std::atoi(argv[3]) : 64, scale, data, 0);
            rgb2spec_write_table(argv[2], scale, data);
        } catch (const std::exception &e) { std::cerr << e.what() << "\n"; return 1; }
        return 0;
    }
    for (int i = 1; i < argc; ++i) {
        std::string a = argv[i];
        if (a == "-o" && i + 1 < argc) out_path = argv[++i];
        else if (a == "-D" && i + 1 < argc) { std::string kv = argv[++i]; size_t eq = kv.find('='); if (eq != std::string::npos) params.emplace_back(kv.substr(0, eq), kv.substr(eq + 1)); }
        else if (a == "-q") set_log_level(Warn);
        else if (a == "-h" || a == "--help") { std::cout << "usage: misaki-cli scene.xml [-o out.exr|out.pfm] [-D name=value] [-q]\n"; return 0; }
        else scene_path = a;
    }
    if (scene_path.empty()) { std::cerr << "usage: misaki-cli scene.xml [-o out.exr|out.pfm] [-D name=value] [-q]\n"; return 2; }
    try {
        Class::static_initialization();
        ref<Object> root = xml::load_file(scene_path, params);
        auto *scene = dynamic_cast<Scene *>(root.get());
        if (!scene || !scene->sensor()) Throw("\"{}\" does not describe a scene with a sensor", scene_path);
        Film *film = scene->sensor()->film();
        if (out_path.empty()) { size_t dot = scene_path.find_last_of('.'); out_path = scene_path.substr(0, dot) + ".exr"; }
        film->set_destination_file(out_path);
        scene->integrator()->render(scene, scene->sensor());
        film->develop();
    } catch (const std::exception &e) {
        Log(Error, "Caught a critical exception: {}", e.what());      // main.cpp:55-57
        return 1;
    }
    return 0;
}
