// xml.cpp — scene loader for the reference's Mitsuba-style XML (src/librender/xml.cpp:344-740),
// re-authored on a small built-in XML reader (pugixml is not available).  Same tag set, same
// attribute checks, same "$param" substitution, same naming of unnamed children ("_arg_<n>", so
// child order follows the reference's std::map iteration, xml.cpp:406-409), same two-phase
// parse -> instantiate scheme.  <rotate> is implemented (the reference registers the tag but
// ignores it); <include>/<alias> are rejected explicitly.
#include <misaki/core.h>
#include <misaki/render.h>

#include <cmath>
#include <cstring>
#include <fstream>
#include <set>

namespace misaki {
const Class *class_for_tag(const std::string &tag);     // core.cpp

namespace xml {
namespace {

struct Node {
    std::string name;
    std::vector<std::pair<std::string, std::string>> attrs;
    std::vector<Node> children;
    size_t offset = 0;
    const std::string *attr(const std::string &k) const {
        for (auto &a : attrs) if (a.first == k) return &a.second;
        return nullptr;
    }
    std::string value(const std::string &k) const { auto *a = attr(k); return a ? *a : std::string(); }
    void set(const std::string &k, const std::string &v) {
        for (auto &a : attrs) if (a.first == k) { a.second = v; return; }
        attrs.emplace_back(k, v);
    }
};

struct Reader {
    const std::string &s; size_t p = 0; std::string id;
    [[noreturn]] void fail(size_t at, const std::string &msg) const {
        int line = 1, col = 1;
        for (size_t i = 0; i < at && i < s.size(); ++i) { if (s[i] == '\n') { ++line; col = 1; } else ++col; }
        Throw("Error while loading \"{}\" (at line {}, col {}): {}", id, line, col, msg);
    }
    void skip_ws() { while (p < s.size() && std::isspace((unsigned char) s[p])) ++p; }
    bool starts(const char *t) const { return s.compare(p, std::strlen(t), t) == 0; }
    void skip_misc() {
        for (;;) {
            skip_ws();
            if (starts("<!--")) { size_t e = s.find("-->", p); if (e == std::string::npos) fail(p, "unterminated comment"); p = e + 3; }
            else if (starts("<?")) { size_t e = s.find("?>", p); if (e == std::string::npos) fail(p, "unterminated declaration"); p = e + 2; }
            else if (starts("<!DOCTYPE")) { size_t e = s.find('>', p); if (e == std::string::npos) fail(p, "unterminated doctype"); p = e + 1; }
            else break;
        }
    }
    static std::string unescape(const std::string &v) {
        std::string o;
        for (size_t i = 0; i < v.size(); ++i) {
            if (v[i] != '&') { o += v[i]; continue; }
            auto eat = [&](const char *e, char c) { size_t n = std::strlen(e); if (v.compare(i, n, e) == 0) { o += c; i += n - 1; return true; } return false; };
            if (!(eat("&amp;", '&') || eat("&lt;", '<') || eat("&gt;", '>') || eat("&quot;", '"') || eat("&apos;", '\''))) o += v[i];
        }
        return o;
    }
    std::string ident() {
        size_t b = p;
        while (p < s.size() && (std::isalnum((unsigned char) s[p]) || s[p] == '_' || s[p] == '-' || s[p] == ':' || s[p] == '.')) ++p;
        if (b == p) fail(p, "expected a name");
        return s.substr(b, p - b);
    }
    Node element() {
        skip_misc();
        if (p >= s.size() || s[p] != '<') fail(p, "unexpected content");
        Node n; n.offset = p; ++p;
        n.name = ident();
        for (;;) {
            skip_ws();
            if (p >= s.size()) fail(n.offset, "unterminated element");
            if (s[p] == '/') { if (p + 1 >= s.size() || s[p + 1] != '>') fail(p, "expected '/>'"); p += 2; return n; }
            if (s[p] == '>') { ++p; break; }
            std::string k = ident();
            skip_ws();
            if (p >= s.size() || s[p] != '=') fail(p, "expected '=' after attribute name");
            ++p; skip_ws();
            if (p >= s.size() || (s[p] != '"' && s[p] != '\'')) fail(p, "expected a quoted attribute value");
            char q = s[p++]; size_t e = s.find(q, p);
            if (e == std::string::npos) fail(p, "unterminated attribute value");
            n.attrs.emplace_back(k, unescape(s.substr(p, e - p)));
            p = e + 1;
        }
        for (;;) {
            skip_misc();
            if (p >= s.size()) fail(n.offset, "missing closing tag for <" + n.name + ">");
            if (starts("</")) {
                p += 2; std::string c = ident(); skip_ws();
                if (c != n.name) fail(p, "mismatched closing tag </" + c + "> for <" + n.name + ">");
                if (p >= s.size() || s[p] != '>') fail(p, "expected '>'");
                ++p; return n;
            }
            if (s[p] != '<') fail(p, "unexpected content");
            n.children.push_back(element());
        }
    }
};

enum class Tag { Boolean, Integer, Float, String, Vector, Transform, Translate, Matrix, Rotate, Scale, LookAt, NamedReference,
                 RGB, Spectrum, Include, Alias, Default, Object, Invalid };

Tag tag_of(const std::string &name) {
    static const std::map<std::string, Tag> t = {
        {"boolean", Tag::Boolean}, {"integer", Tag::Integer}, {"float", Tag::Float}, {"string", Tag::String}, {"vector", Tag::Vector},
        {"point", Tag::Vector}, {"transform", Tag::Transform}, {"translate", Tag::Translate}, {"matrix", Tag::Matrix}, {"rotate", Tag::Rotate},
        {"scale", Tag::Scale}, {"lookat", Tag::LookAt}, {"ref", Tag::NamedReference}, {"rgb", Tag::RGB}, {"spectrum", Tag::Spectrum},
        {"include", Tag::Include}, {"alias", Tag::Alias}, {"default", Tag::Default}};
    auto it = t.find(name);
    if (it != t.end()) return it->second;
    return class_for_tag(name) ? Tag::Object : Tag::Invalid;
}

struct Instance { Properties props; const Class *clazz = nullptr; ref<Object> object; size_t offset = 0; };
struct Context {
    const Reader *rd;
    std::map<std::string, Instance> instances;
    Transform4f transform;
    size_t id_counter = 0;
};

float parse_float(const Reader &rd, const Node &n, const std::string &v) {
    try {
        size_t off = 0; float r = std::stof(v, &off);
        for (size_t i = off; i < v.size(); ++i) if (!std::isspace((unsigned char) v[i])) throw std::invalid_argument("trailing");
        return r;
    } catch (...) { rd.fail(n.offset, "could not parse floating point value \"" + v + "\""); }
}
void check_attributes(const Reader &rd, const Node &n, std::set<std::string> allowed, bool expect_all = true) {
    for (auto &a : n.attrs) {
        if (!allowed.count(a.first)) rd.fail(n.offset, "unexpected attribute \"" + a.first + "\" in element \"" + n.name + "\"");
        allowed.erase(a.first);
    }
    if (expect_all && !allowed.empty()) rd.fail(n.offset, "missing attribute \"" + *allowed.begin() + "\" in element \"" + n.name + "\"");
}
void expand_value_to_xyz(Node &n) {       // xml.cpp: value="a" or "a,b,c" -> x,y,z
    if (auto *v = n.attr("value")) {
        auto t = string::tokenize(*v);
        if (t.size() == 1) t = {t[0], t[0], t[0]};
        if (t.size() == 3) {
            n.set("x", t[0]); n.set("y", t[1]); n.set("z", t[2]);
            for (size_t i = 0; i < n.attrs.size(); ++i) if (n.attrs[i].first == "value") { n.attrs.erase(n.attrs.begin() + i); break; }
        }
    }
}
Vector3f parse_vector(const Reader &rd, const Node &n, float def = 0.f) {
    Vector3f v{def, def, def};
    if (auto *a = n.attr("x")) v.x = parse_float(rd, n, *a);
    if (auto *a = n.attr("y")) v.y = parse_float(rd, n, *a);
    if (auto *a = n.attr("z")) v.z = parse_float(rd, n, *a);
    return v;
}
Vector3f parse_named_vector(const Reader &rd, const Node &n, const std::string &attr) {
    auto t = string::tokenize(n.value(attr));
    if (t.size() != 3) rd.fail(n.offset, "\"" + attr + "\": expected three values");
    return Vector3f{parse_float(rd, n, t[0]), parse_float(rd, n, t[1]), parse_float(rd, n, t[2])};
}

std::pair<std::string, std::string> parse(Context &ctx, Node &n, Tag parent_tag, Properties &props, ParameterList &param,
                                          size_t &arg_counter, bool within_emitter, bool within_spectrum) {
    const Reader &rd = *ctx.rd;
    for (auto &a : n.attrs) {
        if (a.second.find('$') == std::string::npos) continue;
        for (auto &kv : param) {
            std::string key = "$" + kv.first; size_t pos = 0;
            while ((pos = a.second.find(key, pos)) != std::string::npos) { a.second.replace(pos, key.size(), kv.second); pos += kv.second.size(); }
        }
        if (a.second.find('$') != std::string::npos) rd.fail(n.offset, "undefined parameter in \"" + a.second + "\"");
    }
    Tag tag = tag_of(n.name);
    if (tag == Tag::Invalid) rd.fail(n.offset, "unexpected tag \"" + n.name + "\"");
    const bool has_parent = parent_tag != Tag::Invalid, parent_is_object = has_parent && parent_tag == Tag::Object,
               current_is_object = tag == Tag::Object, parent_is_transform = parent_tag == Tag::Transform,
               current_is_transform_op = tag == Tag::Translate || tag == Tag::Rotate || tag == Tag::Scale || tag == Tag::LookAt || tag == Tag::Matrix;
    if (!has_parent && !current_is_object) rd.fail(n.offset, "root element \"" + n.name + "\" must be an object");
    if (parent_is_transform != current_is_transform_op)
        rd.fail(n.offset, parent_is_transform ? "transform nodes can only contain transform operations"
                                              : "transform operations can only occur in a transform node");
    if (has_parent && !parent_is_object && !(parent_is_transform && current_is_transform_op))
        rd.fail(n.offset, "node \"" + n.name + "\" cannot occur as child of a property");
    if (n.name == "scene") n.set("type", "scene");
    if (auto *nm = n.attr("name")) {
        if (!nm->empty() && (*nm)[0] == '_') rd.fail(n.offset, "invalid parameter name \"" + *nm + "\" in element \"" + n.name + "\": leading underscores are reserved for internal identifiers.");
    } else if (current_is_object || tag == Tag::NamedReference) {
        n.set("name", "_arg_" + std::to_string(arg_counter++));
    }
    if (auto *id = n.attr("id")) {
        if (!id->empty() && (*id)[0] == '_') rd.fail(n.offset, "invalid id \"" + *id + "\" in element \"" + n.name + "\": leading underscores are reserved for internal identifiers.");
    } else if (current_is_object) {
        n.set("id", "_unnamed_" + std::to_string(ctx.id_counter++));
    }
    try {
        switch (tag) {
            case Tag::Object: {
                check_attributes(rd, n, {"type", "id", "name"});
                const std::string id = n.value("id"), name = n.value("name"), type = n.value("type");
                Properties nested(type);
                nested.set_id(id);
                if (ctx.instances.count(id)) rd.fail(n.offset, "\"" + n.name + "\" has duplicate id \"" + id + "\"");
                const Class *cls = class_for_tag(n.name);
                if (!cls) rd.fail(n.offset, "could not retrieve class object for tag \"" + n.name + "\"");
                size_t nested_counter = 0;
                ParameterList local = param;                      // <default> is scoped like the reference's in-order attribute rewrite
                for (auto &ch : n.children) {
                    if (tag_of(ch.name) == Tag::Default) {
                        check_attributes(rd, ch, {"name", "value"});
                        bool found = false;
                        for (auto &kv : param) if (kv.first == ch.value("name")) found = true;
                        if (!found) param.emplace_back(ch.value("name"), ch.value("value"));
                        continue;
                    }
                    auto r = parse(ctx, ch, tag, nested, param, nested_counter, n.name == "emitter", n.name == "spectrum");
                    if (!r.second.empty()) nested.set_named_reference(r.first, NamedReference(r.second));
                }
                (void) local;
                Instance &inst = ctx.instances[id];
                inst.props = nested; inst.clazz = cls; inst.offset = n.offset;
                return {name, id};
            }
            case Tag::NamedReference:
                check_attributes(rd, n, {"name", "id"});
                return {n.value("name"), n.value("id")};
            case Tag::Default:
                rd.fail(n.offset, "<default> can only occur directly inside an object");
            case Tag::Include: case Tag::Alias:
                rd.fail(n.offset, "<" + n.name + "> is not supported by this loader");
            case Tag::String:
                check_attributes(rd, n, {"name", "value"});
                props.set_string(n.value("name"), n.value("value"));
                break;
            case Tag::Boolean: {
                check_attributes(rd, n, {"name", "value"});
                std::string v = string::to_lower(n.value("value"));
                if (v != "true" && v != "false") rd.fail(n.offset, "could not parse boolean value \"" + v + "\" -- must be \"true\" or \"false\"");
                props.set_bool(n.value("name"), v == "true");
                break;
            }
            case Tag::Float:
                check_attributes(rd, n, {"name", "value"});
                props.set_float(n.value("name"), parse_float(rd, n, n.value("value")));
                break;
            case Tag::Integer: {
                check_attributes(rd, n, {"name", "value"});
                long long v;
                try { size_t off = 0; v = std::stoll(n.value("value"), &off); if (off != n.value("value").size()) throw 0; }
                catch (...) { rd.fail(n.offset, "could not parse integer value \"" + n.value("value") + "\""); }
                props.set_int(n.value("name"), (int) v);
                break;
            }
            case Tag::Vector:
                expand_value_to_xyz(n);
                check_attributes(rd, n, {"name", "x", "y", "z"});
                props.set_vector3(n.value("name"), parse_vector(rd, n));
                break;
            case Tag::RGB: {
                check_attributes(rd, n, {"name", "value"});
                auto t = string::tokenize(n.value("value"));
                if (t.size() == 1) t = {t[0], t[0], t[0]};
                if (t.size() != 3) rd.fail(n.offset, "'rgb' tag requires one or three values (got \"" + n.value("value") + "\")");
                Color3 c{parse_float(rd, n, t[0]), parse_float(rd, n, t[1]), parse_float(rd, n, t[2])};
                if (!within_spectrum) {       // xml.cpp:269-277: srgb_d65 inside <emitter>, srgb elsewhere
                    // "eta" / "k" are physical quantities, not reflectances: the bounded srgb texture would clamp
                    // them to 1 (rgb2spec.c:83-84).  Like Mitsuba 2's loader they get the unbounded variant here.
                    const std::string pname = n.value("name");
                    const bool unbounded = !within_emitter && (pname == "eta" || pname == "k");
                    Properties p(within_emitter ? "srgb_d65" : unbounded ? "srgb_unbounded" : "srgb");
                    p.set_color("color", c);
                    props.set_object(n.value("name"), InstanceManager::get()->create_instance(p, Class::for_name("Texture")));
                } else {
                    props.set_color("color", c);
                }
                break;
            }
            case Tag::Spectrum: {
                check_attributes(rd, n, {"name", "value", "filename"}, false);
                // xml.cpp:565-627: exactly one of value / filename; one token = a constant, else wavelength:value pairs
                if ((n.attr("value") != nullptr) == (n.attr("filename") != nullptr)) rd.fail(n.offset, "'spectrum' tag requires one of \"value\" or \"filename\" attributes");
                if (n.attr("filename")) rd.fail(n.offset, "<spectrum filename=...> is not implemented (nor is it by the reference loader, xml.cpp:620-622)");
                auto t = string::tokenize(n.value("value"));
                if (t.size() == 1) {
                    float c = parse_float(rd, n, t[0]);
                    // xml.cpp:279-300: uniform c (reflectance) or D65 * c (inside an emitter)
                    Properties p(within_emitter ? "d65" : "uniform");
                    if (within_emitter) p.set_float("scale", c); else p.set_float("value", c);
                    ref<Object> o = InstanceManager::get()->create_instance(p, Class::for_name("Texture"));
                    auto ex = o->expand();
                    props.set_object(n.value("name"), ex.empty() ? o : ex[0]);
                    break;
                }
                // xml.cpp:594-619 + create_texture_from_spectrum (:300-341): the pairs, the unit conversion inside an emitter
                // (values are scaled so that D65 integrates to a luminance of 1), regular or irregular by the steps
                std::vector<float> wavelengths, values;
                for (auto &tok : t) {
                    auto pair = string::tokenize(tok, ":");
                    if (pair.size() != 2) rd.fail(n.offset, "invalid spectrum (expected wavelength:value pairs)");
                    wavelengths.push_back(parse_float(rd, n, pair[0])); values.push_back(parse_float(rd, n, pair[1]));
                }
                const float unit_conversion = within_emitter ? MSK_CIE_Y_NORMALIZATION : 1.f;
                bool is_regular = true;
                float interval = 0.f;
                for (size_t k = 0; k < wavelengths.size(); ++k) {
                    values[k] *= unit_conversion;
                    if (k == 0) continue;
                    const float distance = wavelengths[k] - wavelengths[k - 1];
                    if (distance < 0.f) rd.fail(n.offset, "Wavelengths must be specified in increasing order!");
                    if (k == 1) interval = distance;
                    else if (std::abs(distance - interval) > math::Epsilon) is_regular = false;
                }
                if (!is_regular)
                    rd.fail(n.offset, "spectrum \"" + n.value("name") + "\": unequal wavelength steps make an `irregular` spectrum (spectra/irregular.cpp), which "
                            "the GPU path integrator cannot evaluate; resample it on a regular grid (the `regular` plugin)");
                Properties p("regular");
                p.set_int("size", (int) wavelengths.size());
                p.set_float("lambda_min", wavelengths.front());
                p.set_float("lambda_max", wavelengths.back());
                p.set_pointer("values", values.data());              // read by the plugin's constructor, below
                props.set_object(n.value("name"), InstanceManager::get()->create_instance(p, Class::for_name("Texture")));
                break;
            }
            case Tag::Transform:
                check_attributes(rd, n, {"name"});
                ctx.transform = Transform4f();
                break;
            case Tag::LookAt: {
                check_attributes(rd, n, {"origin", "target", "up"});
                Transform4f r = Transform4f::lookat(parse_named_vector(rd, n, "origin"), parse_named_vector(rd, n, "target"),
                                                    parse_named_vector(rd, n, "up"));
                if (r.matrix().has_nan()) rd.fail(n.offset, "invalid lookat transformation");
                ctx.transform = r * ctx.transform;
                break;
            }
            case Tag::Translate:
                expand_value_to_xyz(n);
                check_attributes(rd, n, {"x", "y", "z"}, false);
                ctx.transform = Transform4f::translate(parse_vector(rd, n)) * ctx.transform;
                break;
            case Tag::Scale:
                expand_value_to_xyz(n);
                check_attributes(rd, n, {"x", "y", "z"}, false);
                ctx.transform = Transform4f::scale(parse_vector(rd, n, 1.f)) * ctx.transform;
                break;
            case Tag::Rotate: {
                expand_value_to_xyz(n);
                check_attributes(rd, n, {"x", "y", "z", "angle"}, false);
                ctx.transform = Transform4f::rotate(parse_vector(rd, n), parse_float(rd, n, n.value("angle"))) * ctx.transform;
                break;
            }
            case Tag::Matrix: {
                check_attributes(rd, n, {"value"});
                auto t = string::tokenize(n.value("value"), " ,");
                if (t.size() != 16) rd.fail(n.offset, "matrix: expected 16 values");
                Matrix4f m;
                for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) m.m[i][j] = parse_float(rd, n, t[i * 4 + j]);
                ctx.transform = Transform4f(m) * ctx.transform;
                break;
            }
            default: break;
        }
        for (auto &ch : n.children) parse(ctx, ch, tag, props, param, arg_counter, false, false);
        if (tag == Tag::Transform) props.set_transform(n.value("name"), ctx.transform);
    } catch (const std::exception &e) {
        if (std::strstr(e.what(), "Error while loading") == nullptr) rd.fail(n.offset, e.what());
        throw;
    }
    return {"", ""};
}

ref<Object> instantiate(Context &ctx, const std::string &id) {
    auto it = ctx.instances.find(id);
    if (it == ctx.instances.end()) Throw("reference to unknown object \"{}\"!", id);
    Instance &inst = it->second;
    if (inst.object) return inst.object;
    for (auto &kv : inst.props.named_references())
        inst.props.set_object(kv.first, instantiate(ctx, (const std::string &) kv.second), false);
    try {
        inst.object = InstanceManager::get()->create_instance(inst.props, inst.clazz);
    } catch (const std::exception &e) {
        if (std::strstr(e.what(), "Error while loading") != nullptr) throw;
        ctx.rd->fail(inst.offset, "could not instantiate " + string::to_lower(inst.clazz->name()) + " instance of type \"" +
                                      inst.props.instance_name() + "\": " + e.what());
    }
    return inst.object;
}

ref<Object> load(const std::string &text, const std::string &id, ParameterList parameters) {
    Class::static_initialization();
    Reader rd{text, 0, id};
    Node root = rd.element();
    rd.skip_misc();
    if (rd.p != text.size()) rd.fail(rd.p, "unexpected content after the root element");
    Context ctx; ctx.rd = &rd;
    Properties props; size_t counter = 0;
    auto r = parse(ctx, root, Tag::Invalid, props, parameters, counter, false, false);
    return instantiate(ctx, r.second);
}
}  // namespace

ref<Object> load_file(const std::string &filename, ParameterList parameters) {
    std::ifstream is(filename, std::ios::binary);
    if (!is) Throw("\"{}\": file not exists.", filename);
    Log(Info, "Loading XML file \"{}\" ..", filename);
    std::stringstream ss; ss << is.rdbuf();
    size_t slash = filename.find_last_of('/');
    get_file_resolver()->prepend(slash == std::string::npos ? "." : filename.substr(0, slash));
    return load(ss.str(), filename, std::move(parameters));
}
ref<Object> load_string(const std::string &text, const std::string &base_dir, ParameterList parameters) {
    get_file_resolver()->prepend(base_dir);
    return load(text, "<string>", std::move(parameters));
}

}  // namespace xml
}  // namespace misaki
