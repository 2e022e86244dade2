// core.cpp — Object / Class / InstanceManager / Properties / Transform4f / logging for the host
// library (see include/misaki/core.h for the reference files each part mirrors).
#include <misaki/core.h>
#include <misaki/render.h>

#include <algorithm>
#include <cctype>
#include <chrono>
#include <cstring>
#include <ctime>
#include <fstream>
#include <iostream>
#include <mutex>
#include <sys/stat.h>

namespace misaki {

// ---------------------------------------------------------------- logging (src/librender/logger.cpp)
static LogLevel g_level = Info;
void set_log_level(LogLevel level) { g_level = level; }
void log_message(LogLevel level, const char *file, int line, const std::string &msg) {
    if (level < g_level) return;
    static const char *names[] = {"TRACE", "DEBUG", "INFO", "WARN", "ERROR"};
    std::time_t t = std::time(nullptr);
    char ts[32];
    std::strftime(ts, sizeof ts, "%Y-%m-%d %H:%M:%S", std::localtime(&t));
    const char *base = std::strrchr(file, '/');
    std::fprintf(stderr, "%s %-5s [%s:%d] %s\n", ts, names[level], base ? base + 1 : file, line, msg.c_str());
}
namespace detail {
void throw_at(const char *file, int line, const std::string &msg) {
    const char *base = std::strrchr(file, '/');
    throw std::runtime_error("[" + std::string(base ? base + 1 : file) + ":" + std::to_string(line) + "] " + msg);
}
}  // namespace detail

// ---------------------------------------------------------------- string helpers
namespace string {
std::vector<std::string> tokenize(const std::string &s, const std::string &delim) {
    std::vector<std::string> out;
    size_t i = 0;
    while (i < s.size()) {
        size_t j = s.find_first_of(delim, i);
        if (j == std::string::npos) j = s.size();
        if (j > i) out.push_back(s.substr(i, j - i));
        i = j + 1;
    }
    return out;
}
std::string to_lower(const std::string &s) {
    std::string r = s;
    for (auto &c : r) c = (char) std::tolower((unsigned char) c);
    return r;
}
std::string indent(const std::string &s, int amount) {
    std::string pad(amount, ' '), out;
    for (char c : s) { out += c; if (c == '\n') out += pad; }
    return out;
}
}  // namespace string

// ---------------------------------------------------------------- file resolver
std::string FileResolver::resolve(const std::string &name) const {
    if (!name.empty() && name[0] == '/') return name;
    struct stat st;
    for (const auto &p : m_paths) {
        std::string c = p + "/" + name;
        if (stat(c.c_str(), &st) == 0) return c;
    }
    return name;
}
FileResolver *get_file_resolver() { static FileResolver fr; return &fr; }

// ---------------------------------------------------------------- matrices
Matrix4f Matrix4f::identity() {
    Matrix4f r;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) r.m[i][j] = i == j ? 1.0 : 0.0;
    return r;
}
Matrix4f Matrix4f::operator*(const Matrix4f &o) const {
    Matrix4f r;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            double a = 0;
            for (int k = 0; k < 4; ++k) a += m[i][k] * o.m[k][j];
            r.m[i][j] = a;
        }
    return r;
}
Matrix4f Matrix4f::inverse() const {     // Gauss-Jordan with partial pivoting
    double a[4][8];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) { a[i][j] = m[i][j]; a[i][4 + j] = i == j ? 1.0 : 0.0; }
    for (int c = 0; c < 4; ++c) {
        int p = c;
        for (int r = c + 1; r < 4; ++r) if (std::fabs(a[r][c]) > std::fabs(a[p][c])) p = r;
        if (a[p][c] == 0.0) { Matrix4f n; for (auto &row : n.m) for (auto &v : row) v = NAN; return n; }
        if (p != c) for (int j = 0; j < 8; ++j) std::swap(a[p][j], a[c][j]);
        double inv = 1.0 / a[c][c];
        for (int j = 0; j < 8; ++j) a[c][j] *= inv;
        for (int r = 0; r < 4; ++r) if (r != c) { double f = a[r][c]; if (f != 0.0) for (int j = 0; j < 8; ++j) a[r][j] -= f * a[c][j]; }
    }
    Matrix4f r;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) r.m[i][j] = a[i][4 + j];
    return r;
}
bool Matrix4f::has_nan() const {
    for (auto &row : m) for (double v : row) if (std::isnan(v)) return true;
    return false;
}

void Transform4f::to_float16(float out[16]) const {
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) out[i * 4 + j] = (float) m_matrix.m[i][j];
}
// transform.h:127-135: fp32 matrix * (p,1), left to right, then / w
Vector3f Transform4f::apply_point(const Vector3f &p) const {
    float f[16]; to_float16(f);
    float r[4];
    for (int i = 0; i < 4; ++i) r[i] = ((f[i * 4] * p.x + f[i * 4 + 1] * p.y) + f[i * 4 + 2] * p.z) + f[i * 4 + 3] * 1.f;
    return Vector3f{r[0] / r[3], r[1] / r[3], r[2] / r[3]};
}
Vector3f Transform4f::apply_vector(const Vector3f &v) const {
    float f[16]; to_float16(f);
    return Vector3f{f[0] * v.x + (f[1] * v.y + f[2] * v.z), f[4] * v.x + (f[5] * v.y + f[6] * v.z), f[8] * v.x + (f[9] * v.y + f[10] * v.z)};
}
Vector3f Transform4f::apply_normal(const Vector3f &n) const {   // inverse-transpose 3x3
    const Matrix4f &i = m_inverse_matrix;
    auto g = [&](int r, int c) { return (float) i.m[c][r]; };
    return Vector3f{g(0, 0) * n.x + (g(0, 1) * n.y + g(0, 2) * n.z), g(1, 0) * n.x + (g(1, 1) * n.y + g(1, 2) * n.z),
                    g(2, 0) * n.x + (g(2, 1) * n.y + g(2, 2) * n.z)};
}
Transform4f Transform4f::translate(const Vector3f &v) {
    Matrix4f m = Matrix4f::identity(), i = Matrix4f::identity();
    m.m[0][3] = v.x; m.m[1][3] = v.y; m.m[2][3] = v.z;
    i.m[0][3] = -(double) v.x; i.m[1][3] = -(double) v.y; i.m[2][3] = -(double) v.z;
    return Transform4f(m, i);
}
Transform4f Transform4f::scale(const Vector3f &v) {
    Matrix4f m = Matrix4f::identity(), i = Matrix4f::identity();
    m.m[0][0] = v.x; m.m[1][1] = v.y; m.m[2][2] = v.z;
    i.m[0][0] = 1.0 / v.x; i.m[1][1] = 1.0 / v.y; i.m[2][2] = 1.0 / v.z;
    return Transform4f(m, i);
}
Transform4f Transform4f::rotate(const Vector3f &axis, float angle_deg) {
    double l = std::sqrt((double) axis.x * axis.x + (double) axis.y * axis.y + (double) axis.z * axis.z);
    double x = axis.x / l, y = axis.y / l, z = axis.z / l, a = angle_deg * M_PI / 180.0, c = std::cos(a), s = std::sin(a), t = 1 - c;
    Matrix4f m = Matrix4f::identity();
    m.m[0][0] = t * x * x + c;     m.m[0][1] = t * x * y - s * z; m.m[0][2] = t * x * z + s * y;
    m.m[1][0] = t * x * y + s * z; m.m[1][1] = t * y * y + c;     m.m[1][2] = t * y * z - s * x;
    m.m[2][0] = t * x * z - s * y; m.m[2][1] = t * y * z + s * x; m.m[2][2] = t * z * z + c;
    Matrix4f i = m;
    for (int r = 0; r < 3; ++r) for (int q = 0; q < 3; ++q) i.m[r][q] = m.m[q][r];
    return Transform4f(m, i);
}
// transform.h:169-178
Transform4f Transform4f::lookat(const Vector3f &origin, const Vector3f &target, const Vector3f &up) {
    auto nrm = [](double *v) { double l = std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]); v[0] /= l; v[1] /= l; v[2] /= l; };
    auto crs = [](const double *a, const double *b, double *c) { c[0] = a[1] * b[2] - a[2] * b[1]; c[1] = a[2] * b[0] - a[0] * b[2]; c[2] = a[0] * b[1] - a[1] * b[0]; };
    double dir[3] = {(double) target.x - origin.x, (double) target.y - origin.y, (double) target.z - origin.z};
    nrm(dir);
    double u[3] = {up.x, up.y, up.z}; nrm(u);
    double left[3]; crs(u, dir, left); nrm(left);
    double nup[3]; crs(dir, left, nup); nrm(nup);
    Matrix4f m = Matrix4f::identity();
    double o[3] = {origin.x, origin.y, origin.z};
    for (int r = 0; r < 3; ++r) { m.m[r][0] = left[r]; m.m[r][1] = nup[r]; m.m[r][2] = dir[r]; m.m[r][3] = o[r]; }
    return Transform4f(m);
}
// transform.h:180-187
Transform4f Transform4f::perspective(float fov, float near_, float far_) {
    double recip = 1.0 / ((double) far_ - (double) near_), cot = 1.0 / std::tan(((double) (fov / 2.0f)) * (M_PI / 180.0));
    Matrix4f m;
    for (auto &row : m.m) for (auto &v : row) v = 0;
    m.m[0][0] = cot; m.m[1][1] = cot; m.m[2][2] = far_ * recip; m.m[2][3] = -(double) near_ * far_ * recip; m.m[3][2] = 1;
    return Transform4f(m);
}

// ---------------------------------------------------------------- Object
Class *Object::m_class = new Class("Object", "");
const Class *Object::clazz() const { return m_class; }
Object::~Object() {}
void Object::dec_ref(bool dealloc) const noexcept {
    if (--m_ref_count == 0 && dealloc) delete this;
}
std::vector<ref<Object>> Object::expand() const { return {}; }
std::string Object::to_string() const { return std::string(clazz()->name()) + "[]"; }

// ---------------------------------------------------------------- Class registry (class.cpp)
static std::map<std::string, Class *> &class_map() { static std::map<std::string, Class *> m; return m; }
static std::map<std::string, Class *> &alias_map() { static std::map<std::string, Class *> m; return m; }
Class::Class(const std::string &name, const std::string &parent, ConstructFunctor construct, const std::string &alias)
    : m_name(name), m_parent_name(parent), m_alias(alias), m_construct(std::move(construct)) {
    class_map()[name] = this;
    if (!alias.empty()) alias_map()[alias] = this;
}
void Class::static_initialization() {
    for (auto &kv : class_map()) {
        Class *c = kv.second;
        if (!c->m_parent_name.empty()) {
            auto it = class_map().find(c->m_parent_name);
            if (it == class_map().end()) Throw("Critical error during the static RTTI initialization: class \"{}\" not found!", c->m_parent_name);
            c->m_parent = it->second;
        }
    }
    // aliases propagate to derived classes (xml tag of a plugin = alias of its interface class)
}
bool Class::derives_from(const Class *clazz) const {
    const Class *c = this;
    while (c) {
        if (c == clazz) return true;
        if (!c->m_parent && !c->m_parent_name.empty()) {      // registry not linked yet (static init order)
            auto it = class_map().find(c->m_parent_name);
            const_cast<Class *>(c)->m_parent = it == class_map().end() ? nullptr : it->second;
        }
        c = c->m_parent;
    }
    return false;
}
const Class *Class::for_name(const std::string &name) {
    auto it = class_map().find(name);
    return it == class_map().end() ? nullptr : it->second;
}
ref<Object> Class::construct(const Properties &props) const {
    if (!m_construct) Throw("RTTI error: Attempted to construct a non-constructible class \"{}\"!", m_name);
    return m_construct(props);
}
const Class *class_for_tag(const std::string &tag) {     // used by xml.cpp
    auto it = alias_map().find(tag);
    if (it != alias_map().end()) return it->second;
    if (tag == "spectrum") return class_for_tag("texture");
    return nullptr;
}

// ---------------------------------------------------------------- InstanceManager (manager.cpp:13-45)
static std::map<std::string, std::string> &plugin_map() { static std::map<std::string, std::string> m; return m; }
InstanceManager *InstanceManager::get() { static InstanceManager im; return &im; }
void InstanceManager::register_instance(const std::string &class_name, const std::string &instance_name) {
    plugin_map()[instance_name] = class_name;
}
ref<Object> InstanceManager::create_instance(const Properties &props, const Class *clazz) {
    auto it = plugin_map().find(props.instance_name());
    if (it == plugin_map().end()) Throw("Plugin \"{}\" not found!", props.instance_name());
    const Class *plugin_class = Class::for_name(it->second);
    if (!plugin_class) Throw("Plugin \"{}\": class \"{}\" is not registered!", props.instance_name(), it->second);
    if (clazz && !plugin_class->derives_from(clazz))
        Throw("Type mismatch when loading plugin \"{}\": Expected an instance of type \"{}\", got an instance of type \"{}\"",
              props.instance_name(), clazz->name(), plugin_class->name());
    return plugin_class->construct(props);
}

// ---------------------------------------------------------------- Properties (properties.cpp)
Properties::Type Properties::type(const std::string &name) const {
    auto it = m_entries.find(name);
    if (it == m_entries.end()) Throw("type(): Could not find property named \"{}\"!", name);
    return (Type) std::vector<Type>{Type::Bool, Type::Int, Type::Float, Type::String, Type::Vector3, Type::Transform, Type::Color,
                                    Type::NamedReference, Type::Object, Type::Pointer}[it->second.index()];
}
std::vector<std::string> Properties::property_names() const {
    std::vector<std::string> r;
    for (auto &kv : m_entries) r.push_back(kv.first);
    return r;
}
std::vector<std::pair<std::string, NamedReference>> Properties::named_references() const {
    std::vector<std::pair<std::string, NamedReference>> r;
    for (auto &kv : m_entries) if (auto *v = std::get_if<NamedReference>(&kv.second)) r.emplace_back(kv.first, *v);
    return r;
}
std::vector<std::pair<std::string, ref<Object>>> Properties::objects() const {
    std::vector<std::pair<std::string, ref<Object>>> r;
    for (auto &kv : m_entries) if (auto *v = std::get_if<ref<Object>>(&kv.second)) r.emplace_back(kv.first, *v);
    return r;
}
template <typename T> const T &Properties::get(const std::string &n, const char *type_name) const {
    auto it = m_entries.find(n);
    if (it == m_entries.end()) Throw("Property \"{}\" has not been specified!", n);
    const T *v = std::get_if<T>(&it->second);
    if (!v) Throw("The property \"{}\" has the wrong type (expected <{}>).", n, type_name);
    return *v;
}
bool Properties::bool_(const std::string &n) const { return get<bool>(n, "boolean"); }
int Properties::int_(const std::string &n) const { return get<int>(n, "integer"); }
float Properties::float_(const std::string &n) const {
    auto it = m_entries.find(n);
    if (it != m_entries.end()) if (auto *i = std::get_if<int>(&it->second)) return (float) *i;
    return get<float>(n, "float");
}
std::string Properties::string(const std::string &n) const { return get<std::string>(n, "string"); }
Vector3f Properties::vector3(const std::string &n) const { return get<Vector3f>(n, "vector"); }
Color3 Properties::color(const std::string &n) const { return get<Color3>(n, "rgb"); }
Transform4f Properties::transform(const std::string &n) const { return get<Transform4f>(n, "transform"); }
ref<Object> Properties::object(const std::string &n) const { return get<ref<Object>>(n, "object"); }
const void *Properties::pointer(const std::string &n) const { return get<const void *>(n, "pointer"); }

// properties.cpp:190-235
ref<Texture> Properties::texture(const std::string &name) const {
    if (!has_property(name)) Throw("Property \"{}\" has not been specified!", name);
    Type t = type(name);
    if (t == Type::Object) {
        ref<Object> o = object(name);
        if (!o->clazz()->derives_from(MSK_CLASS(Texture)))
            Throw("The property \"{}\" has the wrong type (expected  <spectrum> or <texture>).", name);
        return (Texture *) o.get();
    } else if (t == Type::Float || t == Type::Int) {
        Properties p("srgb");
        float v = float_(name);
        p.set_color("color", Color3{v, v, v});
        return InstanceManager::get()->create_instance<Texture>(p);
    } else if (t == Type::Color) {
        Properties p("srgb");
        p.set_color("color", color(name));
        return InstanceManager::get()->create_instance<Texture>(p);
    }
    Throw("The property \"{}\" has the wrong type (expected  <spectrum> or <texture>).", name);
}
ref<Texture> Properties::texture(const std::string &name, ref<Texture> def_val) const {
    return has_property(name) ? texture(name) : def_val;
}
ref<Texture> Properties::texture(const std::string &name, float def_val) const {
    if (has_property(name)) return texture(name);
    // the reference builds Properties("srgb") with key "value", which its srgb plugin does not
    // read (properties.cpp:229 vs spectra/srgb.cpp:16); the key is "color" here
    Properties p("srgb");
    p.set_color("color", Color3{def_val, def_val, def_val});
    return InstanceManager::get()->create_instance<Texture>(p);
}

}  // namespace misaki
