// misaki/core.h — host-side object model of the MI355X build: Object / ref<> / Class /
// InstanceManager / Properties / Transform4f, re-authored without Eigen, fmt or pugixml but with
// the reference's names and semantics so plugin code and scene files carry over:
//   Object, ref<T>            include/misaki/core/object.h:31-148
//   Class, MSK_*_CLASS macros include/misaki/core/class.h:9-60, src/librender/class.cpp
//   InstanceManager           include/misaki/core/manager.h:13-44, src/librender/manager.cpp:13-45
//   Properties                include/misaki/core/properties.h:26-216, src/librender/properties.cpp
//   Transform4f               include/misaki/core/transform.h:87-187
//   Throw / Log               include/misaki/core/logger.h:76-88
#pragma once
#include <array>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <functional>
#include <map>
#include <memory>
#include <sstream>
#include <stdexcept>
#include <string>
#include <utility>
#include <variant>
#include <vector>

namespace misaki {

// ---------------------------------------------------------------- errors + logging
enum LogLevel { Trace = 0, Debug, Info, Warn, Error };
void set_log_level(LogLevel level);
void log_message(LogLevel level, const char *file, int line, const std::string &msg);

namespace detail {
inline void fmt_into(std::ostringstream &os, const char *f) { os << f; }
template <typename T, typename... R> void fmt_into(std::ostringstream &os, const char *f, const T &v, const R &...rest) {
    for (; *f; ++f) {
        if (f[0] == '{' && f[1] == '}') { os << v; fmt_into(os, f + 2, rest...); return; }
        os << *f;
    }
}
template <typename... A> std::string format(const char *f, const A &...a) {
    std::ostringstream os; fmt_into(os, f, a...); return os.str();
}
[[noreturn]] void throw_at(const char *file, int line, const std::string &msg);
}  // namespace detail

#define Throw(...) ::misaki::detail::throw_at(__FILE__, __LINE__, ::misaki::detail::format(__VA_ARGS__))
#define Log(level, ...) ::misaki::log_message(::misaki::level, __FILE__, __LINE__, ::misaki::detail::format(__VA_ARGS__))
#define MSK_NOT_IMPLEMENTED(name) Throw("{}::{}(): not implemented!", clazz()->name(), name)

// ---------------------------------------------------------------- small math (host only)
struct Vector3f { float x = 0, y = 0, z = 0; };
struct Vector2i { int x = 0, y = 0; };
struct Color3 { float r = 0, g = 0, b = 0; };

struct Matrix4f {
    double m[4][4];   // host-side set-up math is done in double and rounded when handed to the back end
    static Matrix4f identity();
    Matrix4f operator*(const Matrix4f &o) const;
    Matrix4f inverse() const;
    bool has_nan() const;
};

struct Transform4f {
    Matrix4f m_matrix = Matrix4f::identity(), m_inverse_matrix = Matrix4f::identity();
    Transform4f() {}
    explicit Transform4f(const Matrix4f &m) : m_matrix(m), m_inverse_matrix(m.inverse()) {}
    Transform4f(const Matrix4f &m, const Matrix4f &inv) : m_matrix(m), m_inverse_matrix(inv) {}
    Transform4f operator*(const Transform4f &t) const { return Transform4f(m_matrix * t.m_matrix, t.m_inverse_matrix * m_inverse_matrix); }
    Transform4f inverse() const { return Transform4f(m_inverse_matrix, m_matrix); }
    const Matrix4f &matrix() const { return m_matrix; }
    Vector3f apply_point(const Vector3f &p) const;     // fp32, the reference's operation order
    Vector3f apply_vector(const Vector3f &v) const;
    Vector3f apply_normal(const Vector3f &n) const;
    void to_float16(float out[16]) const;
    static Transform4f translate(const Vector3f &v);
    static Transform4f scale(const Vector3f &v);
    static Transform4f rotate(const Vector3f &axis, float angle_deg);
    static Transform4f lookat(const Vector3f &origin, const Vector3f &target, const Vector3f &up);
    static Transform4f perspective(float fov, float near_, float far_);
};

// ---------------------------------------------------------------- Object + ref<>
class Class;
class Properties;
template <typename T> class ref;

class Object {
public:
    Object() = default;
    Object(const Object &) {}
    void inc_ref() const { ++m_ref_count; }
    void dec_ref(bool dealloc = true) const noexcept;
    int ref_count() const { return m_ref_count; }
    virtual std::vector<ref<Object>> expand() const;
    virtual const Class *clazz() const;
    virtual std::string id() const { return std::string(); }
    virtual std::string to_string() const;
    static Class *m_class;
protected:
    virtual ~Object();
private:
    mutable std::atomic<int> m_ref_count{0};
};

template <typename T> class ref {
public:
    ref() {}
    ref(T *p) : m_ptr(p) { if (m_ptr) ((Object *) m_ptr)->inc_ref(); }
    ref(const ref &r) : m_ptr(r.m_ptr) { if (m_ptr) ((Object *) m_ptr)->inc_ref(); }
    ref(ref &&r) noexcept : m_ptr(r.m_ptr) { r.m_ptr = nullptr; }
    ~ref() { if (m_ptr) ((Object *) m_ptr)->dec_ref(); }
    ref &operator=(const ref &r) {
        if (m_ptr != r.m_ptr) { if (r.m_ptr) ((Object *) r.m_ptr)->inc_ref(); if (m_ptr) ((Object *) m_ptr)->dec_ref(); m_ptr = r.m_ptr; }
        return *this;
    }
    ref &operator=(ref &&r) noexcept { if (&r != this) { if (m_ptr) ((Object *) m_ptr)->dec_ref(); m_ptr = r.m_ptr; r.m_ptr = nullptr; } return *this; }
    T *operator->() const { return m_ptr; }
    T &operator*() const { return *m_ptr; }
    operator T *() const { return m_ptr; }
    T *get() const { return m_ptr; }
    explicit operator bool() const { return m_ptr != nullptr; }
private:
    T *m_ptr = nullptr;
};

// ---------------------------------------------------------------- Class (RTTI) + plugin registry
class Class {
public:
    using ConstructFunctor = std::function<Object *(const Properties &)>;
    Class(const std::string &name, const std::string &parent, ConstructFunctor construct = {}, const std::string &alias = "");
    const std::string &name() const { return m_name; }
    const std::string &alias() const { return m_alias; }
    bool is_constructible() const { return (bool) m_construct; }
    const Class *parent() const { return m_parent; }
    bool derives_from(const Class *clazz) const;
    static const Class *for_name(const std::string &name);
    ref<Object> construct(const Properties &props) const;
    static void static_initialization();
private:
    std::string m_name, m_parent_name, m_alias;
    Class *m_parent = nullptr;
    ConstructFunctor m_construct;
};

#define MSK_CLASS(x) x::m_class
#define MSK_DECLARE_CLASS()                         \
    virtual const Class *clazz() const override;    \
public:                                             \
    static Class *m_class;

namespace detail {
template <typename T, typename = void> struct constructible_from_props : std::false_type {};
template <typename T> struct constructible_from_props<T, std::void_t<decltype(new T(std::declval<const Properties &>()))>> : std::true_type {};
template <typename T> Class::ConstructFunctor construct_functor() {
    if constexpr (constructible_from_props<T>::value && !std::is_abstract<T>::value)
        return [](const Properties &p) -> Object * { return new T(p); };
    else
        return {};
}
}  // namespace detail

#define MSK_IMPLEMENT_CLASS(Name, Parent, ...)                                                                       \
    Class *Name::m_class = new Class(#Name, #Parent, ::misaki::detail::construct_functor<Name>(), ##__VA_ARGS__);    \
    const Class *Name::clazz() const { return m_class; }

class InstanceManager {
public:
    static InstanceManager *get();
    ref<Object> create_instance(const Properties &, const Class *);
    template <typename T> ref<T> create_instance(const Properties &props) {
        return static_cast<T *>(create_instance(props, MSK_CLASS(T)).get());
    }
    void register_instance(const std::string &class_name, const std::string &instance_name);
};

#define MSK_REGISTER_INSTANCE(ClassName, InstanceName)                                                   \
    static struct Instance_##ClassName {                                                                 \
        Instance_##ClassName() { InstanceManager::get()->register_instance(#ClassName, InstanceName); }  \
    } instance_##ClassName;

// ---------------------------------------------------------------- Properties
class NamedReference {
public:
    NamedReference(const std::string &value) : m_value(value) {}
    operator const std::string &() const { return m_value; }
    bool operator==(const NamedReference &r) const { return r.m_value == m_value; }
private:
    std::string m_value;
};

class Texture;

class Properties {
public:
    enum class Type { Bool, Int, Float, Vector3, Transform, Color, String, NamedReference, Object, Pointer };
    Properties() {}
    explicit Properties(const std::string &instance_name) : m_instance_name(instance_name) {}

    const std::string &instance_name() const { return m_instance_name; }
    void set_instance_name(const std::string &name) { m_instance_name = name; }
    bool has_property(const std::string &name) const { return m_entries.count(name) != 0; }
    Type type(const std::string &name) const;
    const std::string &id() const { return m_id; }
    void set_id(const std::string &id) { m_id = id; }
    std::vector<std::string> property_names() const;
    std::vector<std::pair<std::string, NamedReference>> named_references() const;
    std::vector<std::pair<std::string, ref<Object>>> objects() const;   // sorted by name, like std::map

    void set_bool(const std::string &n, bool v, bool warn = true) { set(n, v, warn); }
    void set_int(const std::string &n, int v, bool warn = true) { set(n, v, warn); }
    void set_float(const std::string &n, float v, bool warn = true) { set(n, v, warn); }
    void set_string(const std::string &n, const std::string &v, bool warn = true) { set(n, v, warn); }
    void set_named_reference(const std::string &n, const NamedReference &v, bool warn = true) { set(n, v, warn); }
    void set_vector3(const std::string &n, const Vector3f &v, bool warn = true) { set(n, v, warn); }
    void set_color(const std::string &n, const Color3 &v, bool warn = true) { set(n, v, warn); }
    void set_transform(const std::string &n, const Transform4f &v, bool warn = true) { set(n, v, warn); }
    void set_object(const std::string &n, const ref<Object> &v, bool warn = true) { set(n, v, warn); }
    void set_pointer(const std::string &n, const void *v, bool warn = true) { set(n, v, warn); }

    bool bool_(const std::string &n) const;
    bool bool_(const std::string &n, bool def) const { return has_property(n) ? bool_(n) : def; }
    int int_(const std::string &n) const;
    int int_(const std::string &n, int def) const { return has_property(n) ? int_(n) : def; }
    float float_(const std::string &n) const;
    float float_(const std::string &n, float def) const { return has_property(n) ? float_(n) : def; }
    std::string string(const std::string &n) const;
    std::string string(const std::string &n, const std::string &def) const { return has_property(n) ? string(n) : def; }
    Vector3f vector3(const std::string &n) const;
    Vector3f vector3(const std::string &n, const Vector3f &def) const { return has_property(n) ? vector3(n) : def; }
    Color3 color(const std::string &n) const;
    Color3 color(const std::string &n, const Color3 &def) const { return has_property(n) ? color(n) : def; }
    Transform4f transform(const std::string &n) const;
    Transform4f transform(const std::string &n, const Transform4f &def) const { return has_property(n) ? transform(n) : def; }
    ref<Object> object(const std::string &n) const;
    const void *pointer(const std::string &n) const;

    // src/librender/properties.cpp:190-235
    ref<Texture> texture(const std::string &name) const;
    ref<Texture> texture(const std::string &name, ref<Texture> def_val) const;
    ref<Texture> texture(const std::string &name, float def_val) const;

private:
    using Value = std::variant<bool, int, float, std::string, Vector3f, Transform4f, Color3, NamedReference, ref<Object>, const void *>;
    template <typename T> void set(const std::string &n, const T &v, bool warn) {
        if (warn && has_property(n)) log_message(Warn, __FILE__, __LINE__, "Property \"" + n + "\" was specified multiple times!");
        m_entries.erase(n);
        m_entries.emplace(n, Value(v));
    }
    template <typename T> const T &get(const std::string &n, const char *type_name) const;
    std::map<std::string, Value> m_entries;
    std::string m_id, m_instance_name;
};

// ---------------------------------------------------------------- XML (include/misaki/core/xml.h:9-10)
namespace xml {
using ParameterList = std::vector<std::pair<std::string, std::string>>;
ref<Object> load_file(const std::string &filename, ParameterList parameters = {});
ref<Object> load_string(const std::string &text, const std::string &base_dir = ".", ParameterList parameters = {});
}  // namespace xml

// file resolver (include/misaki/core/fresolver.h): search paths for relative file names
class FileResolver {
public:
    void prepend(const std::string &dir) { m_paths.insert(m_paths.begin(), dir); }
    std::string resolve(const std::string &name) const;
private:
    std::vector<std::string> m_paths;
};
FileResolver *get_file_resolver();

namespace string {
std::vector<std::string> tokenize(const std::string &s, const std::string &delim = ", ");
std::string to_lower(const std::string &s);
std::string indent(const std::string &s, int amount = 2);
}  // namespace string

}  // namespace misaki
