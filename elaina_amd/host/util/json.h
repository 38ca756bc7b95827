// json.h -- minimal JSON value + recursive-descent parser + writer for the host side.
// Replaces the reference's use of nlohmann::json (ext/json, an empty submodule in the
// reference checkout; the only nlohmann header in this image is not present on the GPU
// box) for exactly what run_expr / Problem / the integrator settings need: objects,
// arrays, strings, numbers, booleans, null, '/'-separated path lookups with the
// reference's throw / optional semantics (core/common.h:127-213).
#pragma once
#include <cstdint>
#include <map>
#include <memory>
#include <optional>
#include <stdexcept>
#include <string>
#include <vector>

namespace elaina {

class json {
public:
    enum class Type { Null, Bool, Number, String, Array, Object };

    json() = default;
    json(std::nullptr_t) {}
    json(bool b) : type_(Type::Bool), bool_(b) {}
    json(double d) : type_(Type::Number), num_(d) {}
    json(int d) : type_(Type::Number), num_(d) {}
    json(int64_t d) : type_(Type::Number), num_((double)d) {}
    json(uint64_t d) : type_(Type::Number), num_((double)d) {}
    json(const char *s) : type_(Type::String), str_(s) {}
    json(const std::string &s) : type_(Type::String), str_(s) {}

    static json array() { json j; j.type_ = Type::Array; return j; }
    static json object() { json j; j.type_ = Type::Object; return j; }
    static json parse(const std::string &text);   // throws std::runtime_error with line:col

    Type type() const { return type_; }
    bool is_null() const { return type_ == Type::Null; }
    bool is_array() const { return type_ == Type::Array; }
    bool is_object() const { return type_ == Type::Object; }
    bool is_number() const { return type_ == Type::Number; }
    bool is_string() const { return type_ == Type::String; }
    bool is_bool() const { return type_ == Type::Bool; }

    bool contains(const std::string &key) const { return type_ == Type::Object && obj_.count(key) != 0; }
    size_t size() const { return type_ == Type::Array ? arr_.size() : type_ == Type::Object ? obj_.size() : 0; }
    const json &operator[](const std::string &key) const;
    json &operator[](const std::string &key);   // creates (turns Null into Object)
    const json &operator[](size_t i) const;
    void push_back(const json &v);
    const std::vector<json> &items() const { return arr_; }
    const std::map<std::string, json> &members() const { return obj_; }

    template <typename T> T get() const;
    std::string dump(int indent = -1) const;

private:
    void dump_to(std::string &out, int indent, int depth) const;
    Type type_ = Type::Null;
    bool bool_ = false;
    double num_ = 0.0;
    std::string str_;
    std::vector<json> arr_;
    std::map<std::string, json> obj_;
};

template <> bool json::get<bool>() const;
template <> int json::get<int>() const;
template <> unsigned json::get<unsigned>() const;
template <> float json::get<float>() const;
template <> double json::get<double>() const;
template <> std::string json::get<std::string>() const;
template <> json json::get<json>() const;
template <> std::vector<float> json::get<std::vector<float>>() const;
template <> std::vector<int> json::get<std::vector<int>>() const;

// reference core/common.h:127-213
const json &get_by_path(const json &j, const std::string &path);

template <typename T> T json_get_or_throw(const json &j, const std::string &path)
{
    try {
        const json &value = get_by_path(j, path);
        if (!value.is_null()) return value.get<T>();
        throw std::runtime_error("Path value is null: " + path);
    } catch (const std::exception &e) {
        throw std::runtime_error("Failed to find json value. " + std::string(e.what()));
    }
}

template <typename T> T json_get_optional(const json &j, const std::string &path, const T &default_value)
{
    try {
        const json &value = get_by_path(j, path);
        if (!value.is_null()) return value.get<T>();
        return default_value;
    } catch (const std::exception &) {
        return default_value;
    }
}

template <typename T> std::optional<T> json_get_optional(const json &j, const std::string &path)
{
    try {
        const json &value = get_by_path(j, path);
        if (!value.is_null()) return value.get<T>();
        return {};
    } catch (const std::exception &) {
        return {};
    }
}

json load_json_file(const std::string &file_path);

}  // namespace elaina
