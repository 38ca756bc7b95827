#include "json.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <sstream>

namespace elaina {

namespace {

struct Parser {
    const std::string &s;
    size_t p = 0;
    explicit Parser(const std::string &text) : s(text) {}

    [[noreturn]] void fail(const std::string &what) const
    {
        size_t line = 1, col = 1;
        for (size_t i = 0; i < p && i < s.size(); ++i) {
            if (s[i] == '\n') { ++line; col = 1; } else ++col;
        }
        throw std::runtime_error("JSON parse error at " + std::to_string(line) + ":" + std::to_string(col) + ": " + what);
    }
    void ws()
    {
        while (p < s.size() && (s[p] == ' ' || s[p] == '\t' || s[p] == '\n' || s[p] == '\r')) ++p;
    }
    bool eat(char c)
    {
        ws();
        if (p < s.size() && s[p] == c) { ++p; return true; }
        return false;
    }
    void expect(char c)
    {
        if (!eat(c)) fail(std::string("expected '") + c + "'");
    }
    json value()
    {
        ws();
        if (p >= s.size()) fail("unexpected end of input");
        const char c = s[p];
        if (c == '{') return object();
        if (c == '[') return array();
        if (c == '"') return json(string());
        if (c == 't' || c == 'f' || c == 'n') return literal();
        return number();
    }
    json literal()
    {
        if (s.compare(p, 4, "true") == 0) { p += 4; return json(true); }
        if (s.compare(p, 5, "false") == 0) { p += 5; return json(false); }
        if (s.compare(p, 4, "null") == 0) { p += 4; return json(nullptr); }
        fail("invalid literal");
    }
    json number()
    {
        const char *b = s.c_str() + p;
        char *e = nullptr;
        const double d = std::strtod(b, &e);
        if (e == b) fail("invalid number");
        p += (size_t)(e - b);
        return json(d);
    }
    static void utf8(std::string &out, unsigned cp)
    {
        if (cp < 0x80) out += (char)cp;
        else if (cp < 0x800) { out += (char)(0xC0 | (cp >> 6)); out += (char)(0x80 | (cp & 0x3F)); }
        else if (cp < 0x10000) { out += (char)(0xE0 | (cp >> 12)); out += (char)(0x80 | ((cp >> 6) & 0x3F)); out += (char)(0x80 | (cp & 0x3F)); }
        else { out += (char)(0xF0 | (cp >> 18)); out += (char)(0x80 | ((cp >> 12) & 0x3F)); out += (char)(0x80 | ((cp >> 6) & 0x3F)); out += (char)(0x80 | (cp & 0x3F)); }
    }
    unsigned hex4()
    {
        if (p + 4 > s.size()) fail("truncated \\u escape");
        unsigned v = 0;
        for (int i = 0; i < 4; ++i) {
            const char c = s[p++];
            v <<= 4;
            if (c >= '0' && c <= '9') v |= (unsigned)(c - '0');
            else if (c >= 'a' && c <= 'f') v |= (unsigned)(c - 'a' + 10);
            else if (c >= 'A' && c <= 'F') v |= (unsigned)(c - 'A' + 10);
            else fail("bad hex digit in \\u escape");
        }
        return v;
    }
    std::string string()
    {
        expect('"');
        std::string out;
        while (true) {
            if (p >= s.size()) fail("unterminated string");
            const char c = s[p++];
            if (c == '"') break;
            if (c != '\\') { out += c; continue; }
            if (p >= s.size()) fail("unterminated escape");
            const char e = s[p++];
            switch (e) {
            case '"': out += '"'; break;
            case '\\': out += '\\'; break;
            case '/': out += '/'; break;
            case 'b': out += '\b'; break;
            case 'f': out += '\f'; break;
            case 'n': out += '\n'; break;
            case 'r': out += '\r'; break;
            case 't': out += '\t'; break;
            case 'u': {
                unsigned cp = hex4();
                if (cp >= 0xD800 && cp <= 0xDBFF && p + 1 < s.size() && s[p] == '\\' && s[p + 1] == 'u') {
                    p += 2;
                    const unsigned lo = hex4();
                    cp = 0x10000 + ((cp - 0xD800) << 10) + (lo - 0xDC00);
                }
                utf8(out, cp);
                break;
            }
            default: fail("unknown escape");
            }
        }
        return out;
    }
    json array()
    {
        expect('[');
        json a = json::array();
        if (eat(']')) return a;
        do { a.push_back(value()); } while (eat(','));
        expect(']');
        return a;
    }
    json object()
    {
        expect('{');
        json o = json::object();
        if (eat('}')) return o;
        do {
            ws();
            const std::string k = string();
            expect(':');
            o[k] = value();
        } while (eat(','));
        expect('}');
        return o;
    }
};

const json kNull;

}  // namespace

json json::parse(const std::string &text)
{
    Parser ps(text);
    json v = ps.value();
    ps.ws();
    if (ps.p != text.size()) ps.fail("trailing characters");
    return v;
}

const json &json::operator[](const std::string &key) const
{
    if (type_ != Type::Object) throw std::runtime_error("json: not an object (key '" + key + "')");
    auto it = obj_.find(key);
    if (it == obj_.end()) throw std::out_of_range("json: missing key '" + key + "'");
    return it->second;
}

json &json::operator[](const std::string &key)
{
    if (type_ == Type::Null) type_ = Type::Object;
    if (type_ != Type::Object) throw std::runtime_error("json: not an object (key '" + key + "')");
    return obj_[key];
}

const json &json::operator[](size_t i) const
{
    if (type_ != Type::Array || i >= arr_.size()) throw std::out_of_range("json: array index out of range");
    return arr_[i];
}

void json::push_back(const json &v)
{
    if (type_ == Type::Null) type_ = Type::Array;
    if (type_ != Type::Array) throw std::runtime_error("json: not an array");
    arr_.push_back(v);
}

template <> bool json::get<bool>() const
{
    if (type_ != Type::Bool) throw std::runtime_error("json: not a boolean");
    return bool_;
}
template <> double json::get<double>() const
{
    if (type_ != Type::Number) throw std::runtime_error("json: not a number");
    return num_;
}
template <> float json::get<float>() const { return (float)get<double>(); }
template <> int json::get<int>() const { return (int)get<double>(); }
template <> unsigned json::get<unsigned>() const { return (unsigned)get<double>(); }
template <> std::string json::get<std::string>() const
{
    if (type_ != Type::String) throw std::runtime_error("json: not a string");
    return str_;
}
template <> json json::get<json>() const { return *this; }
template <> std::vector<float> json::get<std::vector<float>>() const
{
    if (type_ != Type::Array) throw std::runtime_error("json: not an array");
    std::vector<float> v;
    for (const json &e : arr_) v.push_back(e.get<float>());
    return v;
}
template <> std::vector<int> json::get<std::vector<int>>() const
{
    if (type_ != Type::Array) throw std::runtime_error("json: not an array");
    std::vector<int> v;
    for (const json &e : arr_) v.push_back(e.get<int>());
    return v;
}

static void dump_string(std::string &out, const std::string &s)
{
    out += '"';
    for (const char c : s) {
        switch (c) {
        case '"': out += "\\\""; break;
        case '\\': out += "\\\\"; break;
        case '\n': out += "\\n"; break;
        case '\r': out += "\\r"; break;
        case '\t': out += "\\t"; break;
        default:
            if ((unsigned char)c < 0x20) {
                char buf[8];
                std::snprintf(buf, sizeof(buf), "\\u%04x", (unsigned)c);
                out += buf;
            } else out += c;
        }
    }
    out += '"';
}

void json::dump_to(std::string &out, int indent, int depth) const
{
    const bool pretty = indent >= 0;
    auto nl = [&](int d) {
        if (!pretty) return;
        out += '\n';
        out.append((size_t)(indent * d), ' ');
    };
    switch (type_) {
    case Type::Null: out += "null"; break;
    case Type::Bool: out += bool_ ? "true" : "false"; break;
    case Type::Number: {
        char buf[40];
        if (std::isfinite(num_) && num_ == std::floor(num_) && std::fabs(num_) < 1e15) std::snprintf(buf, sizeof(buf), "%.0f", num_);
        else std::snprintf(buf, sizeof(buf), "%.17g", num_);
        out += buf;
        break;
    }
    case Type::String: dump_string(out, str_); break;
    case Type::Array: {
        out += '[';
        bool first = true;
        for (const json &e : arr_) {
            if (!first) out += ',';
            first = false;
            nl(depth + 1);
            e.dump_to(out, indent, depth + 1);
        }
        if (!arr_.empty()) nl(depth);
        out += ']';
        break;
    }
    case Type::Object: {
        out += '{';
        bool first = true;
        for (const auto &kv : obj_) {
            if (!first) out += ',';
            first = false;
            nl(depth + 1);
            dump_string(out, kv.first);
            out += pretty ? ": " : ":";
            kv.second.dump_to(out, indent, depth + 1);
        }
        if (!obj_.empty()) nl(depth);
        out += '}';
        break;
    }
    }
}

std::string json::dump(int indent) const
{
    std::string out;
    dump_to(out, indent, 0);
    return out;
}

const json &get_by_path(const json &j, const std::string &path)
{
    const json *current = &j;
    std::istringstream ss(path);
    std::string token;
    while (std::getline(ss, token, '/')) {
        if (token.empty()) continue;
        if (current->contains(token)) current = &(*current)[token];
        else throw std::out_of_range("Path does not exist: " + path + ", at " + token);
    }
    return *current;
}

json load_json_file(const std::string &file_path)
{
    std::ifstream f(file_path, std::ios::binary);
    if (!f.is_open()) throw std::runtime_error("Failed to open file: " + file_path);
    std::stringstream ss;
    ss << f.rdbuf();
    return json::parse(ss.str());
}

}  // namespace elaina
