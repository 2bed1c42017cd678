// Minimal JSON DOM for the reference's two input formats (config.json, scene JSON).
// The reference reads them with nlohmann/json 3.7.3 (thirdparty/json.hpp); only the accessors it uses are
// mirrored: operator[] / contains / value(key, default) / at(i) / get<T>() / is_array().
#pragma once
#include <cstdint>
#include <cstdlib>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

namespace pth {

struct JsonError : std::runtime_error {
    using std::runtime_error::runtime_error;
};

class Json {
public:
    enum Type { Null, Bool, Int, Float, String, Array, Object };
    Type type = Null;
    bool b = false;
    int64_t i = 0;
    double d = 0.0;
    std::string s;
    std::vector<Json> arr;
    std::vector<std::pair<std::string, Json>> obj;   // file order preserved

    bool is_null() const { return type == Null; }
    bool is_array() const { return type == Array; }
    bool is_object() const { return type == Object; }
    bool is_number() const { return type == Int || type == Float; }
    bool is_string() const { return type == String; }

    bool contains(const std::string &k) const
    {
        if (type != Object) return false;
        for (auto &kv : obj)
            if (kv.first == k) return true;
        return false;
    }
    // const operator[]: missing key -> null (nlohmann inserts null on the non-const path)
    const Json &operator[](const std::string &k) const
    {
        static const Json null_;
        if (type != Object) {
            if (type == Null) return null_;
            throw JsonError("operator[] with a string key on a non-object");
        }
        for (auto &kv : obj)
            if (kv.first == k) return kv.second;
        return null_;
    }
    const Json &at(size_t idx) const
    {
        if (type != Array) throw JsonError("at() on a non-array");
        if (idx >= arr.size()) throw JsonError("array index out of range");
        return arr[idx];
    }
    size_t size() const { return type == Array ? arr.size() : (type == Object ? obj.size() : 0); }

    double as_double() const
    {
        if (type == Int) return (double)i;
        if (type == Float) return d;
        if (type == Bool) return b ? 1.0 : 0.0;
        throw JsonError("type must be number");
    }
    float as_float() const { return (float)as_double(); }   // get<float>(): static_cast from the stored number
    int as_int() const
    {
        if (type == Int) return (int)i;
        if (type == Float) return (int)d;
        if (type == Bool) return b ? 1 : 0;
        throw JsonError("type must be number");
    }
    bool as_bool() const
    {
        if (type == Bool) return b;
        throw JsonError("type must be boolean");
    }
    const std::string &as_string() const
    {
        if (type != String) throw JsonError("type must be string");
        return s;
    }
    // value(key, default): default iff the key is absent; a present key of the wrong type throws
    double value(const std::string &k, double def) const { require_object(); return contains(k) ? (*this)[k].as_double() : def; }
    int value_int(const std::string &k, int def) const { require_object(); return contains(k) ? (*this)[k].as_int() : def; }
    bool value_bool(const std::string &k, bool def) const { require_object(); return contains(k) ? (*this)[k].as_bool() : def; }
    std::string value_str(const std::string &k, const std::string &def) const
    {
        require_object();
        return contains(k) ? (*this)[k].as_string() : def;
    }

    static Json parse(const std::string &text);

private:
    void require_object() const
    {
        if (type != Object) throw JsonError("cannot use value() on a non-object");
    }
};

namespace detail {
struct Parser {
    const std::string &t;
    size_t p = 0;
    explicit Parser(const std::string &text) : t(text) {}
    [[noreturn]] void fail(const char *m) { throw JsonError(std::string("JSON parse error at byte ") + std::to_string(p) + ": " + m); }
    void ws()
    {
        while (p < t.size() && (t[p] == ' ' || t[p] == '\n' || t[p] == '\r' || t[p] == '\t')) p++;
    }
    Json value()
    {
        ws();
        if (p >= t.size()) fail("unexpected end");
        char c = t[p];
        if (c == '{') return object();
        if (c == '[') return array();
        if (c == '"') { Json j; j.type = Json::String; j.s = string(); return j; }
        if (c == 't' && t.compare(p, 4, "true") == 0) { p += 4; Json j; j.type = Json::Bool; j.b = true; return j; }
        if (c == 'f' && t.compare(p, 5, "false") == 0) { p += 5; Json j; j.type = Json::Bool; j.b = false; return j; }
        if (c == 'n' && t.compare(p, 4, "null") == 0) { p += 4; return Json(); }
        return number();
    }
    Json number()
    {
        size_t s0 = p;
        bool is_float = false;
        if (p < t.size() && t[p] == '-') p++;
        while (p < t.size() && ((t[p] >= '0' && t[p] <= '9') || t[p] == '.' || t[p] == 'e' || t[p] == 'E' || t[p] == '+' || t[p] == '-')) {
            if (t[p] == '.' || t[p] == 'e' || t[p] == 'E') is_float = true;
            p++;
        }
        if (p == s0) fail("invalid value");
        std::string tok = t.substr(s0, p - s0);
        Json j;
        if (is_float) { j.type = Json::Float; j.d = strtod(tok.c_str(), nullptr); }
        else { j.type = Json::Int; j.i = strtoll(tok.c_str(), nullptr, 10); j.d = (double)j.i; }
        return j;
    }
    std::string string()
    {
        std::string out;
        p++;   // opening quote
        while (p < t.size() && t[p] != '"') {
            char c = t[p++];
            if (c == '\\') {
                if (p >= t.size()) fail("bad escape");
                char e = t[p++];
                switch (e) {
                case 'n': out += '\n'; break;
                case 't': out += '\t'; break;
                case 'r': out += '\r'; break;
                case 'b': out += '\b'; break;
                case 'f': out += '\f'; break;
                case 'u': {
                    if (p + 4 > t.size()) fail("bad \\u escape");
                    unsigned cp = (unsigned)strtoul(t.substr(p, 4).c_str(), nullptr, 16);
                    p += 4;
                    if (cp < 0x80) out += (char)cp;
                    else if (cp < 0x800) { out += (char)(0xC0 | (cp >> 6)); out += (char)(0x80 | (cp & 0x3F)); }
                    else { out += (char)(0xE0 | (cp >> 12)); out += (char)(0x80 | ((cp >> 6) & 0x3F)); out += (char)(0x80 | (cp & 0x3F)); }
                    break;
                }
                default: out += e;
                }
            } else out += c;
        }
        if (p >= t.size()) fail("unterminated string");
        p++;
        return out;
    }
    Json array()
    {
        Json j;
        j.type = Json::Array;
        p++;
        ws();
        if (p < t.size() && t[p] == ']') { p++; return j; }
        for (;;) {
            j.arr.push_back(value());
            ws();
            if (p >= t.size()) fail("unterminated array");
            if (t[p] == ',') { p++; continue; }
            if (t[p] == ']') { p++; break; }
            fail("expected , or ]");
        }
        return j;
    }
    Json object()
    {
        Json j;
        j.type = Json::Object;
        p++;
        ws();
        if (p < t.size() && t[p] == '}') { p++; return j; }
        for (;;) {
            ws();
            if (p >= t.size() || t[p] != '"') fail("expected string key");
            std::string k = string();
            ws();
            if (p >= t.size() || t[p] != ':') fail("expected :");
            p++;
            Json v = value();
            bool dup = false;
            for (auto &kv : j.obj)
                if (kv.first == k) { kv.second = v; dup = true; }   // last duplicate wins, as in nlohmann
            if (!dup) j.obj.emplace_back(k, std::move(v));
            ws();
            if (p >= t.size()) fail("unterminated object");
            if (t[p] == ',') { p++; continue; }
            if (t[p] == '}') { p++; break; }
            fail("expected , or }");
        }
        return j;
    }
};
}  // namespace detail

inline Json Json::parse(const std::string &text)
{
    detail::Parser ps(text);
    Json j = ps.value();
    ps.ws();
    if (ps.p != text.size()) ps.fail("trailing characters");
    return j;
}

}  // namespace pth
