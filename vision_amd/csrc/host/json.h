// Minimal JSON reader for Vision scene files.  Vision strips `//` and `/* */` comments before parsing
// (reference: src/base/import/json_util.h:12-112); so does this reader (outside string literals).
#pragma once
#include <cstdlib>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

namespace vmk {

class Json {
public:
    enum Type { Null, Bool, Number, String, Array, Object };
    Type type{Null};
    bool b{false};
    double num{0.0};
    std::string str;
    std::vector<Json> arr;
    std::vector<std::pair<std::string, Json>> obj; // insertion order kept (node order matters for lights/shapes)

    bool is_null() const { return type == Null; }
    bool is_number() const { return type == Number; }
    bool is_array() const { return type == Array; }
    bool is_object() const { return type == Object; }
    bool is_string() const { return type == String; }
    bool contains(const std::string &k) const { return find(k) != nullptr; }
    const Json *find(const std::string &k) const {
        if (type != Object) return nullptr;
        for (auto &kv : obj) if (kv.first == k) return &kv.second;
        return nullptr;
    }
    // Vision's ParameterSet semantics: a missing key yields a null node whose as_xxx() return the default
    const Json &operator[](const std::string &k) const {
        static const Json null_node;
        const Json *p = find(k);
        return p ? *p : null_node;
    }
    const Json &at(size_t i) const { static const Json null_node; return (type == Array && i < arr.size()) ? arr[i] : null_node; }
    size_t size() const { return type == Array ? arr.size() : (type == Object ? obj.size() : 0); }
    double as_double(double d) const { return type == Number ? num : (type == Bool ? (b ? 1.0 : 0.0) : d); }
    float as_float(float d) const { return (float) as_double(d); }
    uint32_t as_uint(uint32_t d) const { return type == Number ? (uint32_t) num : (type == Bool ? (uint32_t) b : d); }
    int as_int(int d) const { return type == Number ? (int) num : d; }
    bool as_bool(bool d) const { return type == Bool ? b : (type == Number ? num != 0.0 : d); }
    std::string as_string(const std::string &d = "") const { return type == String ? str : d; }
    std::vector<float> as_float_vector() const {
        std::vector<float> v;
        if (type == Array) for (auto &e : arr) v.push_back(e.as_float(0.f));
        else if (type == Number) v.push_back((float) num);
        return v;
    }

    static std::string strip_comments(const std::string &s) {
        std::string o;
        o.reserve(s.size());
        bool in_str = false;
        for (size_t i = 0; i < s.size(); ++i) {
            char c = s[i];
            if (in_str) {
                o.push_back(c);
                if (c == '\\' && i + 1 < s.size()) { o.push_back(s[++i]); continue; }
                if (c == '"') in_str = false;
                continue;
            }
            if (c == '"') { in_str = true; o.push_back(c); continue; }
            if (c == '/' && i + 1 < s.size() && s[i + 1] == '/') { while (i < s.size() && s[i] != '\n') ++i; o.push_back('\n'); continue; }
            if (c == '/' && i + 1 < s.size() && s[i + 1] == '*') { i += 2; while (i + 1 < s.size() && !(s[i] == '*' && s[i + 1] == '/')) ++i; ++i; continue; }
            o.push_back(c);
        }
        return o;
    }
    static Json parse(const std::string &text) {
        std::string s = strip_comments(text);
        size_t p = 0;
        Json j = parse_value(s, p);
        skip_ws(s, p);
        if (p != s.size()) throw std::runtime_error("json: trailing characters at offset " + std::to_string(p));
        return j;
    }

private:
    static void skip_ws(const std::string &s, size_t &p) { while (p < s.size() && (s[p] == ' ' || s[p] == '\t' || s[p] == '\n' || s[p] == '\r')) ++p; }
    static Json parse_value(const std::string &s, size_t &p) {
        skip_ws(s, p);
        if (p >= s.size()) throw std::runtime_error("json: unexpected end");
        char c = s[p];
        Json j;
        if (c == '{') {
            j.type = Object; ++p; skip_ws(s, p);
            if (p < s.size() && s[p] == '}') { ++p; return j; }
            for (;;) {
                skip_ws(s, p);
                if (p >= s.size() || s[p] != '"') throw std::runtime_error("json: expected key at offset " + std::to_string(p));
                std::string k = parse_string(s, p);
                skip_ws(s, p);
                if (p >= s.size() || s[p] != ':') throw std::runtime_error("json: expected ':' at offset " + std::to_string(p));
                ++p;
                Json v = parse_value(s, p);
                bool replaced = false;
                for (auto &kv : j.obj) if (kv.first == k) { kv.second = v; replaced = true; }
                if (!replaced) j.obj.emplace_back(k, std::move(v));
                skip_ws(s, p);
                if (p < s.size() && s[p] == ',') { ++p; skip_ws(s, p); if (p < s.size() && s[p] == '}') { ++p; return j; } continue; }
                if (p < s.size() && s[p] == '}') { ++p; return j; }
                throw std::runtime_error("json: expected ',' or '}' at offset " + std::to_string(p));
            }
        }
        if (c == '[') {
            j.type = Array; ++p; skip_ws(s, p);
            if (p < s.size() && s[p] == ']') { ++p; return j; }
            for (;;) {
                j.arr.push_back(parse_value(s, p));
                skip_ws(s, p);
                if (p < s.size() && s[p] == ',') { ++p; skip_ws(s, p); if (p < s.size() && s[p] == ']') { ++p; return j; } continue; }
                if (p < s.size() && s[p] == ']') { ++p; return j; }
                throw std::runtime_error("json: expected ',' or ']' at offset " + std::to_string(p));
            }
        }
        if (c == '"') { j.type = String; j.str = parse_string(s, p); return j; }
        if (s.compare(p, 4, "true") == 0) { j.type = Bool; j.b = true; p += 4; return j; }
        if (s.compare(p, 5, "false") == 0) { j.type = Bool; j.b = false; p += 5; return j; }
        if (s.compare(p, 4, "null") == 0) { p += 4; return j; }
        char *end = nullptr;
        double v = std::strtod(s.c_str() + p, &end);
        if (end == s.c_str() + p) throw std::runtime_error("json: bad token at offset " + std::to_string(p));
        j.type = Number; j.num = v; p = (size_t) (end - s.c_str());
        return j;
    }
    static std::string parse_string(const std::string &s, size_t &p) {
        std::string o; ++p;
        while (p < s.size() && s[p] != '"') {
            if (s[p] == '\\' && p + 1 < s.size()) {
                char e = s[++p];
                switch (e) { case 'n': o.push_back('\n'); break; case 't': o.push_back('\t'); break; case 'r': o.push_back('\r'); break;
                    case 'b': o.push_back('\b'); break; case 'f': o.push_back('\f'); break;
                    case 'u': { if (p + 4 < s.size()) { unsigned cp = (unsigned) std::strtoul(s.substr(p + 1, 4).c_str(), nullptr, 16); p += 4; if (cp < 0x80) o.push_back((char) cp); else o.push_back('?'); } break; }
                    default: o.push_back(e); }
                ++p;
            } else o.push_back(s[p++]);
        }
        if (p >= s.size()) throw std::runtime_error("json: unterminated string");
        ++p;
        return o;
    }
};

}// namespace vmk
