// json_min.h -- small JSON reader/writer for the configuration strings and files of the manager and its plugins.
// The reference parses these with nlohmann_json (/root/reference/src/Manager/SlamManager.cpp:613-1003,
// src/Utils/ConfigOptions.h:211-296); only what those call sites need is implemented: objects, arrays, strings,
// numbers, booleans, null; parse errors throw JsonError.
#pragma once
#include <cmath>
#include <cstdlib>
#include <map>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

namespace LpSlam {

struct JsonError : std::runtime_error { using std::runtime_error::runtime_error; };

class Json {
public:
    enum Kind { Null, Bool, Number, String, Array, Object };
    Kind kind = Null;
    bool b = false;
    double num = 0;
    std::string str;
    std::vector<Json> arr;
    std::vector<std::pair<std::string, Json>> obj;     // insertion order kept

    bool isObject() const { return kind == Object; }
    bool isArray() const { return kind == Array; }
    const Json* find(const std::string& key) const {
        if (kind != Object) return nullptr;
        for (auto& kv : obj) if (kv.first == key) return &kv.second;
        return nullptr;
    }
    double asNumber() const { if (kind != Number) throw JsonError("number expected"); return num; }
    bool asBool() const { if (kind != Bool) throw JsonError("boolean expected"); return b; }
    const std::string& asString() const { if (kind != String) throw JsonError("string expected"); return str; }

    static Json parse(const std::string& text) {
        size_t pos = 0;
        Json v = parseValue(text, pos);
        skipWs(text, pos);
        if (pos != text.size()) throw JsonError("trailing characters after JSON value");
        return v;
    }

    std::string dump() const {
        std::ostringstream o;
        write(o);
        return o.str();
    }

private:
    static void skipWs(const std::string& s, size_t& p) { while (p < s.size() && (s[p] == ' ' || s[p] == '\t' || s[p] == '\n' || s[p] == '\r')) ++p; }
    static Json parseValue(const std::string& s, size_t& p) {
        skipWs(s, p);
        if (p >= s.size()) throw JsonError("unexpected end of JSON");
        Json v;
        const char c = s[p];
        if (c == '{') {
            v.kind = Object; ++p; skipWs(s, p);
            if (p < s.size() && s[p] == '}') { ++p; return v; }
            for (;;) {
                skipWs(s, p);
                if (p >= s.size() || s[p] != '"') throw JsonError("object key expected");
                std::string key = parseString(s, p);
                skipWs(s, p);
                if (p >= s.size() || s[p] != ':') throw JsonError("':' expected");
                ++p;
                v.obj.emplace_back(key, parseValue(s, p));
                skipWs(s, p);
                if (p < s.size() && s[p] == ',') { ++p; continue; }
                if (p < s.size() && s[p] == '}') { ++p; return v; }
                throw JsonError("',' or '}' expected");
            }
        }
        if (c == '[') {
            v.kind = Array; ++p; skipWs(s, p);
            if (p < s.size() && s[p] == ']') { ++p; return v; }
            for (;;) {
                v.arr.push_back(parseValue(s, p));
                skipWs(s, p);
                if (p < s.size() && s[p] == ',') { ++p; continue; }
                if (p < s.size() && s[p] == ']') { ++p; return v; }
                throw JsonError("',' or ']' expected");
            }
        }
        if (c == '"') { v.kind = String; v.str = parseString(s, p); return v; }
        if (s.compare(p, 4, "true") == 0) { v.kind = Bool; v.b = true; p += 4; return v; }
        if (s.compare(p, 5, "false") == 0) { v.kind = Bool; v.b = false; p += 5; return v; }
        if (s.compare(p, 4, "null") == 0) { p += 4; return v; }
        if (c == '-' || (c >= '0' && c <= '9')) {
            const char* b0 = s.c_str() + p;
            char* e = nullptr;
            v.num = std::strtod(b0, &e);
            if (e == b0) throw JsonError("bad number");
            v.kind = Number; p += (size_t)(e - b0);
            return v;
        }
        throw JsonError(std::string("unexpected character '") + c + "'");
    }
    static std::string parseString(const std::string& s, size_t& p) {
        std::string out;
        ++p;
        while (p < s.size() && s[p] != '"') {
            if (s[p] == '\\') {
                ++p;
                if (p >= s.size()) break;
                switch (s[p]) {
                case 'n': out += '\n'; break; case 't': out += '\t'; break; case 'r': out += '\r'; break;
                case 'b': out += '\b'; break; case 'f': out += '\f'; break;
                case 'u': { if (p + 4 >= s.size()) throw JsonError("bad \\u escape"); out += (char)std::strtol(s.substr(p + 1, 4).c_str(), nullptr, 16); p += 4; break; }
                default: out += s[p];
                }
                ++p;
            } else out += s[p++];
        }
        if (p >= s.size()) throw JsonError("unterminated string");
        ++p;
        return out;
    }
    void write(std::ostringstream& o) const {
        switch (kind) {
        case Null: o << "null"; break;
        case Bool: o << (b ? "true" : "false"); break;
        case Number: if (num == std::floor(num) && std::fabs(num) < 1e15) o << (long long)num; else { o.precision(17); o << num; } break;
        case String: o << '"'; for (char c : str) { if (c == '"' || c == '\\') o << '\\'; o << c; } o << '"'; break;
        case Array: o << '['; for (size_t i = 0; i < arr.size(); ++i) { if (i) o << ','; arr[i].write(o); } o << ']'; break;
        case Object: o << '{'; for (size_t i = 0; i < obj.size(); ++i) { if (i) o << ','; o << '"' << obj[i].first << "\":"; obj[i].second.write(o); } o << '}'; break;
        }
    }
};

// Typed plugin options with defaults (cf. ConfigOptions of the reference): unknown keys are rejected, keys starting with
// '_' are comments.
class ConfigOptions {
public:
    void optional(const std::string& key, bool v) { Json j; j.kind = Json::Bool; j.b = v; set(key, j); }
    void optional(const std::string& key, int v) { Json j; j.kind = Json::Number; j.num = v; set(key, j); ints_[key] = true; }
    void optional(const std::string& key, double v) { Json j; j.kind = Json::Number; j.num = v; set(key, j); }
    void optional(const std::string& key, const std::string& v) { Json j; j.kind = Json::String; j.str = v; set(key, j); }
    void optional(const std::string& key, const char* v) { optional(key, std::string(v)); }

    void parse(const std::string& text) {
        if (text.empty()) return;
        Json j = Json::parse(text);
        if (!j.isObject()) throw std::invalid_argument("configuration must be a JSON object");
        for (auto& kv : j.obj) {
            if (!kv.first.empty() && kv.first[0] == '_') continue;
            auto it = values_.find(kv.first);
            if (it == values_.end()) throw std::invalid_argument("unknown configuration option '" + kv.first + "'");
            if (it->second.kind != kv.second.kind) throw std::invalid_argument("configuration option '" + kv.first + "' has the wrong type");
            it->second = kv.second;
        }
    }
    bool getBool(const std::string& k) const { return at(k).asBool(); }
    int getInteger(const std::string& k) const { return (int)std::llround(at(k).asNumber()); }
    double getDouble(const std::string& k) const { return at(k).asNumber(); }
    std::string getString(const std::string& k) const { return at(k).asString(); }

private:
    void set(const std::string& k, const Json& j) { values_[k] = j; }
    const Json& at(const std::string& k) const {
        auto it = values_.find(k);
        if (it == values_.end()) throw std::invalid_argument("option '" + k + "' was never declared");
        return it->second;
    }
    std::map<std::string, Json> values_;
    std::map<std::string, bool> ints_;
};

}  // namespace LpSlam
