// toml_lite.hpp -- the subset of TOML that ReadBouncer's config.toml uses (the reference vendors
// toml11, src/toml11; it cannot be fetched here): [tables], key = value with basic/literal strings,
// integers, floats, booleans, one-line or multi-line arrays of those, '#' comments.
#pragma once
#include <cctype>
#include <fstream>
#include <map>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

namespace toml_lite
{

struct Value
{
    enum Kind { String, Integer, Float, Boolean, Array } kind = String;
    std::string s;
    long long i = 0;
    double f = 0.0;
    bool b = false;
    std::vector<Value> arr;
};

class Document
{
    std::map<std::string, Value> kv_;  // "table.key" -> value ("" table for top level)

    static void skip_ws(const std::string& t, size_t& p, bool newlines)
    {
        while (p < t.size()) {
            if (t[p] == ' ' || t[p] == '\t' || t[p] == '\r' || (newlines && t[p] == '\n')) ++p;
            else if (t[p] == '#') while (p < t.size() && t[p] != '\n') ++p;
            else break;
            if (!newlines && p < t.size() && t[p] == '\n') break;
        }
    }

    static Value parse_value(const std::string& t, size_t& p)
    {
        Value v;
        if (p >= t.size()) throw std::runtime_error("toml: value expected");
        if (t[p] == '"' || t[p] == '\'') {
            const char q = t[p++];
            std::string out;
            while (p < t.size() && t[p] != q) {
                if (q == '"' && t[p] == '\\' && p + 1 < t.size()) {
                    ++p;
                    switch (t[p]) {
                    case 'n': out += '\n'; break;
                    case 't': out += '\t'; break;
                    case '\\': out += '\\'; break;
                    case '"': out += '"'; break;
                    default: out += t[p];
                    }
                    ++p;
                } else {
                    if (t[p] == '\n') throw std::runtime_error("toml: unterminated string");
                    out += t[p++];
                }
            }
            if (p >= t.size()) throw std::runtime_error("toml: unterminated string");
            ++p;
            v.kind = Value::String;
            v.s = out;
            return v;
        }
        if (t[p] == '[') {
            ++p;
            v.kind = Value::Array;
            for (;;) {
                skip_ws(t, p, true);
                if (p >= t.size()) throw std::runtime_error("toml: unterminated array");
                if (t[p] == ']') { ++p; break; }
                v.arr.push_back(parse_value(t, p));
                skip_ws(t, p, true);
                if (p < t.size() && t[p] == ',') ++p;
            }
            return v;
        }
        size_t e = p;
        while (e < t.size() && t[e] != '\n' && t[e] != '#' && t[e] != ',' && t[e] != ']') ++e;
        std::string tok = t.substr(p, e - p);
        while (!tok.empty() && std::isspace((unsigned char)tok.back())) tok.pop_back();
        p = e;
        if (tok == "true" || tok == "false") {
            v.kind = Value::Boolean;
            v.b = tok == "true";
            return v;
        }
        std::string clean;
        for (char c : tok) if (c != '_') clean += c;
        if (clean.empty()) throw std::runtime_error("toml: empty value");
        size_t used = 0;
        if (clean.find_first_of(".eE") == std::string::npos || clean.compare(0, 2, "0x") == 0) {
            v.kind = Value::Integer;
            v.i = std::stoll(clean, &used, 0);
            v.f = (double)v.i;
        } else {
            v.kind = Value::Float;
            v.f = std::stod(clean, &used);
        }
        if (used != clean.size()) throw std::runtime_error("toml: bad value '" + tok + "'");
        return v;
    }

public:
    static Document parse_string(const std::string& text)
    {
        Document d;
        std::string table;
        size_t p = 0;
        while (p < text.size()) {
            skip_ws(text, p, true);
            if (p >= text.size()) break;
            if (text[p] == '[') {
                const size_t e = text.find(']', p);
                if (e == std::string::npos) throw std::runtime_error("toml: unterminated table header");
                table = text.substr(p + 1, e - p - 1);
                while (!table.empty() && std::isspace((unsigned char)table.back())) table.pop_back();
                while (!table.empty() && std::isspace((unsigned char)table.front())) table.erase(0, 1);
                p = e + 1;
                continue;
            }
            size_t e = p;
            while (e < text.size() && text[e] != '=' && text[e] != '\n') ++e;
            if (e >= text.size() || text[e] != '=') throw std::runtime_error("toml: '=' expected near '" + text.substr(p, 20) + "'");
            std::string key = text.substr(p, e - p);
            while (!key.empty() && std::isspace((unsigned char)key.back())) key.pop_back();
            if (key.size() >= 2 && (key.front() == '"' || key.front() == '\'')) key = key.substr(1, key.size() - 2);
            p = e + 1;
            skip_ws(text, p, false);
            d.kv_[table + "." + key] = parse_value(text, p);
        }
        return d;
    }

    static Document parse_file(const std::string& path)
    {
        std::ifstream in(path, std::ios_base::binary);
        if (!in.is_open()) throw std::runtime_error("toml: cannot open " + path);
        std::stringstream ss;
        ss << in.rdbuf();
        return parse_string(ss.str());
    }

    bool has(const std::string& table, const std::string& key) const { return kv_.count(table + "." + key) != 0; }
    const Value& at(const std::string& table, const std::string& key) const
    {
        auto it = kv_.find(table + "." + key);
        if (it == kv_.end()) throw std::out_of_range("toml: key '" + key + "' not found in [" + table + "]");
        return it->second;
    }
    std::string get_string(const std::string& table, const std::string& key) const
    {
        const Value& v = at(table, key);
        if (v.kind != Value::String) throw std::runtime_error("toml: '" + key + "' is not a string");
        return v.s;
    }
    long long get_int_or(const std::string& table, const std::string& key, long long dflt) const
    {
        if (!has(table, key)) return dflt;
        const Value& v = at(table, key);
        if (v.kind != Value::Integer) throw std::runtime_error("toml: '" + key + "' is not an integer");
        return v.i;
    }
    double get_double_or(const std::string& table, const std::string& key, double dflt) const
    {
        if (!has(table, key)) return dflt;
        const Value& v = at(table, key);
        if (v.kind != Value::Float && v.kind != Value::Integer) throw std::runtime_error("toml: '" + key + "' is not a number");
        return v.f;
    }
    std::string get_string_or(const std::string& table, const std::string& key, const std::string& dflt) const
    {
        return has(table, key) ? get_string(table, key) : dflt;
    }
    std::vector<std::string> get_string_array(const std::string& table, const std::string& key) const
    {
        const Value& v = at(table, key);
        if (v.kind != Value::Array) throw std::runtime_error("toml: '" + key + "' is not an array");
        std::vector<std::string> out;
        for (const Value& e : v.arr) {
            if (e.kind != Value::String) throw std::runtime_error("toml: '" + key + "' holds a non-string");
            out.push_back(e.s);
        }
        return out;
    }
    std::vector<long long> get_int_array(const std::string& table, const std::string& key) const
    {
        const Value& v = at(table, key);
        if (v.kind != Value::Array) throw std::runtime_error("toml: '" + key + "' is not an array");
        std::vector<long long> out;
        for (const Value& e : v.arr) out.push_back(e.i);
        return out;
    }
};

}  // namespace toml_lite
