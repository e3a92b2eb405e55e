// micv_config.hpp -- header-only reader for the reference's run configurations (config/psN.yaml),
// so a psN-style main() can be rebuilt without yaml-cpp (its submodule is empty in the reference
// tree).  Same access pattern as the reference's Config classes use on YAML::Node
// (ps4_cpp/lib/Config.cpp:25-133: config["harris_trans"], node["window_size"].as<size_t>() ...).
// Accepts the YAML subset those files use: `---` / `...`, comments, `key: scalar`, one level of
// nested maps by indentation.  The Python twin is introtocomputervision_amd/config.py.
#pragma once
#include <cctype>
#include <cstdlib>
#include <fstream>
#include <map>
#include <memory>
#include <sstream>
#include <stdexcept>
#include <string>

namespace micv_config {

class Node {
public:
    bool has(const std::string &key) const { return scalars_.count(key) || maps_.count(key); }
    const Node &child(const std::string &key) const {
        auto it = maps_.find(key);
        if (it == maps_.end()) throw std::runtime_error("config: '" + key + "' is not a map");
        return *it->second;
    }
    const std::string &str(const std::string &key) const {
        auto it = scalars_.find(key);
        if (it == scalars_.end()) throw std::runtime_error("config: key '" + key + "' not found");
        return it->second;
    }
    template <typename T>
    T as(const std::string &key) const;

    static Node load(const std::string &path) {
        std::ifstream f(path);
        if (!f) throw std::runtime_error("config: cannot open " + path);
        std::stringstream ss;
        ss << f.rdbuf();
        return parse(ss.str());
    }
    static Node parse(const std::string &text) {
        Node root;
        Node *current = nullptr;
        int current_indent = -1, lineno = 0;
        std::istringstream in(text);
        std::string raw;
        while (std::getline(in, raw)) {
            lineno++;
            std::string line = strip_comment(raw);
            const size_t first = line.find_first_not_of(" ");
            if (first == std::string::npos) continue;
            const std::string body = trim(line);
            if (body == "---" || body == "...") continue;
            const size_t colon = body.find(':');
            if (colon == std::string::npos || (colon + 1 < body.size() && body[colon + 1] != ' ' && body[colon + 1] != '\t'))
                throw std::runtime_error("config: line " + std::to_string(lineno) + ": expected 'key: value'");
            const std::string key = unquote(trim(body.substr(0, colon)));
            const std::string value = trim(body.substr(colon + 1));
            if (first == 0) {
                if (value.empty()) {
                    root.maps_[key] = std::make_shared<Node>();
                    current = root.maps_[key].get();
                    current_indent = -1;
                } else {
                    root.scalars_[key] = unquote(value);
                    current = nullptr;
                }
            } else {
                if (!current) throw std::runtime_error("config: line " + std::to_string(lineno) + ": indented entry without a parent");
                if (current_indent < 0) current_indent = (int)first;
                if ((int)first != current_indent || value.empty())
                    throw std::runtime_error("config: line " + std::to_string(lineno) + ": only one level of nesting is supported");
                current->scalars_[key] = unquote(value);
            }
        }
        return root;
    }

private:
    std::map<std::string, std::string> scalars_;
    std::map<std::string, std::shared_ptr<Node>> maps_;

    static std::string trim(const std::string &s) {
        const size_t a = s.find_first_not_of(" \t\r"), b = s.find_last_not_of(" \t\r");
        return a == std::string::npos ? std::string() : s.substr(a, b - a + 1);
    }
    static std::string unquote(const std::string &s) {
        if (s.size() >= 2 && s.front() == s.back() && (s.front() == '"' || s.front() == '\'')) return s.substr(1, s.size() - 2);
        return s;
    }
    static std::string strip_comment(const std::string &line) {
        char quote = 0;
        for (size_t i = 0; i < line.size(); i++) {
            const char ch = line[i];
            if (quote) {
                if (ch == quote) quote = 0;
            } else if (ch == '"' || ch == '\'') {
                quote = ch;
            } else if (ch == '#' && (i == 0 || line[i - 1] == ' ' || line[i - 1] == '\t')) {
                return line.substr(0, i);
            }
        }
        return line;
    }
};

template <>
inline std::string Node::as<std::string>(const std::string &key) const { return str(key); }
template <>
inline double Node::as<double>(const std::string &key) const {
    const std::string &v = str(key);
    char *end = nullptr;
    const double d = std::strtod(v.c_str(), &end);
    if (end == v.c_str() || *end) throw std::runtime_error("config: '" + key + ": " + v + "' is not a number");
    return d;
}
template <>
inline float Node::as<float>(const std::string &key) const { return static_cast<float>(as<double>(key)); }
template <>
inline long long Node::as<long long>(const std::string &key) const {
    const std::string &v = str(key);
    char *end = nullptr;
    const long long d = std::strtoll(v.c_str(), &end, 0);
    if (end == v.c_str() || *end) throw std::runtime_error("config: '" + key + ": " + v + "' is not an integer");
    return d;
}
template <>
inline int Node::as<int>(const std::string &key) const { return static_cast<int>(as<long long>(key)); }
template <>
inline size_t Node::as<size_t>(const std::string &key) const { return static_cast<size_t>(as<long long>(key)); }
template <>
inline unsigned Node::as<unsigned>(const std::string &key) const { return static_cast<unsigned>(as<long long>(key)); }
template <>
inline bool Node::as<bool>(const std::string &key) const {
    std::string v = str(key);
    for (auto &c : v) c = (char)std::tolower((unsigned char)c);
    if (v == "true" || v == "yes" || v == "on" || v == "y") return true;
    if (v == "false" || v == "no" || v == "off" || v == "n") return false;
    throw std::runtime_error("config: '" + key + ": " + v + "' is not a boolean");
}

// Config::Harris of ps4 (ps4_cpp/include/Config.h, lib/Config.cpp:43-54).
struct Harris {
    int sobel_kernel_size = 3;
    size_t window_size = 5;
    double gaussian_sigma = 1.5;
    float alpha = 0.04f;
    double response_threshold = 5e8;
    int min_distance = 5;
    explicit Harris(const Node &n)
        : sobel_kernel_size(n.as<int>("sobel_kernel_size")), window_size(n.as<size_t>("window_size")),
          gaussian_sigma(n.as<double>("gaussian_sigma")), alpha(n.as<float>("alpha")),
          response_threshold(n.as<double>("response_threshold")), min_distance(n.as<int>("min_distance")) {}
};

}  // namespace micv_config
