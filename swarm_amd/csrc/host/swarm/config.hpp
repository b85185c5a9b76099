// swarm/config.hpp -- swarm::config, mirror of swarm/types/config.{hpp,cpp}.
#pragma once
#include <cstdio>
#include <fstream>
#include <map>
#include <string>

#include "common.hpp"

namespace swarm {

// types/config.hpp:31-33
struct key_not_found : public std::runtime_error {
    key_not_found(const std::string &key) : std::runtime_error("Key not found: \"" + key + "\"") {}
};

inline std::string trim(const std::string &s) {
    size_t a = s.find_first_not_of(" \t\r\n"), b = s.find_last_not_of(" \t\r\n");
    return a == std::string::npos ? std::string() : s.substr(a, b - a + 1);
}

class config : public std::map<std::string, std::string> {
public:
    config() {}
    // config::load, types/config.cpp:41-65: `#` comments, blank lines, key=value, last one wins
    static config load(const std::string &fn, config cfg = config()) {
        std::ifstream in(fn.c_str());
        if (!in) ERROR("Cannot open configuration file '" + fn + "'.");
        std::string line;
        while (std::getline(in, line)) {
            line = trim(line);
            if (line.empty() || line[0] == '#') continue;
            size_t eqpos = line.find('=');
            if (eqpos == std::string::npos) ERROR("Error on line " + line + ": '=' sign expected.");
            cfg[trim(line.substr(0, eqpos))] = trim(line.substr(eqpos + 1));
        }
        return cfg;
    }
    // typed parsing with the reference's sscanf semantics (types/config.hpp:98-135); bool goes
    // through the int overload, so "=true" is not a valid value
    static bool valid_value(const std::string &v, std::string) { return v.size() > 0; }
    static std::string parse(const std::string &v, std::string) { return v; }
    static bool valid_value(const std::string &v, double) { double d = 0; return sscanf(v.c_str(), "%lg", &d) == 1; }
    static double parse(const std::string &v, double) { double d = 0; sscanf(v.c_str(), "%lg", &d); return d; }
    static bool valid_value(const std::string &v, float) { float d = 0; return sscanf(v.c_str(), "%g", &d) == 1; }
    static float parse(const std::string &v, float) { float d = 0; sscanf(v.c_str(), "%g", &d); return d; }
    static bool valid_value(const std::string &v, int) { int d = 0; return sscanf(v.c_str(), "%i", &d) == 1; }
    static int parse(const std::string &v, int) { int d = 0; sscanf(v.c_str(), "%i", &d); return d; }

    bool valid(const std::string &key) const { return count(key) && (at(key).size() > 0); }
    template <typename T> bool valid(const std::string &key, T t) const { return valid(key) && valid_value(at(key), t); }
    template <typename T> T optional(const std::string &key, const T &default_value) const {
        if (count(key)) {
            std::string value = at(key);
            return valid_value(value, T()) ? (T)parse(value, T()) : default_value;
        }
        return default_value;
    }
    template <typename T> T require(const std::string &key, const T & = T()) const {
        if (count(key) && valid_value(at(key), T())) return (T)parse(at(key), T());
        throw key_not_found(key);
    }
};

} // namespace swarm
