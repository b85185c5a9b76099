// swarm/plugin.hpp -- plugin registry, mirror of swarm/plugin.{hpp,cpp}.
#pragma once
#include <iostream>
#include <map>
#include <string>
#include <vector>

#include "config.hpp"

namespace swarm {

// plugin.hpp:187-196
struct plugin_not_found : std::exception {
    std::string _name;
    plugin_not_found(std::string name) : _name("Plugin " + name + " was not found ") {}
    virtual ~plugin_not_found() throw() {}
    virtual const char *what() const throw() { return _name.c_str(); }
};

class plugin {
public:
    virtual ~plugin() {}
    virtual void *create(const config &cfg) = 0;
    virtual std::string description() { return ""; }
    virtual std::string id() = 0;

    static void add(plugin *p);
    static plugin *find(const std::string &id);
    static void *instance(const std::string &name, const config &cfg);
    static std::vector<plugin *> all();
    struct help_t {};
    static help_t help;
};
std::ostream &operator<<(std::ostream &, const plugin::help_t &);

template <class T> struct basic_plugin : public plugin {
    std::string _id, _description;
    basic_plugin(const std::string &i, const std::string &d = std::string()) : _id(i), _description(d) {}
    virtual void *create(const config &cfg) { return new T(cfg); }
    virtual std::string id() { return _id; }
    virtual std::string description() { return _description; }
};

// plugin.hpp:157-164: id = "integrator_" + name
template <class T> struct integrator_plugin_initializer {
    integrator_plugin_initializer(const std::string &id, const std::string &description = std::string()) {
        plugin::add(new basic_plugin<T>("integrator_" + id, description));
    }
};
template <class T> struct writer_plugin_initializer {
    writer_plugin_initializer(const std::string &id, const std::string &description = std::string()) {
        plugin::add(new basic_plugin<T>("writer_" + id, description));
    }
};

} // namespace swarm
