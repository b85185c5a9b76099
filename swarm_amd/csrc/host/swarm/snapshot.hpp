// swarm/snapshot.hpp -- ensemble file I/O, mirror of swarm/snapshot.{hpp,cpp}.
#pragma once
#include <string>

#include "ensemble.hpp"

namespace swarm { namespace snapshot {

struct header { int nsys, nbod, nsysattr, nbodattr; };                       // snapshot.hpp:153-155
struct sys { double time; int id; int state; double attribute[ensemble::NUM_SYS_ATTRIBUTES]; };
struct body { double pos[3], vel[3], mass, attribute[ensemble::NUM_BODY_ATTRIBUTES]; };

struct readfileexception : public std::exception {
    std::string filename, message, text;
    readfileexception(const std::string &fn, const std::string &m) : filename(fn), message(m), text("Error reading " + fn + " : " + m) {}
    virtual ~readfileexception() throw() {}
    virtual const char *what() const throw() { return text.c_str(); }
};
struct writefileexception : public std::exception {
    std::string filename, message, text;
    writefileexception(const std::string &fn, const std::string &m) : filename(fn), message(m), text("Error writing " + fn + " : " + m) {}
    virtual ~writefileexception() throw() {}
    virtual const char *what() const throw() { return text.c_str(); }
};

defaultEnsemble load(const std::string &filename);
defaultEnsemble load_text(const std::string &filename);
void save(defaultEnsemble &ens, const std::string &filename);
void save_text(defaultEnsemble &ens, const std::string &filename);

} } // namespace swarm::snapshot
