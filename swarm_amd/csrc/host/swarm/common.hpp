// swarm/common.hpp -- error type and output macros of the host mirror.
// Mirrors swarm/runtime_error.hpp:38-51 and swarm/utils.hpp:49-51 of the reference.
#pragma once
#include <cstdio>
#include <iostream>
#include <stdexcept>
#include <string>

extern int DEBUG_LEVEL;

namespace swarm {

class runtime_error : public std::runtime_error {
public:
    runtime_error(const std::string &msg) : std::runtime_error(msg) {}
};

#define ERROR(msg) throw swarm::runtime_error(msg);
#define DEBUG_OUTPUT(level, message) \
    ((DEBUG_LEVEL >= level) ? (std::cerr << __FUNCTION__ << ":" << __LINE__ << ": " << message << std::endl) : std::cerr)
#define INFO_OUTPUT(level, message) ((DEBUG_LEVEL >= level) ? (std::cerr << message) : std::cerr)

// A failed libswarm_hip.so call: the role cudaException plays in the reference
// (swarm/runtime_error.hpp:58-74).
struct hipException : public runtime_error {
    hipException(const std::string &m) : runtime_error(m) {}
    static void check(int rc);
};
#define hipErrCheck(expr) swarm::hipException::check(expr)

} // namespace swarm
