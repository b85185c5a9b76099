// swarm/query.hpp -- binary log files and the `swarm query` text dump.
// Mirrors swarm/log/fileformat.hpp (swarm_header), swarm/log/io.{hpp,cpp} (sorting, ranges) and
// swarm/query.{hpp,cpp} (record formatting: Cartesian and Keplerian (-k) output of EVT_SNAPSHOT records,
// transit/occultation events).  The .db (Berkeley DB) files are not provided.
#pragma once
#include <cstdint>
#include <cstring>
#include <iostream>
#include <limits>
#include <string>

#include "common.hpp"

namespace swarm {

// log/fileformat.hpp:28-68: 96-byte header, padded to 16
struct alignas(16) swarm_header {
    char magic[6];   // "SWARM\0"
    char version[2]; // "0\0"
    char m_type[76]; // content id // description
    uint32_t flags;
    uint64_t datalen; // 0xFFFFFFFFFFFFFFFF = unknown
    static const uint64_t npos = 0xFFFFFFFFFFFFFFFFULL;
    swarm_header(const std::string &type, int flags_ = 0, uint64_t datalen_ = npos) {
        std::memset(this, 0, sizeof(*this));
        std::strcpy(magic, "SWARM");
        std::strcpy(version, "0");
        std::strncpy(m_type, type.c_str(), sizeof(m_type));
        m_type[sizeof(m_type) - 1] = '\0';
        flags = flags_;
        datalen = datalen_;
    }
    std::string type() const; // text before "//", trimmed
};
static_assert(sizeof(swarm_header) == 96, "swarm_header must be 96 bytes");

namespace query {

extern const char *UNSORTED_HEADER_FULL, *UNSORTED_HEADER_CHECK, *SORTED_HEADER_FULL, *SORTED_HEADER_CHECK;

// log/io.hpp:83-99
template <typename T> struct range {
    T first, last;
    range(const T &a) : first(a), last(a) {}
    range(const T &a, const T &b) : first(a), last(b) {}
    range()
        : first(std::numeric_limits<T>::is_integer ? std::numeric_limits<T>::min() : -std::numeric_limits<T>::max()),
          last(std::numeric_limits<T>::max()) {}
    bool in(const T &v) const { return first <= v && v <= last; }
    // query.hpp:52-92: ALL | r | r1..r2 | MIN..r2 | r1..MAX
    static range parse(const std::string &s);
};
typedef range<int> sys_range_t;
typedef range<int> body_range_t;
typedef range<double> time_range_t;

enum planets_coordinate_system_t { astrocentric, barycentric, jacobi, origin };
void set_coordinate_system(const planets_coordinate_system_t &c);
void set_keplerian_output(); // query.cpp:67-69, `swarm query -k`
// Cartesian state relative to the centre -> (a, e, i, O, w, M), angles in radians; the conventions and
// the case split (parabola / ellipse / hyperbola, TINY = 1e-8) of tutorials/kepler.cpp:102-200
void calc_keplerian_for_cartesian(double &a, double &e, double &i, double &O, double &w, double &M, double x, double y,
                                  double z, double vx, double vy, double vz, double GM);

// io.cpp:226-264: read <infn> (unsorted), sort records by (T, sys, position), write <outfn>
bool sort_binary_log_file(const std::string &outfn, const std::string &infn);
// query.cpp:318-341: print the records of a binary log file (sorted or raw) in (T, sys) order
void execute(const std::string &datafile, time_range_t T, sys_range_t sys, body_range_t bod, std::ostream &out = std::cout);

} // namespace query
} // namespace swarm
