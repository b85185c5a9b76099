// swarm/log.hpp -- host side of the event log: writer plugins and the log manager.
// Mirrors swarm/log/writer.{h,cpp}, null_writer.cpp, host_array_writer.cpp (simplified),
// binary_writer.cpp (raw record stream) and swarm/log/logmanager.{hpp,cpp}.
#pragma once
#include <cstdint>
#include <cstdio>
#include <iostream>
#include <memory>
#include <string>
#include <vector>

#include "plugin.hpp"
#include "swarm_hip.h"

namespace swarm { namespace log {

// log/writer.h: a sink that is handed the raw bytes of the flushed log
class writer {
public:
    virtual ~writer() {}
    virtual void process(const char *log_data, size_t length) = 0;
    // writer::create, log/writer.cpp:44-58: cfg `log_writer` selects plugin "writer_<name>"
    static writer *create(const config &cfg);
};
typedef std::shared_ptr<writer> Pwriter;

class null_writer : public writer {
public:
    null_writer(const config &) {}
    virtual void process(const char *, size_t) {}
};

// keeps every flushed byte in memory; get via manager::get_writer()
class host_array_writer : public writer {
public:
    std::vector<uint8_t> bytes;
    host_array_writer(const config &) {}
    virtual void process(const char *p, size_t n) { bytes.insert(bytes.end(), (const uint8_t *)p, (const uint8_t *)p + n); }
};

// log/binary_writer.cpp:39-91: swarm_header("unsorted_output") + the raw gpulog record stream go
// to <log_output>.raw; on close the file is sorted by (time, system) into <log_output> and the
// .raw file is removed.  (The reference also pre-builds .time.idx/.sys.idx index files.)
class binary_writer : public writer {
    FILE *f;
    std::string rawfn, binfn;
public:
    binary_writer(const config &cfg);
    virtual ~binary_writer();
    virtual void process(const char *p, size_t n);
};

// lprintf host side, swarm/log/gpulog/lprintf.h:74-195
std::string run_printf(const char *record);
int replay_printf(std::ostream &out, const char *buffer, size_t bytes);

// log::manager, logmanager.hpp / logmanager.cpp:37-87
class manager {
    Pwriter log_writer;
    uint8_t *d_buf;
    int32_t *d_cursor;
    int capacity;
    uint8_t *host; // pinned, `capacity` bytes
    long long _dropped;
public:
    manager() : d_buf(0), d_cursor(0), capacity(0), host(0), _dropped(0) {}
    ~manager();
    // default 50 MiB each, logmanager.hpp:67
    void init(const config &cfg, int host_buffer_size = 50 * 1024 * 1024, int device_buffer_size = 50 * 1024 * 1024);
    // the default log: null writer, no buffers (logmanager.cpp:37-47)
    static std::shared_ptr<manager> &default_log();
    swarm_hip_log_t get_gpulog() const { swarm_hip_log_t l = {d_buf, d_cursor, capacity}; return l; }
    writer *get_writer() const { return log_writer.get(); }
    long long dropped() const { return _dropped; }
    // flush: D2H copy of the used bytes, clear the device cursor, writer.process (logmanager.cpp:67-87)
    void flush();
};
typedef std::shared_ptr<manager> Pmanager;

} } // namespace swarm::log
