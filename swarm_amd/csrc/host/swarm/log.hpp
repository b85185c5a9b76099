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
// `log_buffers=2` (no reference key; default 1 = the reference's protocol, one synchronous copy after every launch):
// two device logs and two page-locked host areas; flush_async() starts the copy of the log the launches so far have
// filled on a second stream and hands the next launch the other log, so that the copy runs while that launch computes.
// The writer receives the same records (their order in the stream is free in the reference too).
class manager {
    Pwriter log_writer;
    int nbuf, cur;
    uint8_t *d_buf[2];
    int32_t *d_cursor[2];
    int capacity;
    uint8_t *host[2]; // pinned, `capacity` bytes each
    long long _dropped;
    // asynchronous flush
    void *copy_stream, *ev_launch, *ev_copy[2];
    int32_t *cursor_image; // pinned int32[4]
    struct pending_t {
        bool in_flight;
        int32_t used, dropped, nprintf;
    } pending[2];
    bool dirty[2]; // the log's device cursor still holds the count of a copy that was started
    void process(int i, int32_t used, int32_t dropped, int32_t nprintf);
    void finish(int i);

public:
    manager();
    ~manager();
    // default 50 MiB each, logmanager.hpp:67
    void init(const config &cfg, int host_buffer_size = 50 * 1024 * 1024, int device_buffer_size = 50 * 1024 * 1024);
    // the default log: null writer, no buffers (logmanager.cpp:37-47)
    static std::shared_ptr<manager> &default_log();
    // the log the next launch appends to
    swarm_hip_log_t get_gpulog() const { swarm_hip_log_t l = {d_buf[cur], d_cursor[cur], capacity}; return l; }
    writer *get_writer() const { return log_writer.get(); }
    long long dropped() const { return _dropped; }
    bool overlapping() const { return nbuf == 2 && d_buf[0]; }
    // flush: D2H copy of the used bytes, clear the device cursor, writer.process (logmanager.cpp:67-87); also
    // completes the copies flush_async started.  The launch stream must have drained.
    void flush();
    // start flushing what the launches enqueued on `launch_stream` so far have logged and switch logs; returns when
    // those launches have finished and the log that is current from now on has been handed to the writer
    void flush_async(void *launch_stream);
};
typedef std::shared_ptr<manager> Pmanager;

} } // namespace swarm::log
