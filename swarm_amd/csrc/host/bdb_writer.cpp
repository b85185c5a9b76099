// bdb_writer.cpp -- the `bdb` log writer: the event log as a Berkeley DB file that the reference's readers open
// (`swarm query -d`, py/swarmng/logdb.py).  Counterpart of swarm/log/bdb_writer.cpp:55-94 and
// swarm/log/bdb_database.{hpp,cpp}; SURVEY 8(f) N2.
//
// Compiled only where <db_cxx.h> exists -- the Makefile probes for it, as the reference's build adds its BDB plugin only
// `if(BDB_FOUND)` (src/plugins.cmake:36-39).  THIS IMAGE HAS NO libdb: the file has never been linked against a real
// Berkeley DB.  What checks it here is tests/fake_bdb/db_cxx.h, an in-memory test double of the dozen calls used below
// (tests/test_bdb_writer.py builds this file against it and inspects what was stored).
//
// The file format is the contract, so it is the reference's to the letter (bdb_database.cpp:124-165, 182-203, 225-232):
//   one file, five databases:
//     "metadata"    BTREE  name -> value; "fileFormatVersion" = "1", "swarmngVersion" = "1.1"
//     "primary"     BTREE  key {float time; uint32 event_id << 24 | system_id} (8 bytes, ordered by time, then by the
//                          integer), value = the gpulog record as the kernel wrote it; a second record with the same
//                          key REPLACES the first (Db::put without DB_NOOVERWRITE)
//     "system_idx", "time_idx", "event_idx"
//                   BTREE secondaries with sorted duplicates, keyed by the system id (uint32), the time (float) and the
//                   event id (uint8) taken from the PRIMARY KEY, duplicates ordered like the primary
//   records of the kinds 1, 2, 11, 15, 16 carry (double time, int system) behind their header; every other record
//   (lprintf lines) is filed under time = NaN, system 0 -- and sorted BEFORE all numbers, see `before` below.
//   The environment lives in the directory of the file: DB_CREATE | DB_INIT_CDB | DB_INIT_MPOOL, 64 MiB of cache.
#include <db_cxx.h>
#include <libgen.h>

#include <cstdlib>
#include <cstring>
#include <limits>
#include <string>

#include "swarm/swarm.h"

namespace swarm { namespace log {

namespace {

struct record_key {
    float time;
    uint32_t event_and_system;
    record_key() : time(0.f), event_and_system(0) {}
    record_key(float t, int system, int event) : time(t), event_and_system((uint32_t)event << 24 | (uint32_t)system) {}
    uint32_t system() const { return event_and_system & 0x00ffffffu; }
    uint8_t event() const { return (uint8_t)(event_and_system >> 24); }
};
// One deliberate difference from bdb_database.cpp:20-25: there a NaN time is neither before nor after any other, so
// the comparison calls an lprintf record's key EQUAL to whatever key the descent meets first, and Db::put REPLACES that
// record with the printed line.  Here NaN times come before every number (and are equal among themselves), which makes
// the order total; for keys with real times it is the reference's order.
inline bool before(float a, float b) { return (a != a) ? (b == b) : a < b; }
inline bool before(const record_key &a, const record_key &b) {
    const bool same_time = a.time == b.time || (a.time != a.time && b.time != b.time);
    return same_time ? a.event_and_system < b.event_and_system : before(a.time, b.time);
}
inline bool before(uint32_t a, uint32_t b) { return a < b; }
inline bool before(uint8_t a, uint8_t b) { return a < b; }

// B-tree order of fixed-size keys: shorter keys first, equal-sized keys of the expected type by `before`, anything
// else equal (bdb_database.cpp:77-93).  Berkeley DB 6 added a fourth argument to the comparison callbacks.
template <class T>
#if DB_VERSION_MAJOR >= 6
int order_of(DB *, const DBT *a, const DBT *b, size_t *) {
#else
int order_of(DB *, const DBT *a, const DBT *b) {
#endif
    if (a->size != b->size) return a->size < b->size ? -1 : 1;
    if (a->size != sizeof(T)) return 0;
    T x, y;
    std::memcpy(&x, a->data, sizeof(T));
    std::memcpy(&y, b->data, sizeof(T));
    return before(x, y) ? -1 : (before(y, x) ? 1 : 0);
}

// secondary keys are cut from the primary key; the library frees what the callback hands it (DB_DBT_APPMALLOC)
template <class T>
int hand_over(const T &v, Dbt *result) {
    void *p = std::malloc(sizeof(T));
    if (!p) return 1;
    std::memcpy(p, &v, sizeof(T));
    result->set_flags(DB_DBT_APPMALLOC);
    result->set_data(p);
    result->set_size(sizeof(T));
    return 0;
}
inline record_key key_of(const Dbt *k) {
    record_key rk;
    std::memcpy(&rk, k->get_data(), sizeof(rk));
    return rk;
}
int system_of(Db *, const Dbt *k, const Dbt *, Dbt *out) { return hand_over<uint32_t>(key_of(k).system(), out); }
int time_of(Db *, const Dbt *k, const Dbt *, Dbt *out) { return hand_over<float>(key_of(k).time, out); }
int event_of(Db *, const Dbt *k, const Dbt *, Dbt *out) { return hand_over<uint8_t>(key_of(k).event(), out); }

std::string dir_of(const std::string &path) {
    std::string w(path);
    return std::string(dirname(&w[0]));
}
std::string file_of(const std::string &path) {
    std::string w(path);
    return std::string(basename(&w[0]));
}

const int FILE_MODE = 00600;
const unsigned CACHE_BYTES = 64u << 20;

} // namespace

class bdb_writer : public writer {
    DbEnv env;
    Db metadata, primary, by_system, by_time, by_event;
    bool open;

    template <class T>
    void open_index(Db &db, const char *file, const char *name) {
        db.set_flags(DB_DUP | DB_DUPSORT);
        db.set_bt_compare(order_of<T>);
        db.set_dup_compare(order_of<record_key>);
        db.open(NULL, file, name, DB_BTREE, DB_CREATE, FILE_MODE);
    }
    void put_text(const char *name, const char *value) {
        Dbt k((void *)name, (u_int32_t)std::strlen(name)), v((void *)value, (u_int32_t)std::strlen(value));
        metadata.put(NULL, &k, &v, 0);
    }

public:
    bdb_writer(const config &cfg)
        : env(0u), metadata(&env, 0), primary(&env, 0), by_system(&env, 0), by_time(&env, 0), by_event(&env, 0), open(false) {
        const std::string path = cfg.require("log_output_db", std::string());
        const std::string dir = dir_of(path), file = file_of(path);
        env.set_cachesize(0, CACHE_BYTES, 0);
        std::cerr << "Initializing BDB environment in `" << dir << "`" << std::endl;
        env.open(dir.c_str(), DB_CREATE | DB_INIT_CDB | DB_INIT_MPOOL, FILE_MODE);
        const char *fn = file.c_str();
        metadata.open(NULL, fn, "metadata", DB_BTREE, DB_CREATE, FILE_MODE);
        primary.set_bt_compare(order_of<record_key>);
        primary.open(NULL, fn, "primary", DB_BTREE, DB_CREATE, FILE_MODE);
        open_index<uint32_t>(by_system, fn, "system_idx");
        open_index<float>(by_time, fn, "time_idx");
        open_index<uint8_t>(by_event, fn, "event_idx");
        primary.associate(NULL, &by_system, system_of, DB_IMMUTABLE_KEY);
        primary.associate(NULL, &by_time, time_of, DB_IMMUTABLE_KEY);
        primary.associate(NULL, &by_event, event_of, DB_IMMUTABLE_KEY);
        put_text("fileFormatVersion", "1");
        put_text("swarmngVersion", "1.1");
        open = true;
    }

    // every record of the flushed bytes goes into the primary database (the secondaries follow by association)
    virtual void process(const char *p, size_t n) {
        size_t off = 0;
        while (off + 8 <= n) {
            int32_t hdr[2]; // gpulog header: msgid, len
            std::memcpy(hdr, p + off, 8);
            if (hdr[1] < 8 || off + (size_t)hdr[1] > n) break;
            double time = std::numeric_limits<double>::quiet_NaN();
            int32_t system = 0;
            const int id = hdr[0];
            if ((id == 1 || id == 2 || id == 11 || id == 15 || id == 16) && hdr[1] >= 20) {
                std::memcpy(&time, p + off + 8, 8);
                std::memcpy(&system, p + off + 16, 4);
            }
            record_key rk((float)time, system, id);
            Dbt k(&rk, sizeof(rk)), v((void *)(p + off), (u_int32_t)hdr[1]);
            primary.put(NULL, &k, &v, 0);
            off += (size_t)hdr[1];
        }
        metadata.sync(0);
        primary.sync(0);
        by_time.sync(0);
        by_event.sync(0);
        by_system.sync(0);
    }

    virtual ~bdb_writer() {
        if (!open) return;
        try {
            by_event.close(0);
            by_time.close(0);
            by_system.close(0);
            primary.close(0);
            metadata.close(0);
            env.close(0);
        } catch (...) {
        }
    }
};

static writer_plugin_initializer<bdb_writer> bdb_writer_plugin("bdb", "This is the Berkeley DB writer");

} } // namespace swarm::log
