// tests/fake_bdb/db_cxx.h -- an in-memory TEST DOUBLE of the part of Berkeley DB's C++ API that
// swarm_amd/csrc/host/bdb_writer.cpp uses (this image has no libdb).  Not Berkeley DB, not a file format: every
// database keeps its (key, value) pairs in B-tree order by the comparison callbacks it was given and, when it is closed,
// writes them as hex lines to <environment dir>/<file>.<database>.dump, which tests/test_bdb_writer.py reads.  Signatures
// follow the documented Berkeley DB 5 C++ interface (Db, DbEnv, Dbt; DB / DBT in the comparison callbacks).
#pragma once
#include <sys/types.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#define DB_VERSION_MAJOR 5
#define DB_CREATE 0x1u
#define DB_INIT_CDB 0x80u
#define DB_INIT_MPOOL 0x400u
#define DB_DUP 0x10u
#define DB_DUPSORT 0x2u
#define DB_IMMUTABLE_KEY 0x2u
#define DB_DBT_APPMALLOC 0x1u
enum DBTYPE { DB_BTREE = 1 };

struct DBT {
    void *data;
    u_int32_t size, flags;
};
struct DB {};
class DbTxn;

class Dbt : private DBT {
public:
    Dbt() { data = 0, size = 0, flags = 0; }
    Dbt(void *d, u_int32_t n) { data = d, size = n, flags = 0; }
    void *get_data() const { return data; }
    u_int32_t get_size() const { return size; }
    u_int32_t get_flags() const { return flags; }
    void set_data(void *d) { data = d; }
    void set_size(u_int32_t n) { size = n; }
    void set_flags(u_int32_t f) { flags = f; }
    const DBT *raw() const { return this; }
};

class DbEnv {
public:
    std::string home;
    bool opened;
    explicit DbEnv(u_int32_t) : opened(false) {}
    int set_cachesize(u_int32_t, u_int32_t, int) { return 0; }
    int open(const char *dir, u_int32_t flags, int) {
        if (!(flags & DB_CREATE)) std::abort();
        home = dir, opened = true;
        return 0;
    }
    int close(u_int32_t) { opened = false; return 0; }
};

class Db {
    typedef int (*compare_t)(DB *, const DBT *, const DBT *);
    typedef int (*extract_t)(Db *, const Dbt *, const Dbt *, Dbt *);
    struct pair_t {
        std::string k, v;
    };
    DbEnv *env;
    compare_t key_order, dup_order;
    u_int32_t flags;
    std::string path;
    std::vector<pair_t> rows;
    std::vector<std::pair<Db *, extract_t> > secondaries;
    bool opened;
    DB handle;

    int cmp(compare_t f, const std::string &a, const std::string &b) {
        DBT x = {(void *)a.data(), (u_int32_t)a.size(), 0}, y = {(void *)b.data(), (u_int32_t)b.size(), 0};
        if (f) return f(&handle, &x, &y);
        int c = std::memcmp(a.data(), b.data(), a.size() < b.size() ? a.size() : b.size()); // default: lexical
        return c ? c : (a.size() < b.size() ? -1 : (a.size() > b.size() ? 1 : 0));
    }
    void insert(const std::string &k, const std::string &v) {
        size_t i = 0;
        for (; i < rows.size(); i++) {
            int c = cmp(key_order, k, rows[i].k);
            if (c == 0 && (flags & DB_DUP)) c = cmp(dup_order, v, rows[i].v);
            if (c == 0) {
                rows[i].v = v; // same key (and, with sorted duplicates, same datum): replaced
                return;
            }
            if (c < 0) break;
        }
        pair_t p = {k, v};
        rows.insert(rows.begin() + i, p);
    }

public:
    Db(DbEnv *e, u_int32_t) : env(e), key_order(0), dup_order(0), flags(0), opened(false) {}
    int set_flags(u_int32_t f) { flags |= f; return 0; }
    int set_bt_compare(compare_t f) { key_order = f; return 0; }
    int set_dup_compare(compare_t f) { dup_order = f; return 0; }
    int open(DbTxn *, const char *file, const char *database, DBTYPE, u_int32_t oflags, int) {
        if (!env || !env->opened || !(oflags & DB_CREATE)) std::abort();
        path = env->home + "/" + file + "." + database + ".dump";
        opened = true;
        return 0;
    }
    int associate(DbTxn *, Db *secondary, extract_t f, u_int32_t) {
        secondaries.push_back(std::make_pair(secondary, f));
        return 0;
    }
    int put(DbTxn *, Dbt *k, Dbt *v, u_int32_t) {
        if (!opened) std::abort();
        const std::string ks((const char *)k->get_data(), k->get_size()), vs((const char *)v->get_data(), v->get_size());
        insert(ks, vs);
        for (size_t s = 0; s < secondaries.size(); s++) { // a secondary maps its key to the PRIMARY key
            Dbt out;
            if (secondaries[s].second(secondaries[s].first, k, v, &out)) std::abort();
            const std::string sk((const char *)out.get_data(), out.get_size());
            if (out.get_flags() & DB_DBT_APPMALLOC) std::free(out.get_data());
            secondaries[s].first->insert(sk, ks);
        }
        return 0;
    }
    int sync(u_int32_t) { return 0; }
    int close(u_int32_t) {
        if (!opened) return 0;
        FILE *f = std::fopen(path.c_str(), "w");
        if (!f) std::abort();
        for (size_t i = 0; i < rows.size(); i++) {
            for (size_t j = 0; j < rows[i].k.size(); j++) std::fprintf(f, "%02x", (unsigned char)rows[i].k[j]);
            std::fputc(' ', f);
            for (size_t j = 0; j < rows[i].v.size(); j++) std::fprintf(f, "%02x", (unsigned char)rows[i].v[j]);
            std::fputc('\n', f);
        }
        std::fclose(f);
        opened = false;
        return 0;
    }
};
