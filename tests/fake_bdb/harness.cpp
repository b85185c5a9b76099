// tests/fake_bdb/harness.cpp -- feeds a raw gpulog record stream to the `bdb` writer plugin in two process() calls, the
// way log::manager::flush does after two launches.  Built by tests/test_bdb_writer.py against the test double db_cxx.h.
//   harness <records.raw> <log_output_db>
#include <fstream>
#include <iterator>
#include <vector>

#include "swarm/swarm.h"

int main(int argc, char **argv) {
    if (argc != 3) return 2;
    std::ifstream in(argv[1], std::ios::binary);
    std::vector<char> raw((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
    swarm::config cfg;
    cfg["log_writer"] = "bdb";
    cfg["log_output_db"] = argv[2];
    swarm::log::writer *w = swarm::log::writer::create(cfg);
    // split at a record boundary near the middle
    size_t off = 0, cut = 0;
    while (off + 8 <= raw.size()) {
        int len;
        std::memcpy(&len, &raw[off + 4], 4);
        if (off >= raw.size() / 2 && !cut) cut = off;
        off += (size_t)len;
    }
    w->process(raw.data(), cut);
    w->process(raw.data() + cut, raw.size() - cut);
    delete w;
    return 0;
}
