"""The `bdb` log writer (swarm_amd/csrc/host/bdb_writer.cpp; reference: swarm/log/bdb_writer.cpp:55-94,
bdb_database.cpp:124-203).  This image has no Berkeley DB, so the writer is built here against an in-memory TEST
DOUBLE of the calls it uses (tests/fake_bdb/db_cxx.h) and what it stores is checked against the reference's schema:
five databases, the 8-byte primary key {float time; uint32 event << 24 | system} in (time, integer) order, the record
bytes as values, a later record with the same key replacing the earlier one, the three secondaries cut from the
primary key with sorted duplicates, the two metadata entries.  Parity with a real libdb file is UNPINNED: nothing in
this image can read or write one."""
import os
import struct
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "swarm_amd", "csrc", "host")
GOLDEN = os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="module")
def harness(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("bdb") / "harness")
    subprocess.run(["g++", "-O1", "-std=c++17", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "tests", "fake_bdb"),
                    "-I" + os.path.join(ROOT, "include"), "-I" + HOST, "-DSWARM_NUM_PLANET_ATTRIBUTES=1", "-o", out,
                    os.path.join(ROOT, "tests", "fake_bdb", "harness.cpp"), os.path.join(HOST, "bdb_writer.cpp"),
                    os.path.join(HOST, "swarmng.cpp"), "-L" + os.path.join(ROOT, "swarm_amd", "lib"), "-lswarm_hip",
                    "-Wl,-rpath," + os.path.join(ROOT, "swarm_amd", "lib")], check=True, timeout=300)
    return out


def dump(path):
    return [tuple(bytes.fromhex(x) for x in l.split(" ")) for l in open(path).read().splitlines()]


def test_bdb_writer_stores_the_reference_schema(harness, oracle, tmp_path):
    O = oracle
    # the reference's own logging fixture: hermite_cpu_log on test/bdb/test.4.in.txt -> 16 systems x 11 snapshots
    e = O.load_text(os.path.join(GOLDEN, "bdb", "test.4.in.txt"))
    log = O.Log(1 << 20)
    O.integrate(e, O.params("hermite_cpu", 1.0, time_step=0.01), O.monitor(kind=O.MON_LOG_INTERVAL, log_interval=0.1), log)
    raw = log.bytes()
    recs = O.parse_log(raw)
    assert len(recs) == 176 and all(r["len"] == 288 for r in recs)
    # + an lprintf record (msgid -2: no time, no system) and a second snapshot with the key of the first (a rewrite)
    text = b"hello %d\0"
    printf = struct.pack("<ii", -2, 32) + text + b"\0" * (24 - len(text) - 4) + struct.pack("<i", 7)
    assert len(printf) == 32
    again = bytearray(raw[:288])
    again[40:48] = struct.pack("<d", 123.5)            # same (time, system, event), another body coordinate
    stream = raw + printf + bytes(again)
    rawfn, db = tmp_path / "records.raw", tmp_path / "testing_log.db"
    rawfn.write_bytes(stream)
    p = subprocess.run([harness, str(rawfn), str(db)], capture_output=True, text=True, timeout=60)
    assert p.returncode == 0, p.stderr
    assert "Initializing BDB environment in `%s`" % tmp_path in p.stderr          # bdb_database.cpp:97

    meta = dict(dump(str(db) + ".metadata.dump"))
    assert meta == {b"fileFormatVersion": b"1", b"swarmngVersion": b"1.1"}

    primary = dump(str(db) + ".primary.dump")
    keys = [struct.unpack("<fI", k) for k, _ in primary]
    snaps = [(t, s) for t, s in keys if t == t]
    assert len(snaps) == 176 and len(primary) == 177          # the rewrite replaced its twin, the printf line is there
    assert snaps == sorted(snaps)                             # B-tree order: float time, then event << 24 | system
    want = sorted((np.float32(r["time"]).item(), (1 << 24) | r["sys"]) for r in recs)
    assert snaps == want
    by_key = {k: v for k, v in primary}
    first_key = struct.pack("<fI", np.float32(recs[0]["time"]), (1 << 24) | recs[0]["sys"])
    assert by_key[first_key] == bytes(again)                  # the later record with the same key won
    for i, r in enumerate(recs[1:], 1):
        k = struct.pack("<fI", np.float32(r["time"]), (1 << 24) | r["sys"])
        assert by_key[k] == raw[288 * i:288 * (i + 1)]
    nan_keys = [(k, v) for k, v in primary if struct.unpack("<f", k[:4])[0] != struct.unpack("<f", k[:4])[0]]
    assert len(nan_keys) == 1 and nan_keys[0][1] == printf and struct.unpack("<I", nan_keys[0][0][4:])[0] == 0xFE000000

    # secondaries: key cut from the primary key, value = the primary key, duplicates in primary-key order
    sysidx = dump(str(db) + ".system_idx.dump")
    assert len(sysidx) == 177
    assert sorted(struct.unpack("<I", k)[0] for k, _ in sysidx) == sorted([s & 0xFFFFFF for _, s in keys])
    for sid in range(16):
        mine = [struct.unpack("<fI", v) for k, v in sysidx if struct.unpack("<I", k)[0] == sid and v[:4] != nan_keys[0][0][:4]]
        assert len(mine) == 11 and mine == sorted(mine)
    evidx = dump(str(db) + ".event_idx.dump")
    assert sorted(k for k, _ in evidx) == [b"\x01"] * 176 + [b"\xfe"]
    tidx = dump(str(db) + ".time_idx.dump")
    assert len(tidx) == 177 and all(k == v[:4] for k, v in tidx)


def test_makefile_builds_the_writer_only_where_libdb_exists():
    """src/plugins.cmake:36-39 adds the BDB plugin `if(BDB_FOUND)`; the Makefile probes for <db_cxx.h> the same way"""
    mk = open(os.path.join(ROOT, "swarm_amd", "csrc", "Makefile")).read()
    assert "db_cxx.h" in mk and "bdb_writer.cpp" in mk and "-ldb_cxx" in mk
    have = subprocess.run("g++ -E -x c++ -include db_cxx.h /dev/null > /dev/null 2>&1", shell=True).returncode == 0
    listed = subprocess.run([os.path.join(ROOT, "swarm_amd", "bin", "swarm"), "-p"], capture_output=True, text=True)
    assert ("writer_bdb" in listed.stdout + listed.stderr) == have
