"""Indexed access to a log file written by the `binary` log writer.

The reference's py/swarmng/logdb.py serves these queries from the Berkeley-DB file of its `bdb`
writer (primary key (time, event id, system id), secondary indices by system, time and event).
libdb is not part of this image, so the same interface -- PKey, IndexedLogDB and its query methods,
same argument meaning, same (PKey, LogRecord) pairs in the same order -- is served here from the
binary log (swarm/log/binary_writer.cpp, fileformat.hpp: 96-byte swarm_header + the raw record
stream, sorted or not), indexed in memory when the file is opened.

The key follows the bdb writer's: time rounded to float32, event id, system id (logdb.py:35-66)."""
import bisect
import os
import struct
import sys as _sys

import numpy as np

from . import gpulog
from .logrecord import LogRecord

HEADER_BYTES = 96           # swarm_header, swarm/log/fileformat.hpp:28-68


class PKey:
    """Primary key of a record: .time (float32-rounded), .event_id, .system_id"""

    def __init__(self, time, event_id, system_id):
        self.time = float(np.float32(time))
        self.event_id = int(event_id)
        self.system_id = int(system_id)

    def astuple(self):
        return (self.time, self.event_id, self.system_id)

    @staticmethod
    def decomposeBinary(bin):
        if len(bin) != 8:
            return None
        t, se = struct.unpack("<fI", bin)
        return (t, se >> 24, se & 0xFFFFFF)

    @staticmethod
    def fromBinary(bin):
        return PKey(*PKey.decomposeBinary(bin))

    def toBinary(self):
        return struct.pack("<fI", self.time, self.event_id * 2 ** 24 + self.system_id)

    @staticmethod
    def compareBinary(s1, s2):
        d1, d2 = PKey.decomposeBinary(s1), PKey.decomposeBinary(s2)
        return 0 if d1 == d2 else (-1 if d1 < d2 else 1)

    def __repr__(self):
        return repr({"time": self.time, "sys": self.system_id, "evt": self.event_id})


class IndexedLogDB:
    fileFormatVersion = "0"   # swarm_header.version of the binary files

    def __init__(self, pathName):
        with open(pathName, "rb") as f:
            raw = f.read()
        if len(raw) < HEADER_BYTES or raw[:6] != b"SWARM\0":
            raise RuntimeError("'%s' is not a Swarm log file" % pathName)
        self._meta = {"fileFormatVersion": raw[6:8].split(b"\0")[0].decode(),
                      "type": raw[8:84].split(b"\0")[0].decode().split("//")[0].strip(),
                      "path": os.path.abspath(pathName)}
        self.validateVersionInfo()
        recs = []
        for pos, (msgid, rec) in enumerate(gpulog.records(raw[HEADER_BYTES:])):
            if msgid <= 0:          # printf and internal messages carry no (time, system)
                continue
            t, s = struct.unpack_from("<di", rec, 8)
            recs.append((PKey(t, msgid, s).astuple(), pos, rec))
        recs.sort(key=lambda r: (r[0], r[1]))           # primary order: time, event, system (then file order)
        self._keys = [r[0] for r in recs]
        self._recs = [r[2] for r in recs]
        # the `system_idx` secondary index: (system, primary key) order
        self._by_sys = sorted(range(len(recs)), key=lambda i: (self._keys[i][2], self._keys[i], i))
        self._sys_keys = [self._keys[i][2] for i in self._by_sys]

    # ---- metadata
    def getMetadata(self, name):
        return self._meta.get(name)

    def validateVersionInfo(self):
        v = self.getMetadata("fileFormatVersion")
        if v != self.fileFormatVersion:
            raise RuntimeError("Mismatching file format version: {0}, required {1}".format(v, self.fileFormatVersion))

    # ---- helpers
    def _kvp(self, i):
        return (PKey(*self._keys[i]), LogRecord.from_binary(self._recs[i]))

    decodeKVP = staticmethod(lambda r: (PKey.fromBinary(r[0]), LogRecord.from_binary(r[1])))

    @staticmethod
    def _f32(t):
        return float(np.float32(t))

    # ---- queries; every iterable yields (PKey, LogRecord) unless said otherwise
    def all_records(self):
        """all records by time, then event id, then system id"""
        for i in range(len(self._keys)):
            yield self._kvp(i)

    def time_sequence(self, time_range):
        """the distinct (float32) times t0 <= t <= t1 that have records"""
        t0, t1 = time_range
        i = bisect.bisect_left(self._keys, (self._f32(t0),))
        last = None
        while i < len(self._keys) and self._keys[i][0] <= t1:
            if self._keys[i][0] != last:
                last = self._keys[i][0]
                yield last
            i += 1

    def system_range_records(self, sys_range):
        s0, s1 = sys_range
        j = bisect.bisect_left(self._sys_keys, s0)
        while j < len(self._by_sys) and self._sys_keys[j] <= s1:
            yield self._kvp(self._by_sys[j])
            j += 1

    def time_range_records(self, time_range):
        t0, t1 = time_range
        i = bisect.bisect_left(self._keys, (self._f32(t0),))
        while i < len(self._keys) and self._keys[i][0] <= t1:
            yield self._kvp(i)
            i += 1

    def system_at_time(self, sysid, time_range):
        t0, t1 = time_range
        j = bisect.bisect_left(self._sys_keys, sysid)
        while j < len(self._by_sys) and self._sys_keys[j] == sysid:
            i = self._by_sys[j]
            if self._keys[i][0] > t1:
                break
            if self._keys[i][0] >= self._f32(t0):
                yield self._kvp(i)
            j += 1

    def _system_bounds(self, system_range):
        if system_range.isUniversal():
            return 0, _sys.maxsize
        return system_range.ulPair()

    def initial_conditions(self, system_range):
        """(system id, LogRecord) of the earliest record of every system in the range"""
        s0, s1 = self._system_bounds(system_range)
        j = bisect.bisect_left(self._sys_keys, s0)
        while j < len(self._by_sys) and self._sys_keys[j] <= s1:
            sysid = self._sys_keys[j]
            yield sysid, LogRecord.from_binary(self._recs[self._by_sys[j]])
            j = bisect.bisect_right(self._sys_keys, sysid)

    def final_conditions(self, system_range):
        """(system id, LogRecord) of the latest record of every system in the range"""
        s0, s1 = self._system_bounds(system_range)
        j = bisect.bisect_left(self._sys_keys, s0)
        while j < len(self._by_sys) and self._sys_keys[j] <= s1:
            sysid = self._sys_keys[j]
            j = bisect.bisect_right(self._sys_keys, sysid)
            yield sysid, LogRecord.from_binary(self._recs[self._by_sys[j - 1]])

    def system_range_for_time_event(self, time, event_id, system_range):
        s0, s1 = system_range
        t = self._f32(time)
        i = bisect.bisect_left(self._keys, (t, event_id, s0))
        while i < len(self._keys) and self._keys[i][:2] == (t, event_id) and self._keys[i][2] <= s1:
            yield self._kvp(i)
            i += 1

    def query(self, tr, sr, er):
        """records in the time Range tr, system Range sr and event-id Range er (swarmng.range_type.Range)"""
        q = self.queryInternal(tr, sr)
        if er.isUniversal():
            return q
        return ((k, l) for k, l in q if er.contains(k.event_id))

    def queryInternal(self, tr, sr):
        if tr.isUniversal():
            if sr.isUniversal():
                yield from self.all_records()
            else:
                yield from self.system_range_records(sr.ulPair())
        elif sr.isUniversal():
            yield from self.time_range_records(tr.ulPair())
        else:
            for t in self.time_sequence(tr.ulPair()):
                yield from self.system_range_for_time_event(t, 1, sr.ulPair())
