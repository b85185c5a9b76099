"""Host side of the device event buffer: record iteration and the printf replay.

  gpulog::ilogstream / logrecord   swarm/log/gpulog/bits/gpulog_ilogstream.h, gpulog_logrecord.h
  run_printf / replay_printf       swarm/log/gpulog/lprintf.h:74-195

Record = header {int32 msgid, int32 len} + payload, len padded to 16 (bits/gpulog_types.h:60-99).
EVT_SNAPSHOT = 1 (swarm/log/log.hpp:313-326), MSG_PRINTF = -2 (bits/gpulog_constants.h:35)."""
import re
import struct

EVT_SNAPSHOT = 1
MSG_PRINTF = -2

_SPEC = re.compile(r"%[^%cdiouxXeEfgGaAnps]*([%cdiouxXeEfgGaAnps])")


def records(raw):
    """Yield (msgid, bytes-of-record) for every record of a flushed buffer."""
    raw = bytes(raw)
    off, n = 0, len(raw)
    while off + 8 <= n:
        msgid, ln = struct.unpack_from("<ii", raw, off)
        if ln <= 0 or off + ln > n:
            raise ValueError("corrupt log record at byte %d (len=%d)" % (off, ln))
        yield msgid, raw[off:off + ln]
        off += ln


def run_printf(rec):
    """Format one MSG_PRINTF record: format string at offset 8 (NUL-terminated char[N]), then every
    argument at its natural alignment -- int32 for %c %d %i %o %u %x %X %p, double for
    %e %E %f %g %G %a %A (lprintf.h:117-144)."""
    msgid, _ = struct.unpack_from("<ii", rec, 0)
    if msgid != MSG_PRINTF:
        return ""
    end = rec.index(b"\0", 8)
    fmt = rec[8:end].decode("latin-1")
    at, out, pos = end + 1, [], 0
    for m in _SPEC.finditer(fmt):
        out.append(fmt[pos:m.start()])
        pos = m.end()
        spec, k = m.group(0), m.group(1)
        pyspec = spec.replace("l", "").replace("h", "")
        if k in "cdiouxXp":
            at = (at + 3) // 4 * 4
            (v,) = struct.unpack_from("<i", rec, at)
            at += 4
            out.append(chr(v) if k == "c" else (pyspec.replace("p", "x").replace("u", "d").replace("i", "d") % v))
        elif k in "eEfgGaA":
            at = (at + 7) // 8 * 8
            (v,) = struct.unpack_from("<d", rec, at)
            at += 8
            out.append(float.hex(v) if k in "aA" else pyspec % v)
        elif k == "%":
            out.append("%%")  # lprintf.h:154-156 prints both characters
        else:
            out.append(spec)
    out.append(fmt[pos:])
    return "".join(out)


def replay_printf(stream, raw):
    """replay_printf(std::ostream&, log), lprintf.h:174-195: print every printf record; returns count."""
    count = 0
    for msgid, rec in records(raw):
        if msgid == MSG_PRINTF:
            stream.write(run_printf(rec))
            count += 1
    return count
