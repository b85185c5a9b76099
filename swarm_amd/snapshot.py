"""Ensemble file I/O, python twin of swarm/snapshot.{hpp,cpp} (text "SwarmDataFile 2" and the
binary header/sys/body struct stream, snapshot.hpp:153-170)."""
import numpy as np

from .ensemble import HostEnsemble

DEFAULT_IO_TAG, CURRENT_IO_VERSION = "SwarmDataFile", 2


class readfileexception(IOError):
    pass


def load_text(path):
    tok = open(path).read().split()
    if len(tok) < 6 or tok[0] != DEFAULT_IO_TAG:
        raise readfileexception("Error reading %s : Invalid file, header doesn't match" % path)
    if int(tok[1]) != CURRENT_IO_VERSION:
        raise readfileexception("Error reading %s : Incorrect version" % path)
    nbod, nsys, nsattr, nbattr = (int(t) for t in tok[2:6])
    e = HostEnsemble(nsys, nbod, max(nbattr, 1), max(nsattr, 1))
    vals = tok[6:]
    per_sys = 3 + nsattr + nbod * (7 + nbattr)
    if len(vals) != nsys * per_sys:
        raise readfileexception("Error reading %s : File I/O error" % path)
    a = np.array(vals, dtype=object).reshape(nsys, per_sys)
    e.id[:] = a[:, 0].astype(np.int64)
    e.time[:] = a[:, 1].astype(np.float64)
    e.state[:] = a[:, 2].astype(np.int64)
    e.sys_attr[:, :nsattr] = a[:, 3:3 + nsattr].astype(np.float64)
    b = a[:, 3 + nsattr:].astype(np.float64).reshape(nsys, nbod, 7 + nbattr)
    e.mass[:] = b[:, :, 0]
    e.pos[:] = b[:, :, 1:4]
    e.vel[:] = b[:, :, 4:7]
    e.body_attr[:, :, :nbattr] = b[:, :, 7:]
    return e


def save_text(e, path):
    with open(path, "w") as f:
        f.write("%s %i\n%i %i %i %i\n\n\n" % (DEFAULT_IO_TAG, CURRENT_IO_VERSION, e.nbod, e.nsys, e.nsattr, e.nbattr))
        for i in range(e.nsys):
            f.write("%i %.15e %i\n" % (e.id[i], e.time[i], e.state[i]))
            f.write("".join("%.15e " % a for a in e.sys_attr[i]) + "\n")
            for b in range(e.nbod):
                f.write("\t%.15e\n\t%.15e %.15e %.15e\n\t%.15e %.15e %.15e\n\t" % (
                    e.mass[i, b], *e.pos[i, b], *e.vel[i, b]))
                f.write("".join("%.15e " % a for a in e.body_attr[i, b]) + "\n\n")
            f.write("\n")


def _dtypes(nsattr, nbattr, nbod):
    body = np.dtype([("pos", "<f8", 3), ("vel", "<f8", 3), ("mass", "<f8"), ("attribute", "<f8", (nbattr,))])
    sysd = np.dtype([("time", "<f8"), ("id", "<i4"), ("state", "<i4"), ("attribute", "<f8", (nsattr,)),
                     ("body", body, (nbod,))])
    return sysd


def save(e, path):
    rec = np.zeros(e.nsys, dtype=_dtypes(e.nsattr, e.nbattr, e.nbod))
    rec["time"], rec["id"], rec["state"], rec["attribute"] = e.time, e.id, e.state, e.sys_attr
    rec["body"]["pos"], rec["body"]["vel"], rec["body"]["mass"] = e.pos, e.vel, e.mass
    rec["body"]["attribute"] = e.body_attr
    with open(path, "wb") as f:
        np.array([e.nsys, e.nbod, e.nsattr, e.nbattr], dtype="<i4").tofile(f)
        rec.tofile(f)


def load(path):
    with open(path, "rb") as f:
        h = np.fromfile(f, dtype="<i4", count=4)
        if h.size != 4:
            raise readfileexception("Error reading %s : File I/O error" % path)
        nsys, nbod, nsattr, nbattr = (int(x) for x in h)
        rec = np.fromfile(f, dtype=_dtypes(nsattr, nbattr, nbod), count=nsys)
    if rec.size != nsys:
        raise readfileexception("Error reading %s : File I/O error" % path)
    e = HostEnsemble(nsys, nbod, nbattr, nsattr)
    e.time[:], e.id[:], e.state[:], e.sys_attr[:] = rec["time"], rec["id"], rec["state"], rec["attribute"]
    e.pos[:], e.vel[:], e.mass[:] = rec["body"]["pos"], rec["body"]["vel"], rec["body"]["mass"]
    e.body_attr[:] = rec["body"]["attribute"]
    return e
