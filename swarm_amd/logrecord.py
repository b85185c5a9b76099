"""Event records of the device log as Python objects.

Mirror of the reference's py/swarmng/logrecord.py for Python 3: `LogRecord.from_binary` decodes the
EVT_SNAPSHOT record the kernels write (swarm/log/log.hpp:313-326: header {int msgid, int len},
double time, int sys, int state, int nbod, pad to 32, then per body 7 doubles + int body_id + pad,
64 bytes), `center_of_mass` / `bodies_in_keplerian*` are the helpers of L20-130."""
from collections import namedtuple
from struct import unpack_from

KeplerianCoordinates = namedtuple("KeplerianCoordinates", ["a", "e", "i", "O", "w", "M"])


def keplerian_for_cartesian(planet, center):
    """Orbital elements of `planet` about `center`, both ((x,y,z),(vx,vy,vz),mass); GM = the two masses
    (py/swarmng/__init__.py:107-111)."""
    from .swarmng import calc_keplerian_for_cartesian
    (x, y, z), (vx, vy, vz), mass = planet
    (cx, cy, cz), (cvx, cvy, cvz), cmass = center
    return KeplerianCoordinates(*calc_keplerian_for_cartesian(x - cx, y - cy, z - cz, vx - cvx, vy - cvy, vz - cvz,
                                                              mass + cmass))


def center_of_mass(bodies):
    """Mass-weighted mean position and velocity and the total mass of [((x,y,z),(vx,vy,vz),mass), ...].
    (The reference's helper of this name, logrecord.py:20-27, adds the coordinates unweighted; its C++
    counterpart query.cpp:95-119 weights them, and so does this.)"""
    c = [0.0] * 6
    cmass = 0.0
    for (x, y, z), (vx, vy, vz), mass in bodies:
        for k, q in enumerate((x, y, z, vx, vy, vz)):
            c[k] += q * mass
        cmass += mass
    return ((c[0] / cmass, c[1] / cmass, c[2] / cmass), (c[3] / cmass, c[4] / cmass, c[5] / cmass), cmass)


def with_index(lst, starting_index=0):
    return list(zip(range(starting_index, starting_index + len(lst)), lst))


class LogRecord:
    """One snapshot of a system: .msgid .time .sys .state .bodies (list of Body)."""
    Body = namedtuple("Body", ["position", "velocity", "mass"])

    def star(self):
        return self.bodies[0]

    def barycenter(self):
        return center_of_mass(self.bodies)

    def origin(self):
        return LogRecord.Body((0, 0, 0), (0, 0, 0), self.bodies[0].mass)

    def bodies_in_keplerian(self, center):
        """(ordinal, Body, KeplerianCoordinates) for every planet, about the given centre
        (self.star(), self.barycenter() or self.origin())"""
        for i, b in with_index(self.bodies)[1:]:
            yield i, b, keplerian_for_cartesian(b, center)

    def bodies_in_keplerian_jacobi(self):
        """as above about the centre of mass of the star and all interior planets"""
        for i, b in with_index(self.bodies)[1:]:
            yield i, b, keplerian_for_cartesian(b, center_of_mass(self.bodies[0:i]))

    @staticmethod
    def from_binary(s):
        """Decode one record; only snapshots (msgid 1) are decoded, anything else gives None."""
        msgid, _length = unpack_from("<ii", s, 0)
        if msgid != 1:
            return None
        rec = LogRecord()
        rec.msgid = msgid
        rec.time, rec.sys, rec.state, nbod = unpack_from("<diii", s, 8)
        rec.bodies = []
        for b in range(nbod):
            x, y, z, vx, vy, vz, mass, _body_id = unpack_from("<dddddddi", s, 32 + b * 64)
            rec.bodies.append(LogRecord.Body((x, y, z), (vx, vy, vz), mass))
        return rec

    def as_map(self):
        return {"time": self.time, "sys": self.sys, "state": self.state, "bodies": self.bodies}

    def __repr__(self):
        return repr(self.as_map())
