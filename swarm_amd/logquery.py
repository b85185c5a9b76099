"""Table output for binary log files: the Python counterpart of `swarm query`
(the role of py/swarmng/query.py and of its command line py/bin/swarm-query.py in the reference).

    python -m swarm_amd.logquery -d run.log -s 0..3 -t 0..1 -k jacobi

Three pieces: a `Keplerian` frame that knows how to turn a snapshot into per-planet orbits, row writers (one per
output mode, same column formats as the reference prints so that the two tools can be diffed), and a selection
(which records of the log to walk).  The module-level names `Keplerian`, `TABLE`, `KEYS`, `print_record`,
`run_with_args` are the reference module's.
"""
import argparse
import itertools
import math
import sys

from .logdb import IndexedLogDB
from .range_type import Range, RangeType

RAD2DEG = 180.0 / math.pi
MAX_RECORDS = 10000
TABLE, KEYS = range(2)

CARTESIAN_ROW = "%10d %g  %6d %6d  %9.2g  %10.4g %10.4g %10.4g  %10.5g %10.5g %10.5g  %d\n"
KEPLERIAN_ROW = "%10d %g  %6d %6d  %9.2g  %9.5g %9.5g %9.5g  %9.5g %9.5g %9.5g  %d\n"


class Keplerian:
    """The centre the orbital elements refer to (`-k <frame>`)."""
    choices = ["astrocentric", "barycentric", "jacobi", "origin"]
    ASTROCENTRIC, BARYCENTRIC, JACOBI, ORIGIN = range(4)
    # the LogRecord method that gives the fixed centre of a frame (Jacobi has a centre per planet)
    _centre = {ASTROCENTRIC: "star", BARYCENTRIC: "barycenter", ORIGIN: "origin"}

    def __init__(self, type):
        self.type = self.choices.index(type) if isinstance(type, str) else type

    def __repr__(self):
        return "Keplerian(%s)" % self.choices[self.type]

    def orbits(self, snapshot):
        """(ordinal, body, elements) of every planet of a snapshot record in this frame"""
        if self.type == self.JACOBI:
            return snapshot.bodies_in_keplerian_jacobi()
        if self.type not in self._centre:
            raise ValueError("Wrong keplerian index")
        return snapshot.bodies_in_keplerian(getattr(snapshot, self._centre[self.type])())


def _cartesian_rows(snapshot, frame):
    for ordinal, b in enumerate(snapshot.bodies):
        yield ordinal, CARTESIAN_ROW % ((snapshot.msgid, snapshot.time, snapshot.sys, ordinal, b.mass) + tuple(b.position)
                                        + tuple(b.velocity) + (snapshot.state,))


def _keplerian_rows(snapshot, frame):
    for ordinal, b, el in frame.orbits(snapshot):
        angles = tuple(a * RAD2DEG for a in (el.i, el.O, el.w, el.M))
        yield ordinal, KEPLERIAN_ROW % ((snapshot.msgid, snapshot.time, snapshot.sys, ordinal, b.mass, el.a, el.e) + angles
                                        + (snapshot.state,))


def print_record(print_mode, r, body_range, out=None):
    """One (key, record) pair of a query as text: the bodies of a snapshot that `body_range` admits, as Cartesian
    rows (TABLE), as orbital elements (a Keplerian frame), or the key alone (KEYS)."""
    out = out or sys.stdout
    key, snapshot = r
    if print_mode == KEYS:
        out.write("%r\n" % (key,))
        return
    if snapshot is None:        # an event that is not a snapshot: nothing tabular to show
        return
    if print_mode == TABLE:
        rows = _cartesian_rows(snapshot, None)
    elif isinstance(print_mode, Keplerian):
        rows = _keplerian_rows(snapshot, print_mode)
    else:
        out.write("%r %r\n" % (print_mode, type(print_mode)))
        return
    out.writelines(text for ordinal, text in rows if body_range.contains(ordinal))


def _selection(db, args):
    """the records the arguments ask for, in the order the log database yields them"""
    if args.initial_conditions:
        return db.initial_conditions(args.system_range)
    if args.final_conditions:
        return db.final_conditions(args.system_range)
    return db.query(args.time_range, args.system_range, args.evt_id)


def run_with_args(args, out=None):
    mode = TABLE if args.keplerian is None else Keplerian(args.keplerian)
    for r in itertools.islice(_selection(IndexedLogDB(args.database), args), args.max_records):
        print_record(mode, r, args.body_range, out)


def parse_cmd(argv=None):
    everything = Range.universal()
    p = argparse.ArgumentParser(prog="swarm-query.py")
    p.add_argument("-d", "--database", required=True, help="Log filename (binary log writer)")
    p.add_argument("-m", "--max-records", type=int, default=MAX_RECORDS, help="maximum number of records to process")
    for flag, name, kind, what in (("-s", "--system-range", int, "Range of systems to display"),
                                   ("-t", "--time-range", float, "Range of time to display"),
                                   ("-b", "--body-range", int, "Range of bodies to display"),
                                   ("-e", "--evt-id", int, "The type of event to display (provide codes)")):
        p.add_argument(flag, name, type=RangeType(kind), default=everything, help=what)
    p.add_argument("-k", "--keplerian", choices=Keplerian.choices,
                   help="Keplerian output. The argument defines the type of coordinate system used")
    p.add_argument("--initial-conditions", action="store_true", default=False)
    p.add_argument("--final-conditions", action="store_true", default=False)
    return p.parse_args(argv)


if __name__ == "__main__":
    run_with_args(parse_cmd())
