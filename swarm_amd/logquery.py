"""Table output for log files, the Python counterpart of `swarm query`.

Mirror of the reference's py/swarmng/query.py (print_record L31-66, run_with_args L73-91) and of
its command line py/bin/swarm-query.py (L13-27), for Python 3 and over binary log files:

    python -m swarm_amd.logquery -d run.log -s 0..3 -t 0..1 -k jacobi
"""
import argparse
import itertools
import math
import sys

from .logdb import IndexedLogDB
from .range_type import Range, RangeType

RAD2DEG = 180.0 / math.pi
MAX_RECORDS = 10000


class Keplerian:
    ASTROCENTRIC, BARYCENTRIC, JACOBI, ORIGIN = range(4)
    choices = ["astrocentric", "barycentric", "jacobi", "origin"]

    def __init__(self, type):
        self.type = Keplerian.choices.index(type) if isinstance(type, str) else type

    def __repr__(self):
        return "Keplerian(%s)" % Keplerian.choices[self.type]


TABLE, KEYS = range(2)


def _g(v, spec):
    return spec % v


def print_record(print_mode, r, body_range, out=None):
    out = out or sys.stdout
    k, l = r
    if l is None:           # not a snapshot record
        return
    if print_mode == TABLE:
        for i, b in enumerate(l.bodies):
            if body_range.contains(i):
                out.write("%10d %g  %6d %6d  %9.2g  %10.4g %10.4g %10.4g  %10.5g %10.5g %10.5g  %d\n"
                          % (l.msgid, l.time, l.sys, i, b.mass, b.position[0], b.position[1], b.position[2],
                             b.velocity[0], b.velocity[1], b.velocity[2], l.state))
    elif isinstance(print_mode, Keplerian):
        tp = print_mode.type
        if tp == Keplerian.JACOBI:
            body_orbits = l.bodies_in_keplerian_jacobi()
        else:
            if tp == Keplerian.ASTROCENTRIC:
                center = l.star()
            elif tp == Keplerian.BARYCENTRIC:
                center = l.barycenter()
            elif tp == Keplerian.ORIGIN:
                center = l.origin()
            else:
                raise ValueError("Wrong keplerian index")
            body_orbits = l.bodies_in_keplerian(center)
        for i, b, orbit in body_orbits:
            if body_range.contains(i):
                out.write("%10d %g  %6d %6d  %9.2g  %9.5g %9.5g %9.5g  %9.5g %9.5g %9.5g  %d\n"
                          % (l.msgid, l.time, l.sys, i, b.mass, orbit.a, orbit.e, orbit.i * RAD2DEG, orbit.O * RAD2DEG,
                             orbit.w * RAD2DEG, orbit.M * RAD2DEG, l.state))
    elif print_mode == KEYS:
        out.write("%r\n" % (k,))
    else:
        out.write("%r %r\n" % (print_mode, type(print_mode)))


def truncate(number, iterator):
    return itertools.islice(iterator, number)


def run_with_args(args, out=None):
    d = IndexedLogDB(args.database)
    print_mode = Keplerian(args.keplerian) if args.keplerian is not None else TABLE
    if args.initial_conditions:
        q = d.initial_conditions(args.system_range)
    elif args.final_conditions:
        q = d.final_conditions(args.system_range)
    else:
        q = d.query(args.time_range, args.system_range, args.evt_id)
    for r in truncate(args.max_records, q):
        print_record(print_mode, r, args.body_range, out)


def parse_cmd(argv=None):
    parser = argparse.ArgumentParser(prog="swarm-query.py")
    parser.add_argument("-d", "--database", help="Log filename (binary log writer)", required=True)
    parser.add_argument("-m", "--max-records", help="maximum number of records to process", default=MAX_RECORDS, type=int)
    parser.add_argument("-s", "--system-range", help="Range of systems to display", type=RangeType(int), default=Range.universal())
    parser.add_argument("-t", "--time-range", help="Range of time to display", type=RangeType(float), default=Range.universal())
    parser.add_argument("-b", "--body-range", help="Range of bodies to display", type=RangeType(int), default=Range.universal())
    parser.add_argument("-e", "--evt-id", help="The type of event to display (provide codes)", type=RangeType(int),
                        default=Range.universal())
    parser.add_argument("-k", "--keplerian", help="Keplerian output. The argument defines the type of coordinate system used",
                        choices=Keplerian.choices)
    parser.add_argument("--initial-conditions", action="store_true", default=False)
    parser.add_argument("--final-conditions", action="store_true", default=False)
    return parser.parse_args(argv)


if __name__ == "__main__":
    run_with_args(parse_cmd())
