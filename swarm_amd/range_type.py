"""Ranges for log queries: a single value, a closed interval, or everything.

Mirror of the reference's py/swarmng/range_type.py (Range L50-166, RangeType L168-182) for Python 3:
same constructors (`Range.single`, `Range.interval`, `Range.universal`), same inclusive upper bound,
same `'a..b'` grammar for command lines."""
import argparse

SINGLE, INTERVAL, UNIVERSAL = range(3)


class Range:
    def __init__(self, kind, lo=None, hi=None):
        self.type, self._lo, self._hi = kind, lo, hi

    @staticmethod
    def single(s):
        return Range(SINGLE, s, s)

    @staticmethod
    def interval(t, u=None):
        """Range.interval((a, b)) or Range.interval(a, b); both bounds belong to the range."""
        lo, hi = (t, u) if u is not None else t
        return Range(INTERVAL, lo, hi)

    @staticmethod
    def universal():
        return Range(UNIVERSAL)

    def isInterval(self):
        return self.type == INTERVAL

    def isSingle(self):
        return self.type == SINGLE

    def isUniversal(self):
        return self.type == UNIVERSAL

    def contains(self, i):
        return self.type == UNIVERSAL or self._lo <= i <= self._hi

    def with_limits(self, lower, upper):
        """Integers of the range clamped to [lower, upper]; makes a universal range iterable."""
        lo, hi = (lower, upper) if self.type == UNIVERSAL else (max(self._lo, lower), min(self._hi, upper))
        return iter(range(lo, hi + 1))

    def __iter__(self):
        if self.type == UNIVERSAL:
            return iter(())
        return iter(range(self._lo, self._hi + 1))

    def _bounded(self):
        if self.type == UNIVERSAL:
            raise ValueError("a universal range has no bounds")

    def lower(self):
        self._bounded()
        return self._lo

    def upper(self):
        self._bounded()
        return self._hi

    def ulPair(self):
        self._bounded()
        return (self._lo, self._hi)

    def __repr__(self):
        if self.type == SINGLE:
            return "[ %r]" % (self._lo,)
        if self.type == INTERVAL:
            return "[ %r..%r]" % (self._lo, self._hi)
        return "[ .. ]"


class RangeType:
    """argparse type: '3' -> single, '1..2' -> interval"""

    def __init__(self, type=int):
        self.type = type

    def __call__(self, string):
        x = string.split("..")
        if len(x) == 2:
            return Range.interval(self.type(x[0]), self.type(x[1]))
        if len(x) == 1:
            return Range.single(self.type(x[0]))
        raise argparse.ArgumentTypeError("Invalid range '%s' valid range should be like '1..2'" % string)
