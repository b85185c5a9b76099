"""swarm::config mirror (swarm/types/config.{hpp,cpp}): a str->str map loaded from `key=value`
files; `#` comments, blank lines, last assignment wins, case-sensitive (docs/Configuration.dox:20-27).
Typed access follows the reference's sscanf semantics (types/config.hpp:98-135,159-186):
doubles via %lg, ints AND bools via %i -- so `deactivate_on_ejection=1` works and `=true`
silently yields the default."""
import re


class key_not_found(KeyError):
    """types/config.hpp:31-33"""


_INT = re.compile(r"\s*([+-]?(?:0[xX][0-9a-fA-F]+|0[0-7]*|[1-9][0-9]*))")
_FLT = re.compile(r"\s*([+-]?(?:[0-9]+\.?[0-9]*|\.[0-9]+)(?:[eE][+-]?[0-9]+)?|[+-]?(?:inf(?:inity)?|nan))", re.I)


class config(dict):
    @staticmethod
    def load(path, initial=None):
        cfg = config(initial or {})
        with open(path) as f:
            for lineno, line in enumerate(f, 1):
                line = line.strip()
                if not line or line.startswith("#"):
                    continue
                if "=" not in line:
                    raise ValueError("Error in configuration file %s line %d: `%s`" % (path, lineno, line))
                k, v = line.split("=", 1)
                cfg[k.strip()] = v.strip()
        return cfg

    def valid(self, key):
        return key in self

    def _scan(self, key, default):
        s = self[key]
        if isinstance(default, bool) or (isinstance(default, int) and not isinstance(default, float)):
            m = _INT.match(s)
            if not m:
                return None
            v = int(m.group(1), 0)
            return bool(v) if isinstance(default, bool) else v
        if isinstance(default, float):
            m = _FLT.match(s)
            return float(m.group(1)) if m else None
        return s

    def require(self, key, default):
        """cfg.require<T>(key, T()) -- throws key_not_found"""
        if key not in self:
            raise key_not_found(key)
        v = self._scan(key, default)
        if v is None:            # present but not a valid value of type T (types/config.hpp:181-186)
            raise key_not_found(key)
        return v

    def optional(self, key, default):
        if key not in self:
            return default
        v = self._scan(key, default)
        return default if v is None else v


def default_config():
    """swarm/utils.cpp:154-162"""
    return config({"integrator": "hermite_cpu", "time_step": "0.001", "log_writer": "null",
                   "nsys": "16", "nbod": "3"})
