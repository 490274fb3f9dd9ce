"""Run-time switches (``GEOSSL_*`` environment variables) as plain dict lookups.

The switches are read where they act, every call, so that a test or a bench line can flip one in the middle of a process.
``os.environ.get`` encodes the key, looks it up and decodes the value on every call; a replayed step asks for a dozen of
them.  ``env`` does the lookup in ``os.environ``'s own backing dict with keys encoded once (CPython on POSIX keeps the
process environment as ``bytes -> bytes`` in ``os.environ._data``, updated by every ``os.environ[...] = ...`` /
``monkeypatch.setenv``): same answers, a tenth of the cost.  Anything unexpected about the interpreter falls back to
``os.environ.get``."""
import os

_DATA = getattr(os.environ, "_data", None)
_ENCODE = getattr(os.environ, "encodekey", None)
_DECODE = getattr(os.environ, "decodevalue", None)
if not isinstance(_DATA, dict) or _ENCODE is None or _DECODE is None:
    _DATA = None
_KEYS = {}


def env(name, default=None):
    """``os.environ.get(name, default)``."""
    if _DATA is None:
        return os.environ.get(name, default)
    key = _KEYS.get(name)
    if key is None:
        key = _KEYS[name] = _ENCODE(name)
    value = _DATA.get(key)
    return default if value is None else _DECODE(value)
