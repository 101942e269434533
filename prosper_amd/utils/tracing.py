"""Trace points for the EM hot path.

Mirrors the hook points of prosper/utils/tracing.py (``tracepoint`` :84-104, ``@traced``
:58-81): the stage labels of the reference (``E_step:iterating``, ``M_step:update W`` ...)
are kept, so a timeline reads like the reference's trace file.  Two consumers:

  * rocTX (``enable_roctx()`` or ``PM_ROCTX=1``): every ``@traced`` method becomes a
    ``roctxRangePush/Pop`` pair and every ``tracepoint`` a ``roctxMark`` -- visible in
    ``rocprofv3 --marker-trace --kernel-trace`` next to the kernels they enqueue.  The library is
    bound with ctypes (librocprofiler-sdk-roctx.so, else the legacy libroctx64.so); without a
    profiler attached the calls are cheap no-ops inside the library.
  * a Python sink (``set_sink(fn)``): ``fn(label, seconds_since_start)`` per event -- the
    stand-in for the reference's per-rank trace file (``set_tracefile``).

With neither installed ``@traced`` and ``tracepoint`` cost one attribute test.
"""
import ctypes
import functools
import os
import time

_sink = None          # callable(label:str, t:float) or None
_roctx = None         # (push, pop, mark) ctypes functions or None
_t0 = time.time()


def set_sink(fn):
    """Install ``fn(label, seconds_since_start)`` as the trace sink (None = off)."""
    global _sink
    _sink = fn


def enable_roctx(on=True):
    """Forward trace points to rocTX.  Returns True when the library was found."""
    global _roctx
    if not on:
        _roctx = None
        return False
    if _roctx is not None:
        return True
    for name in ("librocprofiler-sdk-roctx.so", "libroctx64.so"):
        for prefix in ("", "/opt/rocm/lib/"):
            try:
                lib = ctypes.CDLL(prefix + name)
            except OSError:
                continue
            push, pop, mark = lib.roctxRangePushA, lib.roctxRangePop, lib.roctxMarkA
            push.argtypes, push.restype = [ctypes.c_char_p], ctypes.c_int
            pop.argtypes, pop.restype = [], ctypes.c_int
            mark.argtypes, mark.restype = [ctypes.c_char_p], None
            _roctx = (push, pop, mark)
            return True
    return False


def roctx_enabled():
    return _roctx is not None


def tracepoint(label):
    if _roctx is not None:
        _roctx[2](str(label).encode())
    if _sink is not None:
        _sink(str(label), time.time() - _t0)


def traced(func):
    """Emit ``name:begin`` / ``name:end`` trace points (a rocTX range when enabled) around ``func`` while
    keeping ``__name__``/``__doc__`` (the property test_tracing.py:30-39 checks upstream)."""
    @functools.wraps(func)
    def wrapped(*args, **kwargs):
        if _sink is None and _roctx is None:
            return func(*args, **kwargs)
        rt = _roctx
        if rt is not None:
            rt[0](func.__name__.encode())
        if _sink is not None:
            _sink(func.__name__ + ":begin", time.time() - _t0)
        try:
            return func(*args, **kwargs)
        finally:
            if _sink is not None:
                _sink(func.__name__ + ":end", time.time() - _t0)
            if rt is not None:
                rt[1]()
    return wrapped


if os.environ.get("PM_ROCTX", "0") == "1":
    enable_roctx()
