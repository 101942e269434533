"""Trace points for the EM hot path.

Mirrors the hook points of prosper/utils/tracing.py (``tracepoint`` :84-104, ``@traced``
:58-81) so the stage labels of the reference (``E_step:iterating``, ``M_step:update W`` ...)
line up with rocprofv3 timelines: when a sink is installed each label is forwarded to it
(bench.py installs a HIP-event based stage timer); with no sink both are no-ops.
"""
import functools
import time

_sink = None          # callable(label:str, t:float) or None
_t0 = time.time()


def set_sink(fn):
    """Install ``fn(label, seconds_since_start)`` as the trace sink (None = off)."""
    global _sink
    _sink = fn


def tracepoint(label):
    if _sink is not None:
        _sink(str(label), time.time() - _t0)


def traced(func):
    """Emit ``name:begin`` / ``name:end`` trace points around ``func`` while keeping
    ``__name__``/``__doc__`` (the property test_tracing.py:30-39 checks upstream)."""
    @functools.wraps(func)
    def wrapped(*args, **kwargs):
        if _sink is None:
            return func(*args, **kwargs)
        tracepoint(func.__name__ + ":begin")
        try:
            return func(*args, **kwargs)
        finally:
            tracepoint(func.__name__ + ":end")
    return wrapped
