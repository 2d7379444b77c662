"""Stage timers of the sub-commands (a measurement aid: tools/bench_cli.py sets SDICE_STAGES=1 and reads TIMES; without
the variable `stage` costs one dictionary look-up)."""
import contextlib
import os
import time

TIMES = {}


def enabled():
    return bool(os.environ.get("SDICE_STAGES"))


@contextlib.contextmanager
def stage(name):
    if not enabled():
        yield
        return
    t = time.perf_counter()
    try:
        yield
    finally:
        TIMES[name] = TIMES.get(name, 0.0) + time.perf_counter() - t


def take():
    """-> dict of the stage seconds since the last call (and the table writers' format / write split)"""
    import ctypes as C
    from . import _ffi
    out = dict(TIMES)
    TIMES.clear()
    io = (C.c_double * 3)()
    _ffi.load().sdice_textio_stats(io, 1)
    out["writer_format_s"], out["writer_write_s"], out["writer_bytes"] = io[0], io[1], io[2]
    return out
