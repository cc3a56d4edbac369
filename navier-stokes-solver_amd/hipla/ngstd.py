"""``hipla.ngstd``: Timer / TaskManager with the reference's call surface
(``Timer(name).Start()/.Stop()/.time``: bramble_pasciak_cg.py:68-72,111-113;
solvers/bramblepasciak_new.py:111-116,194-196,251-253; run.py:34,50,201;
``with TaskManager():`` run.py:239).

Kernels are asynchronous on the engine's stream, so Start/Stop drain the stream
first: ``timer.time`` is wall time of finished device work, as the reference's
synchronous CPU timers report."""

import time

from .engine import current_engine_or_none

_timers = {}


class Timer:
    def __init__(self, name=""):
        self.name = name
        self.time = 0.0
        self.count = 0
        self._t0 = None
        _timers[name] = self

    @staticmethod
    def _drain():
        eng = current_engine_or_none()
        if eng is not None:
            eng.synchronize()

    def Start(self):
        self._drain()
        self._t0 = time.perf_counter()

    def Stop(self):
        if self._t0 is None:
            return
        self._drain()
        self.time += time.perf_counter() - self._t0
        self.count += 1
        self._t0 = None

    def __enter__(self):
        self.Start()
        return self

    def __exit__(self, *exc):
        self.Stop()
        return False


def Timers():
    return [{"name": t.name, "time": t.time, "counts": t.count} for t in _timers.values()]


class TaskManager:
    """Context manager kept for source compatibility; the engine's parallelism is
    the GPU grid, there is no host thread pool to start."""

    def __init__(self, pajetrace=None):
        self.pajetrace = pajetrace

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


def SetHeapSize(nbytes):
    return None
