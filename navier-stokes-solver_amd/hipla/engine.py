"""Engine registry for the hipla operator protocol.

An *engine* owns device buffers and runs every arithmetic operation the protocol
layer issues (fill / copy / scal / axpy / lincomb / dot / CSR SpMV / block-Jacobi /
diagonal scale and the fused Krylov loops).  The product ships exactly one engine:
`hipla.hip_engine.HipEngine`, a ctypes binding of the hand-written gfx950 kernels in
``csrc/`` (C ABI: ``include/nss_krylov.h``).  It is created lazily on first use and
raises if the shared library or the GPU is missing -- there is no CPU fallback in
the product.

`set_engine()` exists so that the *test suite* can inject the numpy checker engine
that lives under ``oracle/`` (CPU-only CI, gloo multi-process tests, golden-vector
generation).  Nothing in this package imports ``oracle``.
"""

_engine = None


class EngineUnavailable(RuntimeError):
    pass


def set_engine(engine):
    """Install `engine` as the process-wide engine (test hook); returns the previous one."""
    global _engine
    prev = _engine
    _engine = engine
    return prev


def get_engine():
    global _engine
    if _engine is None:
        from .hip_engine import HipEngine  # raises EngineUnavailable loudly
        _engine = HipEngine()
    return _engine


def current_engine_or_none():
    return _engine
