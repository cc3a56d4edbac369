"""``ngsolve.ngstd`` of the numpy-only test stand-in (see the package docstring)."""
from . import TaskManager, Timer  # noqa: F401

__all__ = ["Timer", "TaskManager"]
