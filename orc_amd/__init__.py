"""orc_amd — MI355X-native implementation of ORC's per-SIMPLE-iteration hot path.

Host-side mirror of the reference's module surface (discretization, linear_algebra, solver, mesh,
settings) on top of the C ABI in include/orc_amd.h.  All compute runs in liborc_amd.so (HIP,
gfx950); nothing here falls back to a CPU path.
"""
from . import _lib
from ._lib import OrcError, build, device_count, init  # noqa: F401


def reload_environment():
    """orc_reload_environment(): the library reads its ORC_* switches once (at the first orc_init); a caller that changes one afterwards says so."""
    _lib.check(_lib.lib().orc_reload_environment())
