"""Loader for liborc_amd.so (hand-written HIP for gfx950 behind the C ABI of include/orc_amd.h).

There is no CPU fallback anywhere in this package: if the shared library is missing, or no HIP
device is visible when a compute entry is called, the call fails loudly.
"""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "liborc_amd.so")


class OrcError(RuntimeError):
    def __init__(self, status, text, detail=""):
        super().__init__("%s (status %d)%s" % (text, status, (": " + detail) if detail else ""))
        self.status = status


def build(force=False):
    """hipcc --offload-arch=gfx950 build of every HIP translation unit (csrc/Makefile)."""
    cmd = ["make", "-C", os.path.join(_HERE, "csrc"), "-j8"]
    if force:
        subprocess.check_call(["make", "-C", os.path.join(_HERE, "csrc"), "clean"], stdout=subprocess.DEVNULL)
    subprocess.check_call(cmd, stdout=subprocess.DEVNULL)
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise OrcError(12, "liborc_amd.so is not built", "run `python -c 'import __graft_entry__ as g; g.build()'`")
        L = C.CDLL(LIB_PATH)
        L.orc_status_string.restype = C.c_char_p
        L.orc_last_error.restype = C.c_char_p
        for name in ("orc_mesh_create", "orc_solver_create"):
            if hasattr(L, name):
                getattr(L, name).restype = C.c_void_p
        for name in ("orc_mesh_n_cells", "orc_mesh_nnz", "orc_last_jacobi_sweeps"):
            if hasattr(L, name):
                getattr(L, name).restype = C.c_int64
        _lib = L
    return _lib


def last_error():
    """orc_last_error(): the library's text for the last failure (or note) of this thread's context"""
    return lib().orc_last_error().decode()


def check(status):
    if status != 0:
        L = lib()
        raise OrcError(status, L.orc_status_string(C.c_int(status)).decode(), L.orc_last_error().decode())


def device_count():
    return lib().orc_device_count()


def init(device=-1):
    check(lib().orc_init(C.c_int(device)))
