"""Loader for libdynaboa_hip.so (built in-tree by ``python -m dynaboa_amd.build`` /
``__graft_entry__.build()``).  There is NO fallback: if the library is missing every compute
entry point raises - the adaptation path never silently runs on a CPU or eager-PyTorch route."""
from __future__ import annotations

import ctypes
import os

from . import _abi

LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libdynaboa_hip.so")
_lib = None


class MissingExtension(RuntimeError):
    pass


def load() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise MissingExtension(
                f"{LIB_PATH} not found - build it with `python -m dynaboa_amd.build` "
                "(hipcc --offload-arch=gfx950). dynaboa_amd has no CPU / eager fallback.")
        _lib = _abi.bind(ctypes.CDLL(LIB_PATH))
    return _lib


def use_library(lib: ctypes.CDLL) -> None:
    """Test hook: drive the host layer with an already-bound library object."""
    global _lib
    _lib = lib
