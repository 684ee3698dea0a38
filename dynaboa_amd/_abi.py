"""ctypes prototypes for libdynaboa_hip.so, derived from include/dynaboa_hip.h itself so the
Python side cannot drift from the declared C ABI.  Pointers are passed as integers
(``tensor.data_ptr()``); the library sees plain device addresses and sizes, never torch types."""
from __future__ import annotations

import ctypes
import os
import re
from typing import Dict, List, Tuple

HEADER = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "dynaboa_hip.h")

_SCALARS = {"int": ctypes.c_int, "float": ctypes.c_float, "double": ctypes.c_double, "size_t": ctypes.c_size_t,
            "long long": ctypes.c_longlong, "long": ctypes.c_long, "unsigned long long": ctypes.c_ulonglong, "void": None}


def _ctype(t: str):
    t = " ".join(t.replace("const", " ").split())
    if "*" in t or t == "dyb_stream_t":
        return ctypes.c_void_p
    return _SCALARS[t]


def parse_header(path: str = HEADER) -> Dict[str, Tuple[object, List[object]]]:
    """{symbol: (restype, [argtypes])} for every function declared in the public header."""
    src = open(path).read()
    src = re.sub(r"/\*.*?\*/", " ", src, flags=re.S)
    src = re.sub(r"^\s*#.*$", " ", src, flags=re.M)
    src = src.replace('extern "C" {', " ")
    protos: Dict[str, Tuple[object, List[object]]] = {}
    for m in re.finditer(r"([A-Za-z_][\w\s\*]*?)\b(dyb_\w+)\s*\(([^;{}]*?)\)\s*;", src):
        ret, name, args = m.group(1).strip(), m.group(2), m.group(3).strip()
        if ret.startswith("typedef"):
            continue
        argtypes = []
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                a = re.sub(r"\b\w+$", "", a).strip() if not a.endswith("*") else a   # drop the parameter name
                argtypes.append(_ctype(a))
        protos[name] = (_ctype(ret), argtypes)
    return protos


def bind(lib: ctypes.CDLL) -> ctypes.CDLL:
    """Attach restype/argtypes for every declared symbol; raises if the library lacks one."""
    for name, (res, args) in parse_header().items():
        fn = getattr(lib, name)          # AttributeError -> missing export
        fn.restype = res
        fn.argtypes = args
    return lib


ERRORS = {-1: "bad argument", -2: "kernel launch failed", -3: "unsupported shape", -4: "workspace too small"}


def check(rc: int, what: str) -> None:
    if rc != 0:
        raise RuntimeError(f"libdynaboa_hip: {what} failed: {ERRORS.get(rc, rc)}")
