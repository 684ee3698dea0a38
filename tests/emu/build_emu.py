"""Build tests/emu/_build/libdynaboa_emu.so: the product .hip sources compiled for the HOST with
the fake hip_runtime.h of tests/emu/include (test infrastructure only; see that header)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "dynaboa_amd", "csrc")
OUT = os.path.join(HERE, "_build")
LIB = os.path.join(OUT, "libdynaboa_emu.so")
CLANG = "/opt/rocm/lib/llvm/bin/clang++"


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def build(force=False):
    os.makedirs(OUT, exist_ok=True)
    deps = sources() + [os.path.join(HERE, "emu_runtime.cpp"),
                        *sorted(os.path.join(HERE, "include", "hip", f) for f in os.listdir(os.path.join(HERE, "include", "hip"))),
                        *sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".inc"))),
                        os.path.join(os.path.dirname(os.path.dirname(HERE)), "include", "dynaboa_hip.h")]
    if not force and os.path.exists(LIB) and all(os.path.getmtime(LIB) >= os.path.getmtime(d) for d in deps):
        return LIB
    cxx = CLANG if os.path.exists(CLANG) else "clang++"
    objs = []
    flags = ["-std=c++17", "-O2", "-fPIC", "-g0", "-pthread", "-I", os.path.join(HERE, "include"), "-I", CSRC,
             "-Wno-unused-value", "-Wno-vla-cxx-extension"]
    procs = []
    for s in sources() + [os.path.join(HERE, "emu_runtime.cpp")]:
        o = os.path.join(OUT, os.path.basename(s) + ".o")
        objs.append(o)
        procs.append((s, subprocess.Popen([cxx, "-x", "c++", *flags, "-c", s, "-o", o])))
    for s, p in procs:
        if p.wait() != 0:
            raise RuntimeError(f"emu build failed for {s}")
    subprocess.check_call([cxx, "-shared", "-pthread", "-o", LIB, *objs])
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
