"""Build numbotics_amd/csrc/libnbk.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

    python -m numbotics_amd.csrc.build [--force]

-ffp-contract=off is part of the arithmetic contract: the only fused multiply-adds are
the ones written as NBK_FMA in the sources.
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "libnbk.so")
SOURCES = ["nbk.hip", "nbk_device.hpp", os.path.join("..", "..", "include", "nbk.h"), "build.py"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared",
         "-fno-gpu-rdc", "-Wall", "-Wno-unused-function", "-Wno-pass-failed"]


def source_digest() -> str:
    """sha256 (first 16 hex digits) of what determines the device code -- nbk.hip, nbk_device.hpp and the compiler flags:
    profiles record it, bench.py refuses counters of another build.  (The C header and this script only declare / drive.)"""
    import hashlib
    h = hashlib.sha256()
    for name in ("nbk.hip", "nbk_device.hpp"):
        with open(os.path.join(HERE, name), "rb") as f:
            h.update(f.read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()[:16]


def hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC or install ROCm)")


def up_to_date():
    if not os.path.exists(LIB):
        return False
    t = os.path.getmtime(LIB)
    return all(os.path.getmtime(os.path.join(HERE, s)) <= t for s in SOURCES)


ABLATE_LIB = os.path.join(HERE, "libnbk_ablate.so")


def build_ablate(verbose: bool = False) -> str:
    """The diagnostic build with the NBK_ABLATE switches compiled in (tools/ablate.py, tools/narrow_prof.py).  Never loaded by
    the package: tools point numbotics_amd._lib.LIB_PATH at it themselves."""
    cmd = [hipcc()] + FLAGS + ["-DNBK_ABLATE_BUILD", os.path.join(HERE, "nbk.hip"), "-o", ABLATE_LIB]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True, cwd=HERE)
    return ABLATE_LIB


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and up_to_date():
        return LIB
    cmd = [hipcc()] + FLAGS + [os.path.join(HERE, "nbk.hip"), "-o", LIB]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True, cwd=HERE)
    return LIB


if __name__ == "__main__":
    if "--ablate" in sys.argv:
        print(build_ablate(verbose=True))
    else:
        print(build(force="--force" in sys.argv, verbose=True))
