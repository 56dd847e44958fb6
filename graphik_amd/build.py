"""Compile the HIP extension in-tree: graphik_amd/lib/libgraphik_amd.so (gfx950 only)."""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
SRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "lib", "libgraphik_amd.so")
SOURCES = ["gik_solve.hip"]
HEADERS = ["gik_wave.hip.h", os.path.join(REPO, "include", "graphik_amd.h")]


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(SRC, s) for s in SOURCES] + \
        [h if os.path.isabs(h) else os.path.join(SRC, h) for h in HEADERS]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    """hipcc --offload-arch=gfx950 ... -> graphik_amd/lib/libgraphik_amd.so (idempotent)."""
    if not force and not _stale():
        return LIB
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
           "-I" + os.path.join(REPO, "include"), "-I" + SRC]
    cmd += [os.path.join(SRC, s) for s in SOURCES] + ["-o", LIB]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
