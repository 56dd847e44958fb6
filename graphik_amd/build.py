"""Compile the HIP extension in-tree: graphik_amd/lib/libgraphik_amd.so (gfx950 only).

    python -m graphik_amd.build            # the shipped library
    python -m graphik_amd.build --dev      # + lib/exp/libgraphik_amd_dev.so (-DGIK_DEV: the
                                           #   micro-benchmark kernel and the gik_debug_* hooks the
                                           #   tools/dev_*.py probes use; never loaded by the package)
"""
import glob
import hashlib
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
SRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "lib", "libgraphik_amd.so")
DEV_LIB = os.path.join(HERE, "lib", "exp", "libgraphik_amd_dev.so")
SOURCES = ["gik_solve.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared"]


def dependencies():
    """Every file the library is compiled from: all of csrc/ plus the public headers."""
    deps = sorted(glob.glob(os.path.join(SRC, "*.hip")) + glob.glob(os.path.join(SRC, "*.h")) +
                  glob.glob(os.path.join(REPO, "include", "*.h")))
    assert all(os.path.join(SRC, s) in deps for s in SOURCES)
    return deps


def source_digest(extra=()):
    """Content hash of the sources and the compile flags (mtimes lie after a checkout)."""
    h = hashlib.sha256(" ".join(FLAGS + list(extra)).encode())
    for d in dependencies():
        h.update(os.path.relpath(d, REPO).encode())
        h.update(open(d, "rb").read())
    return h.hexdigest()


def _stale(lib, extra=()):
    stamp = lib + ".digest"
    if not (os.path.exists(lib) and os.path.exists(stamp)):
        return True
    return open(stamp).read().strip() != source_digest(extra)


def _compile(lib, extra, verbose):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        # a machine without the toolchain (the GPU box ships the prebuilt library) cannot rebuild
        if os.path.exists(lib):
            return lib
        raise RuntimeError("hipcc not found and " + lib + " is missing")
    os.makedirs(os.path.dirname(lib), exist_ok=True)
    cmd = [hipcc] + FLAGS + list(extra) + ["-I" + os.path.join(REPO, "include"), "-I" + SRC]
    cmd += [os.path.join(SRC, s) for s in SOURCES] + ["-o", lib]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    with open(lib + ".digest", "w") as f:
        f.write(source_digest(extra) + "\n")
    return lib


def build(force=False, verbose=False):
    """hipcc --offload-arch=gfx950 ... -> graphik_amd/lib/libgraphik_amd.so.  Rebuilds whenever
    any file under csrc/ or include/ (or the flags) differs from what the library was built from."""
    if not force and not _stale(LIB):
        return LIB
    return _compile(LIB, (), verbose)


def build_dev(force=False, verbose=False):
    extra = ("-DGIK_DEV",)
    if not force and not _stale(DEV_LIB, extra):
        return DEV_LIB
    return _compile(DEV_LIB, extra, verbose)


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
    if "--dev" in sys.argv:
        print(build_dev(force="--force" in sys.argv, verbose=True))
