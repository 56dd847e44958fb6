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
SOURCES = ["gik_host.hip", "gik_k_wave3.hip", "gik_k_wave3_strict.hip", "gik_k_anch.hip", "gik_k_wave2.hip",
           "gik_k_block.hip", "gik_k_npt.hip", "gik_k_npt4.hip", "gik_k_quad.hip", "gik_k_prep.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC"]
LINK_FLAGS = ["--offload-arch=gfx950", "-shared", "-fPIC"]


def headers():
    return sorted(glob.glob(os.path.join(SRC, "*.h")) + glob.glob(os.path.join(REPO, "include", "*.h")))


def dependencies():
    """Every file the library is compiled from: all of csrc/ plus the public headers."""
    deps = sorted(glob.glob(os.path.join(SRC, "*.hip")) + headers())
    assert all(os.path.join(SRC, s) in deps for s in SOURCES)
    return deps


def source_digest(extra=()):
    """Content hash of the sources and the compile flags (mtimes lie after a checkout)."""
    h = hashlib.sha256(" ".join(FLAGS + LINK_FLAGS + list(extra)).encode())
    for d in dependencies():
        h.update(os.path.relpath(d, REPO).encode())
        h.update(open(d, "rb").read())
    return h.hexdigest()


def _unit_digest(src, extra):
    """... of ONE translation unit: its own text, every header, the flags."""
    h = hashlib.sha256(" ".join(FLAGS + list(extra)).encode())
    for d in [os.path.join(SRC, src)] + headers():
        h.update(os.path.relpath(d, REPO).encode())
        h.update(open(d, "rb").read())
    return h.hexdigest()


def _stale(lib, extra=()):
    stamp = lib + ".digest"
    if not (os.path.exists(lib) and os.path.exists(stamp)):
        return True
    return open(stamp).read().strip() != source_digest(extra)


def have_toolchain():
    return bool(shutil.which("hipcc")) or os.path.exists("/opt/rocm/bin/hipcc")


def _compile(lib, extra, verbose, force=False):
    """One object file per translation unit (kernel groups + the host side), compiled in parallel and cached by content
    digest under lib/obj/, then linked to a temporary file that os.replace() moves into place -- all under an exclusive
    file lock: several ranks (bench.py --gpus N, the gloo tests) may find the library stale at the same time; one of
    them builds, the others wait for the lock, see a current digest and return."""
    import fcntl
    from concurrent.futures import ThreadPoolExecutor
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        # a machine without the toolchain (the GPU box ships the prebuilt library) cannot rebuild
        if os.path.exists(lib):
            return lib
        raise RuntimeError("hipcc not found and " + lib + " is missing")
    os.makedirs(os.path.dirname(lib), exist_ok=True)
    objdir = os.path.join(os.path.dirname(lib), "obj" + ("_dev" if extra else ""))
    os.makedirs(objdir, exist_ok=True)
    with open(lib + ".lock", "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and not _stale(lib, extra):      # another process built it meanwhile
                return lib
            inc = ["-I" + os.path.join(REPO, "include"), "-I" + SRC]

            def unit(src):
                obj = os.path.join(objdir, src.replace(".hip", ".o"))
                dig = _unit_digest(src, extra)
                if (not force and os.path.exists(obj) and os.path.exists(obj + ".digest")
                        and open(obj + ".digest").read().strip() == dig):
                    return obj
                cmd = [hipcc] + FLAGS + list(extra) + inc + ["-c", os.path.join(SRC, src), "-o", obj]
                if verbose:
                    print(" ".join(cmd), flush=True)
                subprocess.check_call(cmd)
                with open(obj + ".digest", "w") as f:
                    f.write(dig + "\n")
                return obj

            with ThreadPoolExecutor(max_workers=max(1, min(len(SOURCES), os.cpu_count() or 1))) as pool:
                objs = list(pool.map(unit, SOURCES))
            tmp = f"{lib}.tmp.{os.getpid()}"
            cmd = [hipcc] + LINK_FLAGS + objs + ["-o", tmp]
            if verbose:
                print(" ".join(cmd), flush=True)
            try:
                subprocess.check_call(cmd)
                os.replace(tmp, lib)
            finally:
                if os.path.exists(tmp):
                    os.remove(tmp)
            with open(lib + ".digest.tmp", "w") as f:
                f.write(source_digest(extra) + "\n")
            os.replace(lib + ".digest.tmp", lib + ".digest")
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return lib


def build(force=False, verbose=False):
    """hipcc --offload-arch=gfx950 ... -> graphik_amd/lib/libgraphik_amd.so.  Rebuilds whenever
    any file under csrc/ or include/ (or the flags) differs from what the library was built from."""
    if not force and not _stale(LIB):
        return LIB
    return _compile(LIB, (), verbose, force)


def build_dev(force=False, verbose=False):
    extra = ("-DGIK_DEV", "-DGIK_NPT_PROF")
    if not force and not _stale(DEV_LIB, extra):
        return DEV_LIB
    return _compile(DEV_LIB, extra, verbose, force)


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
    if "--dev" in sys.argv:
        print(build_dev(force="--force" in sys.argv, verbose=True))
