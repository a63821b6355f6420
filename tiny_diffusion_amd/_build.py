"""In-tree build of libtdx.so (hipcc, gfx950 only).  No torch dependency: the
library is a plain C-ABI shared object (include/tdx.h)."""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libtdx.so")
OBJDIR = os.path.join(HERE, "build")

HIPCC_FLAGS = [
    "--offload-arch=gfx950",
    "-O3",
    "-std=c++17",
    "-fPIC",
    "-fno-gpu-rdc",
    "-mcode-object-version=5",
    # no implicit fma contraction: elementwise kernels reproduce the reference's
    # separately-rounded mul/add sequences bit-for-bit; FMAs are written explicitly
    "-ffp-contract=off",
    "-Wno-unused-result",
]


def _hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: libtdx.so cannot be built")


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _deps_mtime():
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hdrs.append(os.path.join(os.path.dirname(HERE), "include", "tdx.h"))
    return max(os.path.getmtime(h) for h in hdrs)


def _flags_stamp():
    return " ".join(HIPCC_FLAGS)


def _stamp_matches():
    try:
        return open(os.path.join(OBJDIR, "flags.txt")).read() == _flags_stamp()
    except OSError:
        return False


STAMP = LIB + ".src_sha256"


def source_hash() -> str:
    """SHA-256 over every HIP source / header and the compiler flags: written next to the
    library by build(); a differing stamp means the .so was built from other sources (file
    times are not comparable across machines, contents are)."""
    import hashlib

    h = hashlib.sha256(_flags_stamp().encode())
    files = sources() + sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h"))
    files.append(os.path.join(os.path.dirname(HERE), "include", "tdx.h"))
    for f in files:
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def stamp_mismatch() -> bool:
    """True only when a stamp exists and disagrees with the current sources."""
    try:
        return open(STAMP).read().strip() != source_hash()
    except OSError:
        return False


def is_stale():
    if not os.path.exists(LIB):
        return True
    if os.path.isdir(OBJDIR) and not _stamp_matches():
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(s) > t for s in sources()) or _deps_mtime() > t


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile every csrc/*.hip for gfx950 and link libtdx.so next to this file."""
    if not force and not is_stale() and not stamp_mismatch():
        return LIB
    os.makedirs(OBJDIR, exist_ok=True)
    # one builder at a time: under torch.distributed.run every rank imports the package at once
    import fcntl

    with open(os.path.join(OBJDIR, ".lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        if not force and not is_stale() and not stamp_mismatch():
            return LIB  # another process finished the build while this one waited
        return _build_locked(force, verbose)


def _build_locked(force: bool, verbose: bool) -> str:
    hipcc = _hipcc()
    hdr_t = _deps_mtime()
    if not _stamp_matches():
        force = True  # objects were built with other flags

    def compile_one(src):
        obj = os.path.join(OBJDIR, os.path.basename(src) + ".o")
        if (not force and os.path.exists(obj) and os.path.getmtime(obj) > os.path.getmtime(src)
                and os.path.getmtime(obj) > hdr_t):
            return obj
        cmd = [hipcc, *HIPCC_FLAGS, "-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{r.stderr}")
        if verbose and r.stderr:
            print(r.stderr, file=sys.stderr)
        return obj

    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
        objs = list(ex.map(compile_one, sources()))
    tmp = LIB + ".tmp"
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", tmp, *objs]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stderr}")
    os.replace(tmp, LIB)
    with open(os.path.join(OBJDIR, "flags.txt"), "w") as f:
        f.write(_flags_stamp())
    with open(STAMP, "w") as f:
        f.write(source_hash())
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
