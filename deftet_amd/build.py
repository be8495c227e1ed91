"""Builds libdeftet_hip.so in-tree with hipcc for gfx950 (no JIT cache, no torch headers).

    python -m deftet_amd.build [--force] [--out PATH] [--swap point_in_tet.hip=OTHER.hip] [--only a.hip,b.cpp] [-DNAME[=V] ...]

Every source is compiled to its own object (in parallel, cached in a private per-user directory by source mtime and flags) and
the objects are linked into the shared library, so touching one kernel file rebuilds one object.
"""
from __future__ import annotations

import concurrent.futures
import fcntl
import hashlib
import os
import stat
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
_OBJDIR = None
LIB = os.path.join(HERE, "libdeftet_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")

# -ffp-contract=off: the integer-valued outputs (tet index, argmin face, NN index) are
# decided by fp32 sign tests / comparisons that must follow the reference's operation
# order without fused multiply-adds ("numerics").
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
         "-fno-fast-math", "-Wall", "-Wno-unused-function"]


# Per-source flags.  -fno-slp-vectorize for point_in_tet.hip: left to itself the compiler packs adjacent scalar fp32 subtractions
# and products of the per-tet setups into v_pk_add_f32 / v_pk_mul_f32 and pays for it with register shuffles and live ranges —
# k_bary_bwd_hits needs 94 registers with it and 70 without (five -> six waves per SIMD, 56.0 -> 52.5 us at configs[2]),
# k_finalize 64 -> 53 (29.4 -> 28.0), the traversal 62.7 -> 61.5; the packed FMAs the traversal loop WANTS are written as vector
# types and stay.  reduce.hip is the opposite case (k_rowdot_fused 16.9 -> 20.7 us without the packing), so the flag is per file.
# DEFTET_BUILD_NOSLP=a.hip,b.hip adds files for an experiment.
# (surface_ops.hip: k_tri_query_coop 0.632 -> 0.616 ms, geometry step 2.49 -> 2.465; raster.hip: slower without, 2.415 -> 2.44 ms;
# tet_ops / check_sign / vertex_ops: no difference — tools/probes/r05_noslp.sh)
PER_FILE_FLAGS = {"point_in_tet.hip": ["-fno-slp-vectorize"], "surface_ops.hip": ["-fno-slp-vectorize"]}


def _file_flags(src):
    extra = list(PER_FILE_FLAGS.get(os.path.basename(src), []))
    if os.path.basename(src) in (os.environ.get("DEFTET_BUILD_NOSLP") or "").split(","):
        extra.append("-fno-slp-vectorize")
    return extra


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".cpp")))


def _headers():
    out = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hpp")]
    out.append(os.path.join(os.path.dirname(HERE), "include", "deftet_hip.h"))
    out.append(os.path.abspath(__file__))                   # the flags live here: a change of them rebuilds
    return out


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(s) > t for s in sources() + _headers())


def objdir():
    """Object cache, outside the tree (the tree is what travels to the GPU box): DEFTET_BUILD_CACHE, else a per-user directory
    under the temp dir.  The objects in it are linked into the library this process loads, so the directory must be ours and
    closed to everybody else: created 0700; an existing one is only used if it belongs to this user, is a real directory
    (not a link) and is not writable by group / others — otherwise a fresh private directory is made for this run."""
    global _OBJDIR
    if _OBJDIR:
        return _OBJDIR
    explicit = os.environ.get("DEFTET_BUILD_CACHE")
    path = explicit or os.path.join(tempfile.gettempdir(), "deftet_amd_build_%d" % os.getuid())
    why = None
    try:
        os.makedirs(path, mode=0o700, exist_ok=True)
        st = os.lstat(path)
        if not stat.S_ISDIR(st.st_mode):
            why = "it is not a real directory"
        elif st.st_uid != os.getuid():
            why = "it belongs to uid %d" % st.st_uid
        elif st.st_mode & (stat.S_IWGRP | stat.S_IWOTH):
            why = "it is writable by group / others (mode %o)" % stat.S_IMODE(st.st_mode)
    except OSError as e:
        why = "it cannot be created or read (%s)" % e
    if why is None:
        _OBJDIR = path
        return _OBJDIR
    if explicit:                                              # set on purpose: say so instead of silently recompiling every run
        raise RuntimeError("DEFTET_BUILD_CACHE=%s is not usable as the object cache: %s" % (path, why))
    import atexit
    import shutil
    _OBJDIR = tempfile.mkdtemp(prefix="deftet_amd_build_")
    atexit.register(shutil.rmtree, _OBJDIR, True)             # one-run cache: removed with the process (objects, link lock)
    print("deftet_amd.build: object cache %s rejected (%s); using a private directory for this run only (full recompile): %s"
          % (path, why, _OBJDIR), file=sys.stderr, flush=True)
    return _OBJDIR


def _obj_for(src, flags):
    key = hashlib.sha1((src + "\0" + " ".join(flags)).encode()).hexdigest()[:12]
    return os.path.join(objdir(), "%s.%s.o" % (os.path.basename(src), key))


def _compile(src, flags, force, verbose):
    flags = flags + _file_flags(src)
    obj = _obj_for(src, flags)
    newest = max(os.path.getmtime(p) for p in [src] + _headers())
    if not force and os.path.exists(obj) and os.path.getmtime(obj) >= newest:
        return obj
    tmp = "%s.%d.tmp" % (obj, os.getpid())                  # ranks that build at the same time never share a temporary
    cmd = [HIPCC] + flags + ["-I", CSRC, "-x", "hip", "-c", src, "-o", tmp]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    os.replace(tmp, obj)
    return obj


def build(force: bool = False, verbose: bool = False, out: str = LIB, extra_flags=(), swap=None, only=None) -> str:
    """swap: {basename in csrc: replacement source path}; only: basenames to include (probe builds only)."""
    if out == LIB and not extra_flags and not swap and not force and not needs_build():
        return LIB
    if not os.path.exists(HIPCC):
        raise RuntimeError("hipcc not found at %s — cannot build libdeftet_hip.so" % HIPCC)
    flags = FLAGS + list(extra_flags)
    srcs = [(swap or {}).get(os.path.basename(s), s) for s in sources() if only is None or os.path.basename(s) in only]
    with concurrent.futures.ThreadPoolExecutor(max_workers=min(len(srcs), os.cpu_count() or 1)) as ex:
        objs = list(ex.map(lambda s: _compile(s, flags, force, verbose), srcs))
    os.makedirs(os.path.dirname(os.path.abspath(out)), exist_ok=True)
    tmp = "%s.%d.tmp" % (out, os.getpid())
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", tmp]
    if verbose:
        print(" ".join(cmd), flush=True)
    with open(os.path.join(objdir(), ".link.lock"), "w") as lock:   # one linker at a time per cache (several ranks, one tree)
        fcntl.flock(lock, fcntl.LOCK_EX)
        subprocess.check_call(cmd)
        os.replace(tmp, out)
    return out


def main(argv):
    force, out, extra, swap, only = False, LIB, [], {}, None
    it = iter(argv)
    for a in it:
        if a == "--force":
            force = True
        elif a == "--out":
            out = next(it)
        elif a == "--only":
            only = set(next(it).split(","))
        elif a == "--swap":
            k, v = next(it).split("=", 1)
            swap[k] = os.path.abspath(v)
        else:
            extra.append(a)
    print(build(force=force, verbose=True, out=out, extra_flags=extra, swap=swap or None, only=only))


if __name__ == "__main__":
    main(sys.argv[1:])
