"""Builds libmsnet_hip.so (gfx950) in-tree with hipcc.  No torch, no cmake: the library is a plain
C-ABI shared object (include/msnet_hip.h) that is loaded with ctypes.

    python -m ms-nets_amd.build            # or: from __graft_entry__ import build; build()
"""
import fcntl
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libmsnet_hip.so")
ARCH = "gfx950"

# (source, extra flags).  volume.hip reproduces float32 operation order => no FMA contraction there.
SOURCES = [
    ("api.cpp", []),
    ("pack.hip", []),
    ("peaks.hip", []),
    ("conv3d.hip", []),
    ("conv3d_f16s.hip", []),
    ("conv3d_f16s_ws_s2.hip", []),
    ("conv3d_f16s_ws_c16.hip", []),
    ("conv3d_f16s_ws_co64.hip", []),
    ("conv3d_f16s_ws_co32.hip", []),
    ("conv3d_f16s_c8.hip", []),
    ("conv3d_f16s_deconv.hip", []),
    ("conv3d_f16s_direct.hip", []),
    ("conv3d_f16s_wd.hip", []),
    ("tail.hip", []),
    ("metrics.hip", []),
    ("volume.hip", ["-ffp-contract=off"]),
    # -fno-slp-vectorize: hipcc otherwise packs the ZSAD add chains into v_pk_add_f32 + v_and (no |x| modifier) + v_mov
    ("volume_fused.hip", ["-ffp-contract=off", "-fno-slp-vectorize"]),
]
COMMON = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=" + ARCH, "-Wall", "-Wno-unused-function",
          "-fno-gpu-rdc"]


def _hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    raise RuntimeError("hipcc not found")


def _digest(paths, extra=()):
    """sha256 over the CONTENT of the given files (+ the command line): what an object file was compiled from."""
    h = hashlib.sha256()
    for p in paths:
        with open(p, "rb") as f:
            h.update(hashlib.sha256(f.read()).digest())
    for e in extra:
        h.update(str(e).encode() + b"\0")
    return h.hexdigest()


_compiler_id = {}


def _compiler_identity(hipcc):
    """`hipcc --version` (HIP / clang version, install dir): a ROCm upgrade must not reuse the old compiler's objects."""
    if hipcc not in _compiler_id:
        try:
            _compiler_id[hipcc] = subprocess.run([hipcc, "--version"], capture_output=True, text=True, timeout=120).stdout.strip()
        except (OSError, subprocess.SubprocessError):
            _compiler_id[hipcc] = "unknown"
    return _compiler_id[hipcc]


def _write_stamp(stamp, digest):
    tmp = "%s.%d.tmp" % (stamp, os.getpid())
    with open(tmp, "w") as f:
        f.write(digest + "\n")
    os.replace(tmp, stamp)                              # atomic: a concurrent reader sees the old stamp or the new one


def _stale(target, stamp, digest):
    """An object is reused only if it exists AND was compiled from exactly these bytes with exactly this command (its .sha
    stamp): modification times say nothing on a box that received the tree by copy."""
    if not (os.path.exists(target) and os.path.exists(stamp)):
        return True
    with open(stamp) as f:
        return f.read().strip() != digest


def build(force=False, verbose=True, defines=(), lib=None):
    """defines/lib: build an experiment variant (-D flags) into another .so without touching the shipped one.
    Returns the library path; `build.last_compiled` lists the sources this call actually compiled."""
    hipcc = _hipcc()
    tag = "" if not defines else "_" + "_".join(d.replace("=", "") for d in defines)
    objdir = os.path.join(HERE, "build" + tag)
    out = lib or LIB                                   # (a variant's path must not stick to later default builds)
    os.makedirs(objdir, exist_ok=True)
    # several ranks of one launch may call build() at once: one builds, the others wait and then find fresh stamps
    with open(os.path.join(objdir, ".lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            return _build_locked(hipcc, objdir, out, force, verbose, defines)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def _build_locked(hipcc, objdir, out, force, verbose, defines):
    headers = [os.path.join(CSRC, h) for h in ("common.h", "conv_common.h", "conv_f16s.h", "conv_f16s_ws.h")] + [
        os.path.join(HERE, "..", "include", "msnet_hip.h")]
    objs, procs, stamps = [], [], {}
    for src, extra in SOURCES:
        sp = os.path.join(CSRC, src)
        op = os.path.join(objdir, src + ".o")
        objs.append(op)
        cmd = [hipcc, "-x", "hip"] + COMMON + extra + ["-D" + d for d in defines] + ["-c", sp, "-o", op]
        # flags + the source's NAME (not its absolute path: a tree copied elsewhere reuses its objects) + the compiler's identity
        flags = ["-x", "hip"] + COMMON + extra + ["-D" + d for d in defines]
        digest = _digest([sp] + headers, flags + [src, _compiler_identity(hipcc)])
        if force or _stale(op, op + ".sha", digest):
            try:
                os.remove(op + ".sha")
            except FileNotFoundError:
                pass
            if verbose:
                print("[build]", " ".join(cmd), flush=True)
            procs.append((src, subprocess.Popen(cmd)))
            stamps[src] = (op + ".sha", digest)
    failed = [s for s, p in procs if p.wait() != 0]
    if failed:
        raise RuntimeError("hipcc failed for: " + ", ".join(failed))
    for src, (stamp, digest) in stamps.items():
        _write_stamp(stamp, digest)
    link_digest = _digest(objs, [ARCH])
    if force or procs or _stale(out, out + ".sha", link_digest):
        cmd = [hipcc, "-shared", "-fPIC", "--offload-arch=" + ARCH, "-o", out] + objs
        if verbose:
            print("[build]", " ".join(cmd), flush=True)
        subprocess.check_call(cmd)
        _write_stamp(out + ".sha", link_digest)
    build.last_compiled = [s for s, _ in procs]
    if verbose:
        print("[build] %s: compiled %d of %d sources (%s), objects keyed by content sha256" % (
            os.path.basename(out), len(procs), len(SOURCES), "forced" if force else "stale or missing"), flush=True)
    return out


build.last_compiled = []


if __name__ == "__main__":
    defs = [a[2:] for a in sys.argv[1:] if a.startswith("-D")]
    out = [a[6:] for a in sys.argv[1:] if a.startswith("--lib=")]
    build(force="--force" in sys.argv, defines=defs, lib=out[0] if out else None)
