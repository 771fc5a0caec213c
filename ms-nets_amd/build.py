"""Builds libmsnet_hip.so (gfx950) in-tree with hipcc.  No torch, no cmake: the library is a plain
C-ABI shared object (include/msnet_hip.h) that is loaded with ctypes.

    python -m ms-nets_amd.build            # or: from __graft_entry__ import build; build()
"""
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


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True, defines=(), lib=None):
    """defines/lib: build an experiment variant (-D flags) into another .so without touching the shipped one."""
    hipcc = _hipcc()
    tag = "" if not defines else "_" + "_".join(d.replace("=", "") for d in defines)
    objdir = os.path.join(HERE, "build" + tag)
    out = lib or LIB                                   # (a variant's path must not stick to later default builds)
    os.makedirs(objdir, exist_ok=True)
    headers = [os.path.join(CSRC, h) for h in ("common.h", "conv_common.h", "conv_f16s.h", "conv_f16s_ws.h")] + [
        os.path.join(HERE, "..", "include", "msnet_hip.h"), os.path.abspath(__file__)]
    objs = []
    procs = []
    for src, extra in SOURCES:
        sp = os.path.join(CSRC, src)
        op = os.path.join(objdir, src + ".o")
        objs.append(op)
        if force or _stale(op, [sp] + headers):
            cmd = [hipcc, "-x", "hip"] + COMMON + extra + ["-D" + d for d in defines] + ["-c", sp, "-o", op]
            if verbose:
                print("[build]", " ".join(cmd), flush=True)
            procs.append((src, subprocess.Popen(cmd)))
    failed = [s for s, p in procs if p.wait() != 0]
    if failed:
        raise RuntimeError("hipcc failed for: " + ", ".join(failed))
    if force or procs or _stale(out, objs):
        cmd = [hipcc, "-shared", "-fPIC", "--offload-arch=" + ARCH, "-o", out] + objs
        if verbose:
            print("[build]", " ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return out


if __name__ == "__main__":
    defs = [a[2:] for a in sys.argv[1:] if a.startswith("-D")]
    out = [a[6:] for a in sys.argv[1:] if a.startswith("--lib=")]
    build(force="--force" in sys.argv, defines=defs, lib=out[0] if out else None)
