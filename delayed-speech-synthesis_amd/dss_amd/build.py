"""Build libdss_hip.so (HIP kernels + C ABI) in-tree with hipcc for gfx950.

hipcc cross-compiles without a GPU, so this also runs on the CPU-only build host.  The .so is written
next to the sources' package (delayed-speech-synthesis_amd/libdss_hip.so) so it travels with the tree.

Every source is compiled to its own object (build/obj/, kept between builds; a source is recompiled when it, a
header or the flag set changed), up to DSS_BUILD_JOBS compilers at a time, and the objects are linked into the
library: one changed kernel file costs one compile, not nine.
"""
from __future__ import annotations

import concurrent.futures
import hashlib
import os
import shutil
import subprocess

PKG_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(PKG_ROOT, "csrc")
LIB_PATH = os.path.join(PKG_ROOT, "libdss_hip.so")
OBJ_DIR = os.path.join(PKG_ROOT, "build", "obj")
SOURCES = ["dss_capi.cpp", "dss_async.cpp", "hga_kernels.hip", "lpcnet_frame.hip", "lpcnet_sample.hip", "lpcnet_sample_pair.hip",
           "lpcnet_sample_generic.hip", "speech_gate.hip", "vad_lstm.hip", "bilstm_decoder.hip"]
HEADERS = ["dss_common.h", "dss_host.h", "lpcnet_device.h", "lpcnet_sample_common.h", "../../include/dss_hip.h",
           "../../include/dss_lpcnet_blob.h"]
# -ffp-contract=off: the path's parity contract is "same products, same sums, same order" as the scalar C
# reference; a fused multiply-add anywhere would change results.
# -fno-slp-vectorize: the hot loops are written with explicit 2-wide products where packing pays; automatic
# packing of the scalar sum chains only adds register shuffles.
CFLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize", "-fPIC",
          "-x", "hip", "-Wno-unused-result", "-Wno-unused-value"]
LDFLAGS = ["--offload-arch=gfx950", "-fPIC", "-shared"]


def _hipcc() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: libdss_hip.so cannot be built (there is no CPU fallback)")


def needs_build() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    return any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in SOURCES + HEADERS)


def _obj_path(src: str, flags) -> str:
    tag = hashlib.sha1(" ".join(flags).encode()).hexdigest()[:10]
    return os.path.join(OBJ_DIR, f"{os.path.splitext(src)[0]}.{tag}.o")


def _stale(src: str, obj: str) -> bool:
    if not os.path.exists(obj):
        return True
    t = os.path.getmtime(obj)
    return any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in [src] + HEADERS)


def build_library(force: bool = False, verbose: bool = False, extra_flags=(), out: str | None = None) -> str:
    """Compile what changed and link.  force=True recompiles every source (what __graft_entry__.build() does)."""
    out = out or LIB_PATH
    if not force and out == LIB_PATH and not extra_flags and not needs_build():
        return LIB_PATH
    hipcc = _hipcc()
    flags = [*CFLAGS, *extra_flags]
    os.makedirs(OBJ_DIR, exist_ok=True)
    jobs = []
    for src in SOURCES:
        obj = _obj_path(src, flags)
        if force or _stale(src, obj):
            jobs.append((src, obj))

    def compile_one(job):
        src, obj = job
        cmd = [hipcc, *flags, "-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd, cwd=CSRC)

    n = max(1, int(os.environ.get("DSS_BUILD_JOBS", str(min(6, os.cpu_count() or 1)))))
    with concurrent.futures.ThreadPoolExecutor(max_workers=n) as ex:
        list(ex.map(compile_one, jobs))
    cmd = [hipcc, *LDFLAGS, "-o", out, *[_obj_path(s, flags) for s in SOURCES]]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd, cwd=CSRC)
    return out


if __name__ == "__main__":
    import sys
    print(build_library(force="--force" in sys.argv, verbose=True))
