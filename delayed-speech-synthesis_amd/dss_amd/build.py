"""Build libdss_hip.so (HIP kernels + C ABI) in-tree with hipcc for gfx950.

hipcc cross-compiles without a GPU, so this also runs on the CPU-only build host.  The .so is written
next to the sources' package (delayed-speech-synthesis_amd/libdss_hip.so) so it travels with the tree.
"""
from __future__ import annotations

import os
import shutil
import subprocess

PKG_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(PKG_ROOT, "csrc")
LIB_PATH = os.path.join(PKG_ROOT, "libdss_hip.so")
SOURCES = ["dss_capi.cpp", "hga_kernels.hip", "lpcnet_frame.hip", "lpcnet_sample.hip", "lpcnet_sample_pair.hip", "lpcnet_sample_generic.hip", "speech_gate.hip", "vad_lstm.hip", "bilstm_decoder.hip"]
HEADERS = ["dss_common.h", "lpcnet_device.h", "lpcnet_sample_common.h", "../../include/dss_hip.h", "../../include/dss_lpcnet_blob.h"]
# -ffp-contract=off: the path's parity contract is "same products, same sums, same order" as the scalar C
# reference; a fused multiply-add anywhere would change results.
# -fno-slp-vectorize: the hot loops are written with explicit 2-wide products where packing pays; automatic
# packing of the scalar sum chains only adds register shuffles.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize", "-fPIC", "-shared",
         "-x", "hip",
         "-Wno-unused-result", "-Wno-unused-value"]


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


def build_library(force: bool = False, verbose: bool = False, extra_flags=()) -> str:
    if not force and not needs_build():
        return LIB_PATH
    cmd = [_hipcc(), *FLAGS, *extra_flags, "-o", LIB_PATH, *[os.path.join(CSRC, s) for s in SOURCES]]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd, cwd=CSRC)
    return LIB_PATH


if __name__ == "__main__":
    print(build_library(force=True, verbose=True))
