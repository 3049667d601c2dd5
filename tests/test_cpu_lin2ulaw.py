"""The device's short form of xiph's lin2ulaw() (csrc/lpcnet_device.h dss_lin2ulaw: division by a constant as a product
and two fused corrections, clamp as max/min, float rounding) against the reference form in oracle/lpcnet_oracle.c.

tools/verify/lin2ulaw_exhaustive.c holds the C restatement of the device sequence and visits fp32 bit patterns with a
stride; stride 1 (all 2^32 inputs, ~1 minute on 8 cores) is the proof and is how the change was accepted -- the CPU
suite runs every 251st pattern; the GPU parity tests cover the values the synthesis actually produces.  The speculation of the sample
kernels calls this function (lpcnet.c lpcnet_synthesize_tail_impl: lin2ulaw(pcm), lin2ulaw(pred))."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_short_form_matches_the_reference_form_on_a_strided_sweep(oracle, tmp_path):
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("no gcc")
    exe = str(tmp_path / "l2u")
    subprocess.check_call([gcc, "-O2", "-ffp-contract=off", "-fopenmp", "-o", exe,
                           os.path.join(ROOT, "tools", "verify", "lin2ulaw_exhaustive.c"),
                           "-L" + os.path.join(ROOT, "oracle"), "-loracle", "-lm"])
    env = dict(os.environ, LD_LIBRARY_PATH=os.path.join(ROOT, "oracle"), OMP_NUM_THREADS="4")
    out = subprocess.run([exe, "251"], env=env, capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "lin2ulaw mismatches: 0" in out.stdout and "division mismatches for 2^-100 < |x| < 2^100: 0" in out.stdout
