"""9 x 29-bit signed-limb field (csrc/field9.cuh) and its XYZZ arithmetic (csrc/curve9.cuh), host build, against the 8 x 32 field:
products / squares / fused pairs with lazy and negative operands, both Montgomery-form conversions, to_words over its whole input
range, mixed and general additions with doubling, cancellation and infinity.  The device build differs only in `mac` (one
v_mad_i64_i32 instead of the int64 expression); the GPU parity tests cover that through the MSM kernels."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not found")
def test_field9_and_curve9_match_the_32_bit_field(tmp_path):
    exe = tmp_path / "field9_check"
    subprocess.run([HIPCC, "-O1", "-std=c++17", "-x", "hip", "--cuda-host-only", "-I", os.path.join(ROOT, "zk-symmetric-crypto_amd", "csrc"),
                    "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "host", "field9_check.cpp"), "-o", str(exe)],
                   check=True, capture_output=True, timeout=600)
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stdout + r.stderr


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not found")
def test_glv_scalar_multiplication_matches_double_and_add(tmp_path):
    """csrc/glv.cuh (the proof tail's s * Ar and r * Bs1 through the BN254 endomorphism: 127 doublings instead of 254), host build,
    against curve.cuh's plain double-and-add: edge scalars (0, 1, r - 1, r, 2^256 - 1, lambda, 2^128), 2^254- and 2^256-range random
    ones, the point at infinity; lambda * G == (beta x, y)."""
    exe = tmp_path / "glv_check"
    subprocess.run([HIPCC, "-O1", "-std=c++17", "-x", "hip", "--cuda-host-only", "-I", os.path.join(ROOT, "zk-symmetric-crypto_amd", "csrc"),
                    "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "host", "glv_check.cpp"), "-o", str(exe)],
                   check=True, capture_output=True, timeout=600)
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stdout + r.stderr
