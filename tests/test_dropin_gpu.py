"""GPU: runs the drop-in test binary (oracle/dropin_test.cpp built by oracle/build_ref.sh): the reference's own AVX2 / sw
objects and the "hip" adapter objects (srsran_project_23.5_amd/adapters) receive the same stimuli through the reference's
C++ interfaces; results must be identical (bit-exact integer paths, float paths within 5e-6 / 1e-4)."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "oracle", "_ref", "dropin_test")


@pytest.mark.skipif(not os.path.exists(EXE), reason="oracle/_ref/dropin_test not built (needs /root/reference at build time)")
def test_reference_interfaces_with_hip_adapters():
    r = subprocess.run([EXE], capture_output=True, text=True, timeout=900)
    print(r.stdout[-3000:])
    print(r.stderr[-2000:])
    assert r.returncode == 0 and "DROPIN TEST PASSED" in r.stdout
