"""The library's HOST code under AddressSanitizer (no GPU needed): every .hip translation unit compiled host-only with
-fsanitize=address, linked against tools/asan/fake_hip.cpp (host-memory stand-in for the HIP runtime; kernels are not
executed) and driven through the C ABI by tools/asan/host_asan_driver.cpp -- context life cycle, both DoF orders, meshes
of the latency and of the bandwidth regime, every trajectory sweep with growing and shrinking batch / step counts, the
info calls with matching and mismatching sizes, error paths.  Written for the round-2 SIGSEGV audit."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(not (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")), reason="hipcc not available")
def test_host_code_is_clean_under_address_sanitizer():
    d = os.path.join(ROOT, "tools", "asan")
    b = subprocess.run(["make", "-C", d, "-j8"], capture_output=True, text=True, timeout=900)
    assert b.returncode == 0, b.stdout[-2000:] + b.stderr[-2000:]
    r = subprocess.run([os.path.join(d, "_build", "host_asan_driver")], capture_output=True, text=True, timeout=900,
                       env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1:halt_on_error=1"))
    out = r.stdout + r.stderr
    assert r.returncode == 0, out[-3000:]
    assert "host_asan_driver: 0 unexpected return codes" in out
    assert "AddressSanitizer" not in out and "LeakSanitizer" not in out and "bad launch geometry" not in out, out[-3000:]
