"""Host launch code of libltxmi under AddressSanitizer (CPU only; the device code is not instrumented -- the GPU pool
has no sanitizer support).  `make asan` builds libltxmi_asan.so + tests/native/abi_host_asan.cpp; the driver walks
every entry point's argument checks and launcher arithmetic with NO device visible, so nothing is launched."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "ltx-video-gpupoor_amd", "csrc")


def test_host_launch_code_under_asan():
    b = subprocess.run(["make", "-j4", "asan"], cwd=CSRC, capture_output=True, text=True, timeout=1500)
    assert b.returncode == 0, b.stdout[-2000:] + b.stderr[-4000:]
    env = dict(os.environ, HIP_VISIBLE_DEVICES="-1", ROCR_VISIBLE_DEVICES="-1", ASAN_OPTIONS="detect_leaks=0:abort_on_error=0")
    r = subprocess.run([os.path.join(CSRC, "asan", "abi_host_asan")], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-6000:]
    assert "0 unexpected" in r.stdout
    assert "AddressSanitizer" not in r.stderr, r.stderr[-6000:]
