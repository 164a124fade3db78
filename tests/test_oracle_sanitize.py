"""The oracle's whole path (detection, voting, validation, refinement, tracking state machine) under
AddressSanitizer + UBSan.  CPU only."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_oracle_under_asan_ubsan():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "sanitize_check"])
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    out = subprocess.run([os.path.join(ROOT, "oracle", "sanitize_check")], capture_output=True, text=True, env=env)
    assert out.returncode == 0, (out.returncode, out.stdout[-500:], out.stderr[-3000:])
    assert "sanitize_check ok" in out.stdout
