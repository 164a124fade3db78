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


def test_device_code_on_the_host_under_asan_ubsan():
    """Round 6: the CPU tier of the DEVICE code (tests/host/*.cpp compile the blob extraction, the voting item, the tail
    geometry and mpe_ddmath.h for the host) once more under AddressSanitizer + UndefinedBehaviorSanitizer
    (tools/host_sanitize.sh; GPU sanitizers are not available on this pool): the blob extraction's and the voting item's
    tests — no report, every test passes."""
    out = subprocess.run([os.path.join(ROOT, "tools", "host_sanitize.sh"), "-k",
                          "column_run or band_scans or raw_frame_blur or strict or fast_item or glibc"],
                         capture_output=True, text=True)
    assert out.returncode == 0, (out.stdout[-2000:], out.stderr[-2000:])
    assert "runtime error" not in out.stdout and "AddressSanitizer" not in out.stdout, out.stdout[-3000:]
    assert out.stdout.count("passed") >= 2, out.stdout[-2000:]
