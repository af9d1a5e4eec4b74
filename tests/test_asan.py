"""Host-side sanitizer build of the C ABI (sanitizers run on the CPU build only - GPU AddressSanitizer is not available
on this pool).  `make asan` compiles every translation unit of libsarx with -fsanitize=address,undefined for the host pass
(-fno-gpu-sanitize: device code as shipped) into build/asan/ and links tests/asan/abi_asan_test.cpp against it; the program drives the
argument checking of every entry point of include/sarx.h, sarx_last_error, the no-device path, the exception guard, and - with host
stand-ins for the runtime behind the function-pointer table staged_copy uses - the staged transfer's chunking, its thread /
inline-share / join logic and its error path.  A sanitizer report aborts the program (non-zero exit)."""
import os
import re
import subprocess

from conftest import ROOT

CSRC = os.path.join(ROOT, "nis-sar-amtigmti-video_amd", "csrc")


def test_host_side_of_the_abi_under_address_and_ub_sanitizer():
    r = subprocess.run(["make", "-j8", "asan"], cwd=CSRC, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    exe = os.path.join(ROOT, "build", "asan", "abi_asan_test")
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0:halt_on_error=1", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, (r.stdout + r.stderr)[-4000:]
    assert "all checks passed" in r.stdout
    assert not re.search(r"ERROR: (Address|Leak)Sanitizer|runtime error:", r.stdout + r.stderr), (r.stdout + r.stderr)[-4000:]


def test_the_driver_calls_every_entry_point_of_the_header():
    """The argument-checking pass must not fall behind the header: every function include/sarx.h declares appears in the driver."""
    hdr = open(os.path.join(ROOT, "include", "sarx.h")).read()
    names = set(re.findall(r"\b(sarx_[a-z0-9_]+)\s*\(", hdr))
    names -= {"sarx_radar_params", "sarx_ati_outputs", "sarx_tdbp_params"}
    drv = open(os.path.join(ROOT, "tests", "asan", "abi_asan_test.cpp")).read()
    missing = sorted(n for n in names if not re.search(r"\b" + n + r"\s*\(", drv))
    assert not missing, missing
