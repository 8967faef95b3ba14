"""CPU sanitizer leg (SURVEY.md section 5 "Race detection"): the host-compilable native pieces --
the C oracle (oracle/c/gs_oracle.c, both arithmetic widths) and the host build of the device math
(easy_gaussian_splatting_amd/csrc/gs_math.h through tests/hostmath) -- built with -fsanitize=address,undefined
-fno-sanitize-recover=all and driven over their edge cases.  CPU build only: sanitizers are never run on the GPU pool."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SAN_ENV = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0:halt_on_error=1", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")


def _run(binary):
    p = subprocess.run([binary], capture_output=True, text=True, timeout=600, env=SAN_ENV)
    assert p.returncode == 0, f"{binary} failed under the sanitizers:\n{p.stdout[-2000:]}\n{p.stderr[-6000:]}"
    assert "runtime error" not in p.stderr and "AddressSanitizer" not in p.stderr and "LeakSanitizer" not in p.stderr, p.stderr[-6000:]
    assert "sanitizer leg ok" in p.stdout
    return p.stdout


@pytest.mark.parametrize("width", ["f32", "f64"])
def test_c_oracle_under_asan_ubsan(width):
    d = os.path.join(ROOT, "oracle", "c")
    subprocess.run(["make", "-C", d, f"san_driver_{width}"], check=True, capture_output=True)
    out = _run(os.path.join(d, f"san_driver_{width}"))
    assert out.count("checksum") == 6


def test_device_math_host_build_under_asan_ubsan():
    d = os.path.join(ROOT, "tests", "hostmath")
    exe = os.path.join(d, "san_driver")
    subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-fno-omit-frame-pointer", "-fsanitize=address,undefined",
                    "-fno-sanitize-recover=all", "-o", exe, os.path.join(d, "san_driver.cpp")], check=True, capture_output=True)
    out = _run(exe)
    assert out.count("checksum") == 5
