"""AddressSanitizer + UndefinedBehaviorSanitizer run of everything that executes on the CPU: the oracle and the host sources of
the product library (scene builder, YAML reader, flattener, wide-tree collapse, host refit, C ABI glue with the image and
checkpoint writers), linked into one driver by `make -C oracle asan`.  The driver feeds them every shipped config, malformed
YAML, truncated checkpoints, empty and invalid trees; any sanitizer report or failed expectation fails the test.  The
reference leans on Rust ownership for the same guarantees and panics on bad input (src/main.rs:510-513)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DRIVER = os.path.join(ROOT, "tests", "asan", "host_asan_driver")


def test_host_code_is_clean_under_asan_and_ubsan(tmp_path):
    if shutil.which("g++") is None or shutil.which("make") is None:
        pytest.skip("no host toolchain here (the GPU box runs the prebuilt libraries only)")
    subprocess.run(["make", "-s", "asan"], check=True, cwd=os.path.join(ROOT, "oracle"))
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0:halt_on_error=1", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    r = subprocess.run([DRIVER, os.path.join(ROOT, "configs"), str(tmp_path)], capture_output=True, text=True, env=env, timeout=600)
    report = r.stdout[-2000:] + r.stderr[-6000:]
    assert r.returncode == 0, report
    assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error:" not in r.stderr and "LeakSanitizer" not in r.stderr, report
    assert "host_asan_driver: ok" in r.stdout
