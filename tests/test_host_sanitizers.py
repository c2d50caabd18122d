"""CPU sanitizer build of the C-ABI layer's host-only code paths: configuration
validation, constant folding (live and Mars models), the tile layout, the staging plan of the state
exchange -- compiled with g++ -fsanitize=address,undefined and run without a GPU."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("g++") is None or not os.path.exists("/opt/rocm/include/hip/hip_runtime_api.h"),
                    reason="needs g++ and the HIP runtime headers")
def test_host_logic_under_asan_and_ubsan(tmp_path):
    exe = str(tmp_path / "host_logic_san")
    build = ["g++", "-std=c++17", "-x", "c++", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
             "-I", os.path.join(ROOT, "include"), "-I", "/opt/rocm/include",
             os.path.join(ROOT, "tests", "host", "host_logic_san.cpp"), "-L/opt/rocm/lib", "-lamdhip64", "-ldl",
             "-Wl,-rpath,/opt/rocm/lib", "-o", exe]
    p = subprocess.run(build, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0 and "host_logic_san: OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
