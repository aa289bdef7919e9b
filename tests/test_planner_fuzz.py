"""ASan + UBSan build of the host planner (pure C++, no HIP) driven by a descriptor fuzzer: random junction
trees, half of them damaged (cycles, CSR offsets, unknown variables, owners, sizes ...).  jtp_build_plan must
plan or refuse with a message - never crash or touch memory it does not own (sanitizers on
the host library; the GPU pool has no GPU AddressSanitizer)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_planner_under_asan_ubsan_with_fuzzed_descriptors(tmp_path):
    exe = str(tmp_path / "fuzz_plan")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
                           os.path.join(ROOT, "tests", "fuzz", "fuzz_plan.cpp"),
                           os.path.join(ROOT, "junction-tree_amd", "csrc", "jtp_plan.cpp"), "-o", exe])
    for seed in (12345, 7):
        out = subprocess.run([exe, "2500", str(seed)], capture_output=True, text=True, timeout=600,
                             env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1"))
        assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
        assert "planned" in out.stdout and "rejected with a message" in out.stdout
