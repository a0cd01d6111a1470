"""CPU: ThreadSanitizer build + run of the host entry points' producer / consumer skeleton (csrc/prep_pipeline.h) with a
stub device layer (tests/native/prep_pipeline_tsan.cpp) - SURVEY.md 5, "race detection".  GPU sanitizers are not
available on the pool; the threaded host logic is HIP-free by construction, so the CPU build is the real thing."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "native", "prep_pipeline_tsan.cpp")


@pytest.mark.parametrize("san", ["thread", "address,undefined"])
def test_prep_pipeline_under_sanitizers(tmp_path, san):
    gxx = shutil.which("g++")
    if not gxx:
        pytest.skip("no g++")
    exe = str(tmp_path / ("pp_" + san.replace(",", "_")))
    subprocess.check_call([gxx, "-std=c++17", "-O1", "-g", f"-fsanitize={san}", "-fno-omit-frame-pointer", "-pthread", "-o", exe, SRC])
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=1 second_deadlock_stack=1", ASAN_OPTIONS="detect_leaks=1")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "all cases passed" in r.stdout and "WARNING: ThreadSanitizer" not in r.stderr
