"""ThreadSanitizer build of the staging ring's host side (csrc/staging_ring.hip: reader thread pool, slot state machine,
tickets), run on the CPU: host-memory ring, several files in flight, chunked reads, slots freed and re-submitted at
once, the short-read error path, teardown with reads pending (tests/native/ring_tsan.cpp).  GPU sanitizers are not
available on the pool; the device half of the ring (pinned slot reuse behind the H2D copy) is covered by
tests/test_reader_gpu.py::test_slot_reuse_with_large_samples."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLANG = "/opt/rocm/lib/llvm/bin/clang++"


@pytest.mark.skipif(not os.path.isfile(CLANG) or not os.path.isfile("/opt/rocm/lib/libamdhip64.so"),
                    reason="needs the ROCm clang and libamdhip64 to build the host half")
def test_ring_host_threads_under_tsan(tmp_path):
    exe = str(tmp_path / "ring_tsan")
    src = [os.path.join(ROOT, "bias-gan_amd", "csrc", f) for f in ("staging_ring.hip", "api.hip")]
    cmd = [CLANG, "-x", "hip", "--offload-host-only", "-std=c++17", "-O1", "-g", "-fsanitize=thread", "-I/opt/rocm/include",
           *src, os.path.join(ROOT, "tests", "native", "ring_tsan.cpp"), "-L/opt/rocm/lib", "-lamdhip64", "-lpthread",
           "-Wl,-rpath,/opt/rocm/lib", "-o", exe]
    b = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert b.returncode == 0, b.stderr[-2000:]
    work = tmp_path / "files"
    work.mkdir()
    r = subprocess.run([exe, str(work)], capture_output=True, text=True, timeout=300,
                       env=dict(os.environ, TSAN_OPTIONS="halt_on_error=0 exitcode=66"))
    shutil.rmtree(work, ignore_errors=True)
    assert "ThreadSanitizer" not in r.stderr and "ThreadSanitizer" not in r.stdout, r.stderr[-4000:]
    assert r.returncode == 0 and "ring_tsan: ok" in r.stdout, (r.returncode, r.stdout[-500:], r.stderr[-2000:])
