"""The host engine of the exact policies (csrc/evs_hostcache.hip) under AddressSanitizer + UBSan on the CPU: the .hip file is
compiled host-only with hipcc's clang together with a random stress driver (tools/hostcache_asan.cpp) and run.  GPU
sanitizers are not available on the pool; this is the part of the product that runs on the host."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = "/opt/rocm/bin/hipcc"


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_host_engine_is_clean_under_asan_and_ubsan(tmp_path):
    exe = str(tmp_path / "hostcache_asan")
    csrc = os.path.join(ROOT, "ev-store-dlrm_amd", "csrc")
    cmd = [HIPCC, "--cuda-host-only", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
           "-fno-omit-frame-pointer", "-I" + os.path.join(ROOT, "include"), os.path.join(csrc, "evs_hostcache.hip"),
           os.path.join(csrc, "evs_api.hip"), os.path.join(ROOT, "tools", "hostcache_asan.cpp"), "-o", exe]
    b = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert b.returncode == 0, b.stderr[-2000:]
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and "sanitizer stress ok" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_file_tier_reader_pool_is_clean_under_tsan(tmp_path):
    """csrc/evs_filetier.hip host-only under ThreadSanitizer: two caller threads share one tier and fetch through the
    persistent reader pool (tools/filetier_tsan.cpp).  (It found the unguarded lazy creation of the pool.)"""
    exe = str(tmp_path / "filetier_tsan")
    csrc = os.path.join(ROOT, "ev-store-dlrm_amd", "csrc")
    cmd = [HIPCC, "--cuda-host-only", "-O1", "-g", "-std=c++17", "-fsanitize=thread", "-fno-omit-frame-pointer",
           "-I" + os.path.join(ROOT, "include"), os.path.join(csrc, "evs_filetier.hip"), os.path.join(csrc, "evs_api.hip"),
           os.path.join(ROOT, "tools", "filetier_tsan.cpp"), "-o", exe, "-lpthread"]
    b = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert b.returncode == 0, b.stderr[-2000:]
    d = tmp_path / "tabs"
    d.mkdir()
    for threads in ("2", "6"):
        env = dict(os.environ, EVS_FILETIER_THREADS=threads, TSAN_OPTIONS="halt_on_error=0")
        r = subprocess.run([exe, str(d)], capture_output=True, text=True, timeout=600, env=env)
        assert r.returncode == 0 and "sanitizer stress ok" in r.stdout and "ThreadSanitizer" not in r.stderr, (r.stdout[-800:], r.stderr[-3000:])
