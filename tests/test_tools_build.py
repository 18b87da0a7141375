"""The developer probes under tools/ (HIP programs of their own, not part of the library) still cross-compile for gfx950:
tools/sector_probe.hip (random line requests per second) and tools/coexec_probe.hip (fp32 MFMA vs VALU on one SIMD) are the
evidence behind docs/HISTORY.md 3.2c *Round 3*, tools/store_pattern_probe.hip (store patterns, random reads + a store stream) behind
DESIGN.md 3.1 "what bounds"; a probe that no longer builds cannot be re-run on another part."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = "/opt/rocm/bin/hipcc"


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
@pytest.mark.parametrize("name", ["sector_probe", "coexec_probe", "store_pattern_probe", "atomic_probe"])
def test_probe_cross_compiles(tmp_path, name):
    out = tmp_path / (name + ".o")
    r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-c", os.path.join(ROOT, "tools", name + ".hip"), "-o", str(out)],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert out.stat().st_size > 0
