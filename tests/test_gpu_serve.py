"""GPU: the resident dispatcher of the fused launch (include/evstore_hip.h: evs_emb_interact_serve_*; dlrm_ops.InteractServer) gives the
bits of the launched kernel -- run in a child process under a hard time-out (a resident kernel that did not leave would otherwise
hold the suite)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("publish", ["host", "leader"])
def test_resident_dispatcher_gives_the_bits_of_the_launched_kernel(publish):
    """publish = "host": the default where the host can address device memory (large BAR) -- the descriptors are written
    through the aperture into the lines the blocks poll; "leader": block 0 reads a mailbox in host memory and republishes
    (EVS_SERVE_PUBLISH=leader; what a part without a large BAR runs)."""
    env = dict(os.environ)
    if publish == "leader":
        env["EVS_SERVE_PUBLISH"] = "leader"
    else:
        env.pop("EVS_SERVE_PUBLISH", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_serve_child.py")], capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert r.returncode == 0 and "SERVE_OK" in r.stdout, (r.stdout[-2000:], r.stderr[-3000:])
