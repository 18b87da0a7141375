#!/usr/bin/env python3
"""Developer probe: where do the kernel arguments live?  HIP_FORCE_DEV_KERNARG=1 (arguments written to DEVICE memory) is
this stack's default: unset, set from the shell, set in-process before or after `import torch` all read the same
(B = 2 048: 7.4 us per fused launch, B = 16 384: 19.9 us).  Setting it to 0 -- or to the EMPTY string, which parses as 0 --
puts the arguments in host memory and costs 3.5-3.9 us per launch (11.8 / 23.6 us): every block's first scalar loads of the
2 KB argument block then cross PCIe.  Nothing to gain; kept so that the check can be repeated on another stack.
usage: [HIP_FORCE_DEV_KERNARG=0|1] python tools/kernarg_probe.py {shell|before|after|never}"""
import os
import sys
mode = sys.argv[1] if len(sys.argv) > 1 else "never"
if mode == "before":
    os.environ["HIP_FORCE_DEV_KERNARG"] = "1"
import torch  # noqa: E402
if mode == "after":
    os.environ["HIP_FORCE_DEV_KERNARG"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import evstore_dlrm_amd as E  # noqa: E402

d, B = 36, int(os.environ.get("PROBE_B", "2048"))
ev = bench.make_tables(bench.KAGGLE_LN, d)
batches = bench.make_batches(bench.KAGGLE_LN, B, 16, 1, "cuda", "uniform")
x = torch.rand(B, d, device="cuda")
for rep in range(3):
    for _ in range(200):
        E.apply_emb_interact(x, batches[0][0], batches[0][1], ev)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(400):
        E.apply_emb_interact(x, batches[i % 16][0], batches[i % 16][1], ev)
    e1.record()
    torch.cuda.synchronize()
print("%s (HIP_FORCE_DEV_KERNARG=%s): B=%d %.2f us per launch" % (mode, os.environ.get("HIP_FORCE_DEV_KERNARG"), B, e0.elapsed_time(e1) / 400 * 1e3))
