#!/usr/bin/env python3
"""Developer experiment: where does the fused launch spend its time?  Variant libraries (tools/variants.sh) with one
piece removed each, Kaggle tables and 3-row tables (every row an L1/L2 hit), steady-state batch."""
import os
import subprocess
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT)
    import torch
    import bench
    import evstore_dlrm_amd as E
    from tools.kbench import timeit
    for name, ln in (("kaggle", bench.KAGGLE_LN), ("all 3 rows", [3] * 26)):
        ev = bench.make_tables(ln, 36)
        for B in (16384, 65536):
            batches = bench.make_batches(ln, B, 8, 1, "cuda", "uniform")
            x = torch.rand(B, 36, device="cuda")
            us = timeit(lambda i: E.apply_emb_interact(x, batches[i % 8][0], batches[i % 8][1], ev, one_index_per_bag=True), 300)
            print("  %-12s B=%6d %7.1f us" % (name, B, us), flush=True)
    sys.exit(0)
for v in sys.argv[1:]:
    print("== " + v, flush=True)
    env = dict(os.environ, EVS_LIB_PATH=os.path.join(ROOT, "ev-store-dlrm_amd", "lib", "var", "libevstore_hip_%s.so" % v), EVS_FUSED_RF=os.environ.get("EVS_ATTRIB_RF", "0"))
    subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=env, timeout=600)
