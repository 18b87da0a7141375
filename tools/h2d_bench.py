"""bench.py's h2d_inclusive section alone (the reference loop's per-batch H2D, the packed single copy, the copy-stream forms).
python tools/h2d_bench.py"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
ev = bench.make_tables(bench.KAGGLE_LN, 36, seed=0, device=dev)
r = bench.h2d_inclusive_section(ev, bench.KAGGLE_LN, 36, 16384, dev)
for k, v in r.items():
    if isinstance(v, dict):
        print(k, json.dumps({a: (round(b, 4) if isinstance(b, float) else b) for a, b in v.items() if a != "unit"}))
