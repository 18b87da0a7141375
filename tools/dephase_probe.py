"""Do the phases of co-resident blocks cost time?  The u8 fused launch at B = 65 536 as ONE launch, and as two / four
independent part-launches alternated over two streams (their blocks start out of phase with each other's).
python tools/dephase_probe.py [bits]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import evstore_dlrm_amd as E

bits = int(sys.argv[1]) if len(sys.argv) > 1 else 8
d, T, B = 36, 26, 65536
ev = bench.make_tables(bench.KAGGLE_LN, d, bits=bits, codes="encoded" if bits != 32 else "random")
batches = bench.make_batches(bench.KAGGLE_LN, B, 4, 1, "cuda", "uniform")
x = torch.rand(B, d, device="cuda")
out = torch.empty(B, d + 351, device="cuda")
streams = [torch.cuda.Stream(), torch.cuda.Stream()]


def run(parts, n=60):
    def once(i):
        off, idx = batches[i % 4]
        if parts == 1:
            E.apply_emb_interact(x, off, idx, ev, None, out=out, one_index_per_bag=True)
            return
        h = B // parts
        for p in range(parts):
            with torch.cuda.stream(streams[p % 2]):
                E.apply_emb_interact(x[p * h:(p + 1) * h], off[:, :h], idx[:, p * h:(p + 1) * h].contiguous() if False else idx[:, p * h:(p + 1) * h], ev, None,
                                     out=out[p * h:(p + 1) * h], one_index_per_bag=True)
    for i in range(10):
        once(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for s in streams:
        s.wait_event(e0)
    for i in range(n):
        once(i)
    for s in streams:
        torch.cuda.current_stream().wait_stream(s)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for parts in (1, 2, 4, 1, 2, 4):
    print("u%d B=%d as %d launch(es): %.1f us per batch" % (bits, B, parts, run(parts)), flush=True)
