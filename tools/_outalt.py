import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, evstore_dlrm_amd as E
from kbench import timeit
B, d = 16384, 36
ev = bench.make_tables(bench.KAGGLE_LN, d)
batches = bench.make_batches(bench.KAGGLE_LN, B, 8, 1, "cuda", "uniform")
xs = [torch.rand(B, d, device="cuda") for _ in range(2)]
P = 27 * 26 // 2
R = [torch.empty(B, d + P, device="cuda") for _ in range(4)]
for rep in range(2):
    print("out=None        %.2f us" % timeit(lambda i: E.apply_emb_interact(xs[0], batches[i % 8][0], batches[i % 8][1], ev, one_index_per_bag=True), 400))
    print("out=one buffer  %.2f us" % timeit(lambda i: E.apply_emb_interact(xs[0], batches[i % 8][0], batches[i % 8][1], ev, None, out=R[0], one_index_per_bag=True), 400))
    print("out=two buffers %.2f us" % timeit(lambda i: E.apply_emb_interact(xs[0], batches[i % 8][0], batches[i % 8][1], ev, None, out=R[i % 2], one_index_per_bag=True), 400))
    print("two buffers + two x %.2f us" % timeit(lambda i: E.apply_emb_interact(xs[i % 2], batches[i % 8][0], batches[i % 8][1], ev, None, out=R[i % 2], one_index_per_bag=True), 400))
    print("one batch, one buffer %.2f us" % timeit(lambda i: E.apply_emb_interact(xs[0], batches[0][0], batches[0][1], ev, None, out=R[0], one_index_per_bag=True), 400))
