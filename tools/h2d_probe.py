"""Developer probe: what one pinned -> device copy of a 16 384-batch's inputs costs, by tensor shape / dtype / stream."""
import time, torch
dev = torch.device("cuda")
n = 7667712
def t(fn, iters=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(iters): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / iters * 1e3
src8 = torch.empty(n, dtype=torch.uint8).pin_memory(); dst8 = torch.empty(n, dtype=torch.uint8, device=dev)
src64 = torch.empty(n // 8, dtype=torch.int64).pin_memory(); dst64 = torch.empty(n // 8, dtype=torch.int64, device=dev)
print("uint8 copy_ default stream: %.3f ms" % t(lambda: dst8.copy_(src8, non_blocking=True)))
print("int64 copy_ default stream: %.3f ms" % t(lambda: dst64.copy_(src64, non_blocking=True)))
print("int64 view of the uint8 block: %.3f ms" % t(lambda: dst8.view(torch.int64).copy_(src8.view(torch.int64), non_blocking=True)))
cs = torch.cuda.Stream()
def side():
    with torch.cuda.stream(cs):
        dst64.copy_(src64, non_blocking=True)
print("int64 copy_ side stream: %.3f ms" % t(side))
hp = torch.empty(n // 8, dtype=torch.int64)
print("pageable int64 copy_: %.3f ms" % t(lambda: dst64.copy_(hp)))
