"""Does the NUMA node a pinned batch was allocated on decide its host-to-device rate?  The packed one-copy loop with the process
pinned to each node's cores in turn (the pinned blocks are allocated -- first-touched -- by the pinned thread).
python tools/numa_h2d_probe.py"""
import glob, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch


def cpus_of(node):
    s = open("/sys/devices/system/node/node%d/cpulist" % node).read().strip()
    out = []
    for part in s.split(","):
        a, _, b = part.partition("-")
        out += list(range(int(a), int(b or a) + 1))
    return out


torch.cuda.set_device(0)
p = torch.cuda.get_device_properties(0)
bus = "%04x:%02x:%02x.0" % (getattr(p, "pci_domain_id", 0), getattr(p, "pci_bus_id", 0), getattr(p, "pci_device_id", 0))
try:
    print("GPU 0 at", bus, "numa_node", open("/sys/bus/pci/devices/%s/numa_node" % bus).read().strip())
except Exception as e:
    print("GPU 0 at", bus, "numa_node unknown:", e)
nodes = sorted(int(d.rsplit("node", 1)[1]) for d in glob.glob("/sys/devices/system/node/node[0-9]*"))
all_cpus = sorted(os.sched_getaffinity(0))
dev_buf = torch.empty(8 << 20, dtype=torch.uint8, device="cuda")
for node in nodes + nodes:
    cp = [c for c in cpus_of(node) if c in all_cpus]
    if not cp:
        continue
    os.sched_setaffinity(0, cp)
    blocks = [torch.empty(8 << 20, dtype=torch.uint8).pin_memory() for _ in range(8)]
    for b in blocks:
        b.fill_(1)
    for i in range(40):
        dev_buf.copy_(blocks[i % 8], non_blocking=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(200):
        dev_buf.copy_(blocks[i % 8], non_blocking=True)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("node %d (%d cpus): %.3f ms per 8 MiB copy = %.1f GB/s" % (node, len(cp), dt / 200 * 1e3, (8 << 20) * 200 / dt / 1e9), flush=True)
    del blocks
os.sched_setaffinity(0, all_cpus)
