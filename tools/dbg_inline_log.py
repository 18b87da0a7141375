#!/usr/bin/env python3
"""Developer tool: replay the event log of the update folded into the probe launch (a library built with -DEVS_X_LOG,
EVS_LIB_PATH=.../libevstore_hip_xlog.so) -- every successful raise / claim as (type, word address, old word, new word) --
and look for a word whose chain of (old -> new) events does not link up, or whose final value is not what the dump shows."""
import ctypes as C, os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, torch
import evstore_dlrm_amd as E
from evstore_dlrm_amd import _lib
from oracle import oracle as orc
from test_gpu_cache import _zipf_requests
L = _lib.lib()
L.evs_x_log_fetch.restype = C.c_longlong
L.evs_x_log_fetch.argtypes = [C.c_void_p, C.c_longlong, C.c_int]
n_rows = [3000, 40, 20000, 700, 5, 9000, 1500, 12, 26000, 300, 8000, 64, 2200, 17000, 3, 450, 5000, 90, 13000, 2, 7000, 30, 1000, 11000, 150, 4000]
tabs = orc.kaggle_tables(n_rows, 21)
dev = [torch.from_numpy(t).cuda() for t in tabs]
cap_frac, batch = 0.02, 160
cap = int(cap_frac * sum(n_rows))
buf = np.zeros((1 << 18, 4), np.uint64)
for seed in range(2, 8):
    reqs = _zipf_requests(n_rows, 4 * batch, 5 + seed)
    x = torch.rand(batch, 36, device="cuda")
    c = E.GpuCache("evlfu", cap, 26, 36, 32, "python").set_batch_policy("setassoc"); c.set_backing(dev)
    r = torch.from_numpy(reqs).cuda()
    act0 = np.zeros(27, int); h0 = np.zeros(27, int)
    for k in range(2):
        L.evs_x_log_fetch(None, 0, 1)
        c.lookup_interact(r[k * batch:(k + 1) * batch].contiguous(), x)
        n = L.evs_x_log_fetch(buf.ctypes.data, 1 << 18, 1)
        d = c.batch_dump(); st = c.batch_stats()
        act = np.bincount(d[:, 0], minlength=27); h = np.array(st["hist"])
        ev = buf[:n].copy()
        print("seed", seed, "batch", k, "events", n, "hist ok" if np.array_equal(act, h) else ("DIFF", np.nonzero(act - h)[0], (act - h)[np.nonzero(act - h)[0]]))
        dl = np.zeros(27, int)
        for t, a, o, nw in ev:
            if int(o): dl[int(o) >> 26] -= 1
            dl[int(nw) >> 26] += 1
        print("   events' net:", dl[np.nonzero(dl)[0]], "at", np.nonzero(dl)[0])
        print("   words' net :", (act - act0)[np.nonzero(act - act0)[0]], "at", np.nonzero(act - act0)[0])
        print("   stats' net :", (h - h0)[np.nonzero(h - h0)[0]], "at", np.nonzero(h - h0)[0])
        act0, h0 = act, h
        if np.array_equal(act, h):
            continue
        by = {}
        for t, a, o, nw in ev:
            by.setdefault(int(a), []).append((int(t), int(o), int(nw)))
        t2 = [(int(a), int(o), int(nw)) for t, a, o, nw in ev if int(t) == 2]
        print("   claims:", len(t2), "evictions:", sum(1 for _, o, _ in t2 if o), "stamps of new words:", sorted({(nw >> 10) & 0x7fff for _, _, nw in t2}), "sel of new words:", sorted({nw >> 25 & 1 for _, _, nw in t2}))
        for a, o, nw in [x for x in t2 if x[1]][:6]:
            print("     evict %x: %08x -> %08x" % (a, o, nw))
        hip = C.CDLL("libamdhip64.so")
        hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        for a, evs in by.items():
            news = [nw for _, _, nw in evs]; olds = [o for _, o, _ in evs]
            finals = [nw for nw in news if nw not in olds]
            cur = C.c_uint32(0)
            hip.hipMemcpy(C.byref(cur), C.c_void_p(a), 4, 2)
            if len(finals) != 1 or finals[0] != cur.value:
                print("  addr %x: memory %08x (prio %d), chain ends %s" % (a, cur.value, cur.value >> 26, [hex(f) for f in finals]))
                for t, o, nw in evs:
                    print("     type %d  %08x (prio %d) -> %08x (prio %d)" % (t, o, o >> 26, nw, nw >> 26))
                for t, a2, o, nw in ev:
                    if (int(nw) & 0x3ffffff) == (cur.value & 0x3ffffff) or (int(o) & 0x3ffffff) == (cur.value & 0x3ffffff):
                        print("     elsewhere: type %d addr %x  %08x -> %08x" % (int(t), int(a2), int(o), int(nw)))
        for a, evs in by.items():
            # can the events be ordered into a chain?
            olds = [o for _, o, _ in evs]; news = [nw for _, _, nw in evs]
            starts = [o for o in olds if o not in news]
            if len(starts) != 1 or len(set(olds)) != len(olds):
                print("  addr %x: %d events, chain starts %s" % (a, len(evs), [hex(s) for s in starts]))
                for t, o, nw in evs:
                    print("     type %d  %08x (prio %d) -> %08x (prio %d)" % (t, o, o >> 26, nw, nw >> 26))
                for t, a2, o, nw in ev:
                    if (int(nw) & 0x3ffffff) == (cur.value & 0x3ffffff) or (int(o) & 0x3ffffff) == (cur.value & 0x3ffffff):
                        print("     elsewhere: type %d addr %x  %08x -> %08x" % (int(t), int(a2), int(o), int(nw)))
        sys.exit(0)
