#!/usr/bin/env python3
"""Differential fuzz of the cache tier on the GPU against the oracle (checker only):
  exact path   EvLFU (three variants, approx mode) / LRU / LFU, random tables, capacities, streams and chunk sizes:
               hit flags, rows, final list order and counters
  two tiers    request_c1c2 against oracle.C1C2: tier codes, rows, both tiers' final lists
  batched3     lookup_batch_c1c2c3 (alt-key tier as a key set): tier codes, rows, C3 membership / flags / size
  batched      lookup_batch with random capacities / batch sizes (incl. caches smaller than one batch): snapshot hit
               flags, exact rows, no duplicate keys, size <= capacity, histogram consistent
usage: python tools/fuzz_cache.py [seconds] [seed]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import evstore_dlrm_amd as E  # noqa: E402
from evstore_dlrm_amd import gpu_cache  # noqa: E402
from oracle import oracle as orc  # noqa: E402


def _stream(rs, n_rows, n_req):
    T = len(n_rows)
    kind = rs.choice(["zipf", "uniform", "hot"])
    if kind == "zipf":
        a = rs.uniform(1.05, 1.6)
        r = np.stack([np.minimum(rs.zipf(a, size=n_req) - 1, n_rows[k] - 1) for k in range(T)], axis=1)
    elif kind == "uniform":
        r = np.stack([rs.randint(0, n_rows[k], size=n_req) for k in range(T)], axis=1)
    else:
        r = np.stack([rs.randint(0, min(n_rows[k], 4), size=n_req) for k in range(T)], axis=1)
    return r.astype(np.int32)


def exact_case(rs, case):
    T = int(rs.choice([1, 2, 5, 13, 26, 26, 26, 40]))
    codec = int(rs.choice([32, 32, 32, 16, 8, 4]))          # (round 6: the one-tier reduced-precision builds too -- rows of 18 .. 256 bytes)
    d = int(rs.choice([4, 16, 36, 36, 64] if codec == 4 else [4, 5, 9, 16, 36, 36, 64]))   # (5, 9: rows that are no multiple of 16 bytes)
    n_rows = [int(rs.choice([1, 3, 20, 200, 3000])) for _ in range(T)]
    policy = rs.choice(["evlfu", "evlfu", "evlfu", "lru", "lfu"])
    variant = rs.choice(["python", "cpp", "cython"]) if policy == "evlfu" else "python"
    cap = int(rs.choice([1, 2, 7, T, 3 * T, 64, 500, 5000]))
    n_req = int(rs.choice([1, 10, 200, 1200]))
    approx = int(rs.choice([-1, -1, -1, max(1, T // 2), T])) if policy == "evlfu" and variant == "python" else -1
    tag = "exact case %d: %s/%s T=%d d=%d codec=%d cap=%d n_req=%d approx=%d rows=%s" % (case, policy, variant, T, d, codec, cap, n_req, approx, n_rows[:6])
    ws = [rs.uniform(-1, 1, size=(n, d)).astype(np.float32) for n in n_rows]
    if codec == 32:
        raws, tabs = ws, ws
    else:
        raws = [orc.encode_table(w, codec) for w in ws]
        tabs = [orc.decode(r, codec, d) for r in raws]
    reqs = _stream(rs, n_rows, n_req)

    def oracle():
        if policy == "evlfu":
            return orc.EvLFU(cap, tabs, d, variant)
        return orc.LRU(cap, tabs, d) if policy == "lru" else orc.LFU(cap, tabs, d)
    o = oracle()
    if approx > 0:
        return tag + " (skipped: approx mode draws random vectors in the reference)"
    want_h, want_o = [], []
    try:
        for rq in reqs:
            h, out = o.request(rq) if policy != "evlfu" else o.request(rq, approx)
            want_h.append(h.copy()); want_o.append(out.copy())
    except RuntimeError:
        return tag + " (skipped: the reference raises on this stream)"
    want_h, want_o = np.stack(want_h), np.stack(want_o)
    backing = [torch.from_numpy(np.ascontiguousarray(t)).cuda() for t in raws]
    c = E.GpuCache(policy, cap, T, d, codec, variant)
    c.set_backing(backing)
    r = torch.from_numpy(reqs).cuda()
    chunk = int(rs.choice([1, 3, 64, 5000]))
    hits, outs = [], []
    for s in range(0, n_req, chunk):
        h, out = c.request(r[s:s + chunk].contiguous(), approx)
        hits.append(h.cpu().numpy().astype(bool)); outs.append(out.cpu().numpy())
    hits, outs = np.concatenate(hits), np.concatenate(outs)
    assert np.array_equal(hits, want_h), tag + ": hit flags"
    assert np.array_equal(outs.view(np.uint32), want_o.view(np.uint32)), tag + ": rows"

    def final_state(cc, what):
        got = cc.dump()
        if policy == "lru":
            got = got[:, 1:]   # the GPU dump carries a (constant) priority column for LRU
        assert np.array_equal(got, o.dump()), tag + ": final lists" + what
        if policy == "evlfu":
            st, so = cc.stats(), o.state()
            assert [st["min_c1"], st["n_perfect"], st["size"], st["n_flush"]] == [so["min_c1"], so["n_perfect"], so["size"], so["n_flush"]], tag + ": state" + what
    final_state(c, "")
    # round 6: the same stream through the RESIDENT SERVER of a second cache -- rows into a ring slot, into a buffer of the caller's
    # (evs_cache_serve_request_to), ids by value or by address, a launch-per-request call and an idle gap in between
    if T <= 28 and n_req <= 200:
        c2 = E.GpuCache(policy, cap, T, d, codec, variant)
        c2.set_backing(backing)
        c2.serve_start(approx, n_slots=int(rs.choice([1, 2, 5])), idle_us=int(rs.choice([30, 300])))
        for i, rq in enumerate(reqs):
            way = int(rs.randint(0, 5)) if T <= 26 else int(rs.choice([0, 3, 4]))
            if way == 0:
                h, rows = c2.serve_request(rq)
                rows = rows.clone()
            elif way == 1:
                rows = torch.full((T, d), -7.0, dtype=torch.float32, device="cuda")
                torch.cuda.synchronize()
                h = c2.serve_request_to(rq, rows)
            elif way == 2:
                ids = torch.from_numpy(np.stack([rq.astype(np.int64), np.full(T, -3, np.int64)], 1)).cuda()
                rows = torch.full((T, d), -7.0, dtype=torch.float32, device="cuda")
                torch.cuda.synchronize()
                h = c2.serve_request_to(ids, rows)
            elif way == 3:
                hh, rows = c2.request(r[i:i + 1].contiguous(), approx)      # (sends the server home first)
                h = hh[0].cpu().numpy()
                rows = rows[0]
            else:
                import time
                time.sleep(0.0005)
                h, rows = c2.serve_request(rq)
                rows = rows.clone()
            assert np.array_equal(np.asarray(h).astype(bool), want_h[i]), tag + ": server hit flags, request %d way %d" % (i, way)
            assert np.array_equal(rows.cpu().numpy().view(np.uint32), want_o[i].view(np.uint32)), tag + ": server rows, request %d way %d" % (i, way)
        c2.serve_stop()
        final_state(c2, " (server)")
    return tag


def c1c2_case(rs, case):
    T, d = 26, 36
    n = int(rs.choice([5, 60, 400]))
    ws = [rs.uniform(-1, 1, size=(n, d)).astype(np.float32) for _ in range(T)]
    raw8 = [orc.encode_table(w, 8) for w in ws]
    raw4 = [orc.encode_table(w, 4) for w in ws]
    dec8 = [orc.decode(r, 8, d) for r in raw8]
    dec4 = [orc.decode(r, 4, d) for r in raw4]
    cap1, cap2 = int(rs.choice([30, 100, 600])), int(rs.choice([30, 200, 1200]))
    thr = int(rs.choice([23, 23, 15, 26]))
    n_req = int(rs.choice([50, 600, 1500]))
    tag = "c1c2 case %d: n=%d cap1=%d cap2=%d thr=%d n_req=%d" % (case, n, cap1, cap2, thr, n_req)
    reqs = _stream(rs, [n] * T, n_req)
    o = orc.C1C2(cap1, cap2, dec8, dec4, d, thr)
    want_t, want_o = [], []
    for rq in reqs:
        t, out = o.request(rq)[:2]
        want_t.append(t.copy()); want_o.append(out.copy())
    c1 = E.GpuCache("evlfu", cap1, T, d, 8, "cpp")
    c2 = E.GpuCache("evlfu", cap2, T, d, 4, "cpp")
    c1.set_backing([torch.from_numpy(r).cuda() for r in raw8])
    c2.set_backing([torch.from_numpy(r).cuda() for r in raw4])
    r = torch.from_numpy(reqs).cuda()
    chunk = int(rs.choice([1, 17, 4000]))
    tiers, outs = [], []
    for s in range(0, n_req, chunk):
        t, out = gpu_cache.request_c1c2(c1, c2, r[s:s + chunk].contiguous(), threshold=thr)
        tiers.append(t.cpu().numpy()); outs.append(out.cpu().numpy())
    assert np.array_equal(np.concatenate(tiers), np.stack(want_t)), tag + ": tier codes"
    assert np.array_equal(np.concatenate(outs).view(np.uint32), np.stack(want_o).view(np.uint32)), tag + ": rows"
    assert np.array_equal(c1.dump(), o.c1.dump()) and np.array_equal(c2.dump(), o.c2.dump()), tag + ": final lists"
    return tag


def c1c2c3_case(rs, case):
    T, d = 26, 36
    n = int(rs.choice([8, 80, 400]))
    ws = [rs.uniform(-1, 1, size=(n, d)).astype(np.float32) for _ in range(T)]
    raw8 = [orc.encode_table(w, 8) for w in ws]
    raw4 = [orc.encode_table(w, 4) for w in ws]
    dec8 = [orc.decode(r, 8, d) for r in raw8]
    dec4 = [orc.decode(r, 4, d) for r in raw4]
    alt = [(rs.randint(0, n, size=n) * 100 + rs.randint(1, T + 1, size=n)).astype(np.uint32) for _ in range(T)]
    cap1, cap2, cap3 = int(rs.choice([30, 150, 600])), int(rs.choice([30, 300, 1200])), int(rs.choice([50, 51, 99, 120, 400]))
    thr = int(rs.choice([23, 23, 12]))
    n_req = int(rs.choice([60, 700, 1600]))
    tag = "c1c2c3 case %d: n=%d caps=%d/%d/%d thr=%d n_req=%d" % (case, n, cap1, cap2, cap3, thr, n_req)
    reqs = _stream(rs, [n] * T, n_req)
    o = orc.C1C2C3(cap1, cap2, cap3, dec8, dec4, alt, d, thr)
    want_t, want_o = [], []
    for rq in reqs:
        t, out, _ = o.request(rq)
        want_t.append(t.copy()); want_o.append(out.copy())
    c1 = E.GpuCache("evlfu", cap1, T, d, 8, "cpp")
    c2 = E.GpuCache("evlfu", cap2, T, d, 4, "cpp")
    c1.set_backing([torch.from_numpy(r).cuda() for r in raw8])
    c2.set_backing([torch.from_numpy(r).cuda() for r in raw4])
    c3 = gpu_cache.GpuAltKeyTier(cap3, [torch.from_numpy(a.view(np.int32)).cuda() for a in alt])
    r = torch.from_numpy(reqs).cuda()
    chunk = int(rs.choice([1, 29, 5000]))
    tiers, outs = [], []
    for s in range(0, n_req, chunk):
        t, out = gpu_cache.request_c1c2c3(c1, c2, c3, r[s:s + chunk].contiguous(), threshold=thr)
        tiers.append(t.cpu().numpy()); outs.append(out.cpu().numpy())
    assert np.array_equal(np.concatenate(tiers), np.stack(want_t)), tag + ": tier codes"
    assert np.array_equal(np.concatenate(outs).view(np.uint32), np.stack(want_o).view(np.uint32)), tag + ": rows"
    assert np.array_equal(c1.dump(), o.c1.dump()) and np.array_equal(c2.dump(), o.c2.dump()), tag + ": final lists"
    assert c3.stats() == o.c3_state(), tag + ": alt-key tier counters"
    return tag


def _sa_geom(cap, n_rows, cap2=None):
    """(nset, ways of C1, bits of the key universe) of the set-associative policy -- csrc/evs_cache.hip: sa_single_feasible /
    sa_pair_geometry restated (as tests/test_gpu_cache.py: _sa_geom): one tier cap // 8 sets of 8 ways; a pair that starts
    out together max(cap1 // 8, ceil(cap2 / 16)) shared set records unless a tier would get fewer than 4 ways"""
    total = sum(n_rows)
    bits = 1
    while (1 << bits) < total:
        bits += 1
    if cap2 is None:
        return cap // 8, 8, bits
    nset = max(cap // 8, (cap2 + 15) // 16, 1)
    w1, w2 = min(cap // nset, 16), min(cap2 // nset, 16)
    if w1 < 4 or w2 < 4:
        return cap // 8, 8, bits
    return nset, w1, bits


def _sa_sets(keys_tr, nset, n_rows, bits):
    """set of each (table_1based, row) key (csrc/evs_hash.h: sa_perm / sa_divmod): the dense row number through two rounds of
    odd multiply + xorshift on `bits` bits, modulo the number of sets"""
    base = np.concatenate([[0], np.cumsum(np.asarray(n_rows, np.uint64))]).astype(np.uint64)
    t = np.asarray([t for t, _ in keys_tr], np.int64) - 1
    x = base[t] + np.asarray([r for _, r in keys_tr], np.uint64)
    mask, half = np.uint64((1 << bits) - 1), np.uint64((bits + 1) // 2)
    x = (x * np.uint64(0x9E3779B1)) & mask
    x ^= x >> half
    x = (x * np.uint64(0x85EBCA6B)) & mask
    x ^= x >> half
    return (x % np.uint64(nset)).astype(np.int64)


def batched_case(rs, case):
    T = int(rs.choice([1, 7, 26, 26, 32]))
    d = int(rs.choice([16, 36, 36, 64]))
    n_rows = [int(rs.choice([2, 30, 500, 8000])) for _ in range(T)]
    cap = int(rs.choice([1, 5, 60, 700, 6000]))
    B = int(rs.choice([1, 9, 100, 700, 3000]))
    n_batches = int(rs.choice([2, 6, 15]))
    tag = "batched case %d: T=%d d=%d cap=%d B=%d batches=%d rows=%s" % (case, T, d, cap, B, n_batches, n_rows[:6])
    codec = int(rs.choice([32, 32, 16, 8, 4]))
    host = codec == 32 and rs.randint(0, 4) == 0     # the miss tier in pinned host memory
    tag += " codec=%d host=%s" % (codec, host)
    src = [rs.uniform(-1, 1, size=(n, d)).astype(np.float32) for n in n_rows]
    raws = [orc.encode_table(t, codec) for t in src]
    tabs = src if codec == 32 else [orc.decode(r, codec, d) for r in raws]      # what a lookup must return
    policy = str(rs.choice(["sampled", "setassoc", "setassoc", "plan"]))
    if policy == "setassoc" and (host or cap < 8):   # (the set-associative policy: tables in HBM, at least one full set)
        policy = "sampled"
    tag += " policy=%s" % policy
    c = E.GpuCache("evlfu", cap, T, d, codec, rs.choice(["python", "cpp"])).set_batch_policy(policy)
    c.set_backing([torch.from_numpy(r).pin_memory() if host else torch.from_numpy(r).cuda() for r in raws])
    reqs = _stream(rs, n_rows, B * n_batches)
    resident = {}
    hits_total = 0
    st_prev = None
    use_interact = T + 1 <= 32 and ((codec == 32 and d in (16, 36, 64)) or (codec != 32 and d in (16, 32, 36) and not host))
    for s in range(0, len(reqs), B):
        rq = reqs[s:s + B]
        rt = torch.from_numpy(rq).cuda()
        was_interact = False
        lost_keys = set()
        if use_interact and rs.randint(0, 2):
            was_interact = True
            x = torch.rand(len(rq), d, device="cuda")
            hit, R = c.lookup_interact(rt, x)
            rows = np.stack([tabs[k][rq[:, k]] for k in range(T)], axis=1)
            want = orc.interact_features(x.cpu().numpy(), [rows[:, k, :] for k in range(T)])
            np.testing.assert_allclose(R.cpu().numpy(), want, rtol=1e-5, atol=2e-6 * (d / 36.0), err_msg=tag)
        else:
            hit, out = c.lookup_batch(rt)
            out = out.cpu().numpy()
            for k in range(T):
                assert np.array_equal(out[:, k, :], tabs[k][rq[:, k]]), tag + ": rows of table %d" % k
        hit = hit.cpu().numpy().astype(bool)
        want_hit = np.array([[(k + 1, int(rq[b, k])) in resident for k in range(T)] for b in range(len(rq))])
        hits_total += int(hit.sum())
        ev_before = st_prev["n_evict"] if st_prev else 0
        dmp, st = c.batch_dump(), c.batch_stats()
        if was_interact and policy == "setassoc" and os.environ.get("EVS_CACHE_INLINE", "1") != "0" and os.environ.get("EVS_SA_DUAL", "1") != "0":
            # round 5: the update runs inside the probe + interaction launch -- a flag says "served from the cache": never for a key
            # that was not resident when the batch arrived; a resident key reported as a miss was retired by this batch's own inserts
            assert not (hit & ~want_hit).any(), tag + ": a hit flag for a key that was not resident"
            lost_keys = {(k + 1, int(rq[b, k])) for b in range(len(rq)) for k in range(T) if want_hit[b, k] and not hit[b, k]}
            assert len(lost_keys) <= st["n_evict"] - ev_before, tag + ": more resident keys reported as misses than the batch evicted"
        else:
            assert np.array_equal(hit, want_hit), tag + ": snapshot hit flags"
        st_prev = st
        keys = [(int(t), int(rw)) for _, t, rw in dmp]
        assert len(set(keys)) == len(keys) == st["size"] <= cap, tag + ": duplicates / size"
        assert np.array_equal(np.bincount(dmp[:, 0], minlength=T + 1)[:T + 1], np.array(st["hist"])), tag + ": histogram"
        new_res = {(int(t), int(rw)): int(p) for p, t, rw in dmp}
        for key, p in new_res.items():
            if key in resident and key not in lost_keys:   # (a key retired and inserted again by the same batch starts over at that request's agg_hit)
                assert p >= resident[key], tag + ": priority went down"
        resident = new_res
    st = c.batch_stats()
    assert st["n_hits"] == hits_total and st["n_requests"] == len(reqs), tag + ": counters"
    return tag


def batched2_case(rs, case):
    T, d = 26, 36
    n = int(rs.choice([20, 300]))
    ws = [rs.uniform(-1, 1, size=(n, d)).astype(np.float32) for _ in range(T)]
    raw8 = [orc.encode_table(w, 8) for w in ws]
    raw4 = [orc.encode_table(w, 4) for w in ws]
    dec8 = [orc.decode(r, 8, d) for r in raw8]
    dec4 = [orc.decode(r, 4, d) for r in raw4]
    cap1, cap2 = int(rs.choice([40, 300, 2000])), int(rs.choice([40, 600, 4000]))
    thr = int(rs.choice([23, 20, 26]))
    B = int(rs.choice([1, 30, 250, 900]))
    n_batches = int(rs.choice([3, 8, 14]))
    tag = "batched two-tier case %d: n=%d cap1=%d cap2=%d thr=%d B=%d batches=%d" % (case, n, cap1, cap2, thr, B, n_batches)
    policy = str(rs.choice(["sampled", "setassoc", "setassoc", "plan"]))
    tag += " policy=%s" % policy
    c1 = E.GpuCache("evlfu", cap1, T, d, 8, "cpp").set_batch_policy(policy)
    c2 = E.GpuCache("evlfu", cap2, T, d, 4, "cpp").set_batch_policy(policy)
    c1.set_backing([torch.from_numpy(r).cuda() for r in raw8])
    c2.set_backing([torch.from_numpy(r).cuda() for r in raw4])
    reqs = _stream(rs, [n] * T, B * n_batches)
    R1, R2 = {}, {}
    sa = policy == "setassoc"
    if sa:   # set of every key in C1 (csrc/evs_hash.h: sa_set_of)
        nset1, ways1, bits1 = _sa_geom(cap1, [n] * T, cap2)
        set_of = _sa_sets([(k + 1, v) for k in range(T) for v in range(n)], nset1, [n] * T, bits1).reshape(T, n)
    for s in range(0, len(reqs), B):
        rq = reqs[s:s + B]
        tier, out = gpu_cache.lookup_batch_c1c2(c1, c2, torch.from_numpy(rq).cuda(), threshold=thr)
        tier, out = tier.cpu().numpy(), out.cpu().numpy()
        c1_full = len(R1) >= cap1
        if sa:
            occ = np.bincount(_sa_sets(list(R1), nset1, [n] * T, bits1), minlength=nset1) if R1 else np.zeros(nset1, int)
        for b in range(len(rq)):
            in1 = np.array([(k + 1, int(rq[b, k])) in R1 for k in range(T)])
            in2 = np.array([(k + 1, int(rq[b, k])) in R2 for k in range(T)]) & ~in1
            assert np.array_equal(tier[b] == 1, in1) and np.array_equal(tier[b] == 2, in2), tag + ": tier flags"
            agg = int(in1.sum() + in2.sum())
            for k in range(T):
                row = int(rq[b, k])
                if in1[k]:
                    want = dec8[k][row]
                elif in2[k]:
                    want = dec4[k][row]
                else:
                    full = occ[set_of[k, row]] >= ways1 if sa else c1_full   # set-associative tiers: "C1 full" = the key's own C1 set
                    dest = 1 if not full else ((1 if k % 2 == 1 else 2) if agg < thr else 2)
                    want = dec8[k][row] if dest == 1 else dec4[k][row]
                assert np.array_equal(out[b, k].view(np.uint32), want.view(np.uint32)), tag + ": row (%d,%d)" % (b, k)
        d1, d2 = c1.batch_dump(), c2.batch_dump()
        n1 = {(int(t), int(rw)): int(p) for p, t, rw in d1}
        n2 = {(int(t), int(rw)): int(p) for p, t, rw in d2}
        assert len(n1) == len(d1) == c1.batch_stats()["size"] <= cap1 and len(n2) == len(d2) == c2.batch_stats()["size"] <= cap2, tag + ": sizes"
        assert not (set(n1) & set(n2)), tag + ": a key in both tiers"
        if not c1_full and not sa:
            assert len(n2) == len(R2), tag + ": C2 touched while C1 had room"
        R1, R2 = n1, n2
    return tag


def batched3_case(rs, case):
    """Batched three-tier lookup against the snapshot: tier codes (3 = key in C3 and its alt row resident), served rows
    (tables of -1 / 0 / 1: exact in every codec), C3 members are keys the tiers gave up, flags only on keys served
    through their alt key, size <= capacity, hit counter."""
    T = int(rs.choice([3, 26]))
    d = int(rs.choice([16, 36]))
    n = int(rs.choice([20, 300]))
    ca, cb = [(8, 4), (32, 8), (32, 4)][int(rs.randint(0, 3))]
    ws = [rs.randint(-1, 2, size=(n, d)).astype(np.float32) for _ in range(T)]
    raws = {c: [orc.encode_table(w, c) for w in ws] for c in (ca, cb)}
    spread = int(rs.choice([4, 16]))
    alt = [np.array([(r % spread) * 100 + ((t + 1) % T + 1) for r in range(n)], dtype=np.uint32) for t in range(T)]
    cap1, cap2, cap3 = int(rs.choice([40, 300])), int(rs.choice([40, 600])), int(rs.choice([50, 64, 800]))
    thr = int(rs.choice([23, 20]))
    B = int(rs.choice([1, 30, 250]))
    n_batches = int(rs.choice([4, 12]))
    tag = "batched three-tier case %d: T=%d d=%d n=%d codecs=%d/%d caps=%d/%d/%d thr=%d B=%d batches=%d" % (
        case, T, d, n, ca, cb, cap1, cap2, cap3, thr, B, n_batches)
    policy = str(rs.choice(["sampled", "setassoc", "setassoc", "plan"]))
    tag += " policy=%s" % policy
    c1 = E.GpuCache("evlfu", cap1, T, d, ca, "cpp").set_batch_policy(policy)
    c2 = E.GpuCache("evlfu", cap2, T, d, cb, "cpp").set_batch_policy(policy)
    c1.set_backing([torch.from_numpy(a).cuda() for a in raws[ca]])
    c2.set_backing([torch.from_numpy(a).cuda() for a in raws[cb]])
    c3 = E.GpuAltKeyTier(cap3, [torch.from_numpy(a.view(np.int32)).cuda() for a in alt])
    reqs = _stream(rs, [n] * T, B * n_batches)
    R1, R2, M3 = set(), set(), set()
    ever_alt, removed, n3 = set(), set(), 0
    # half of the cases go through the interaction consumer (rows decoded inside the kernel, nothing materialised): R is
    # checked against the rows the tier codes imply
    use_interact = bool(rs.randint(0, 2)) and T + 1 <= 28
    itself = bool(rs.randint(0, 2))
    tag += " interact=%s" % use_interact
    for s in range(0, len(reqs), B):
        rq = reqs[s:s + B]
        if use_interact:
            x_np = rs.uniform(-1, 1, size=(len(rq), d)).astype(np.float32)
            tier, Rg = gpu_cache.lookup_interact_c1c2c3(c1, c2, c3, torch.from_numpy(rq).cuda(), torch.from_numpy(x_np).cuda(),
                                                       threshold=thr, itself=itself)
            tier = tier.cpu().numpy()
            out = np.empty((len(rq), T, d), dtype=np.float32)
            for b in range(len(rq)):
                for k in range(T):
                    a = int(alt[k][rq[b, k]])
                    out[b, k] = ws[a % 100 - 1][a // 100] if tier[b, k] == 3 else ws[k][rq[b, k]]
            want_R = orc.interact_features(x_np, [out[:, k] for k in range(T)], itself)
            np.testing.assert_allclose(Rg.cpu().numpy(), want_R, rtol=1e-5, atol=2e-6, err_msg=tag + ": R")
        else:
            tier, out = gpu_cache.lookup_batch_c1c2c3(c1, c2, c3, torch.from_numpy(rq).cuda(), threshold=thr)
            tier, out = tier.cpu().numpy(), out.cpu().numpy()
        for b in range(len(rq)):
            for k in range(T):
                key = (k + 1, int(rq[b, k]))
                a = int(alt[k][rq[b, k]])
                akey = (a % 100, a // 100)
                want_t = 1 if key in R1 else 2 if key in R2 else 3 if (key in M3 and (akey in R1 or akey in R2)) else 0
                assert tier[b, k] == want_t, tag + ": tier code (%d,%d) %d != %d" % (b, k, tier[b, k], want_t)
                want = ws[akey[0] - 1][akey[1]] if want_t == 3 else ws[k][rq[b, k]]
                assert np.array_equal(out[b, k], want), tag + ": row (%d,%d)" % (b, k)
                if want_t == 3:
                    ever_alt.add(key)
                    n3 += 1
        n1 = {(int(t), int(rw)) for _, t, rw in c1.batch_dump()}
        n2 = {(int(t), int(rw)) for _, t, rw in c2.batch_dump()}
        assert len(n1) <= cap1 and len(n2) <= cap2 and not (n1 & n2), tag + ": tiers"
        removed |= (R1 - n1) | (R2 - n2)
        m3, st3 = c3.batch_dump()
        members = {(int(t), int(rw)) for t, rw, _ in m3}
        assert len(members) == len(m3) == st3["members"] <= st3["capacity"] <= cap3, tag + ": C3 size"
        assert members <= removed, tag + ": C3 member the tiers never gave up"
        assert {(int(t), int(rw)) for t, rw, f in m3 if f} <= ever_alt, tag + ": recency flag without an alt hit"
        assert st3["n_hit"] == n3, tag + ": alt hit counter"
        R1, R2, M3 = n1, n2, members
    return tag


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rs = np.random.RandomState(seed)
    t0 = time.time()
    n = [0, 0, 0, 0, 0, 0]
    last = ""
    while time.time() - t0 < seconds:
        which = int(rs.choice([0, 0, 1, 2, 2, 3, 4, 5]))
        if os.environ.get("EVS_FUZZ_VERBOSE"):
            print("-> case %d kind %d" % (sum(n), which), flush=True)
        last = (exact_case, c1c2_case, batched_case, batched2_case, c1c2c3_case, batched3_case)[which](rs, sum(n))
        n[which] += 1
    print("cache fuzz ok: %d exact, %d two-tier, %d batched, %d batched two-tier, %d three-tier, %d batched three-tier cases in %.0f s (seed %d); last %s" % (
        n[0], n[1], n[2], n[3], n[4], n[5], time.time() - t0, seed, last))


if __name__ == "__main__":
    main()
