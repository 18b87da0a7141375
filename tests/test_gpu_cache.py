"""GPU parity of the cache tier (csrc/evs_cache.hip) against the golden traces recorded from the
reference's Python policies and against the oracle (bit-exact: hit flags, returned rows, final
list order, counters)."""
import numpy as np
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def E():
    import evstore_dlrm_amd as E
    assert torch.cuda.is_available()
    E._lib.lib()
    return E


@pytest.fixture(scope="module")
def orc():
    from oracle import oracle
    return oracle


def _tables(orc, t):
    return orc.kaggle_tables([int(n) for n in t["n_rows"]], int(t["table_seed"]))


def _unpack(packed, n):
    return np.unpackbits(packed, axis=1)[:, :26].astype(bool)[:n]


def _run(E, policy, cap, tabs, reqs, chunk, approx=-1, variant="python", codec=32, raws=None):
    c = E.GpuCache(policy, cap, 26, 36, codec, variant)
    dev = [torch.from_numpy(np.ascontiguousarray(t)).cuda() for t in (raws if raws is not None else tabs)]
    c.set_backing(dev)
    hits, outs = [], []
    r = torch.from_numpy(np.ascontiguousarray(reqs, dtype=np.int32)).cuda()
    for s in range(0, len(reqs), chunk):
        h, o = c.request(r[s:s + chunk].contiguous(), approx)
        hits.append(h.cpu().numpy().astype(bool))
        outs.append(o.cpu().numpy())
    return c, np.concatenate(hits), np.concatenate(outs)


def test_evlfu_trace_through_pinned_host_buffers(E, orc):
    """The reference's loop is one request at a time with ids and rows on the host: the same golden trace
    through PINNED host tensors (read / written by the kernel itself, no copies) gives the same bits."""
    t = load_golden("cache_traces")
    tabs = _tables(orc, t)
    reqs = t["requests"][:400]
    c = E.GpuCache("evlfu", 300, 26, 36, 32, "python")
    c.set_backing([torch.from_numpy(np.ascontiguousarray(x)).cuda() for x in tabs])
    rows = torch.empty((1, 26), dtype=torch.int32).pin_memory()
    out = torch.empty((1, 26, 36), dtype=torch.float32).pin_memory()
    hit = torch.empty((1, 26), dtype=torch.uint8).pin_memory()
    want = _unpack(t["evlfu_cap300_hits"], len(t["requests"]))[:400]
    for i, rq in enumerate(reqs):
        rows[0] = torch.from_numpy(rq.astype(np.int32))
        c.request(rows, out=out, hit=hit)
        torch.cuda.synchronize()
        assert np.array_equal(hit[0].numpy().astype(bool), want[i]), i
        for k in range(26):
            assert np.array_equal(out[0, k].numpy(), tabs[k][rq[k]])
    with pytest.raises(ValueError):   # pageable host memory is refused, not silently copied
        c.request(torch.zeros((1, 26), dtype=torch.int32))


@pytest.mark.parametrize("cap,chunk", [(64, 1), (300, 7), (768, 1500), (2000, 64), (79, 1), (80, 33), (82, 1200)])
def test_evlfu_trace_matches_reference(E, orc, cap, chunk):
    t = load_golden("cache_traces")
    tabs = _tables(orc, t)
    reqs = t["requests_flush"] if cap in (79, 80, 82) else t["requests"]
    c, hits, outs = _run(E, "evlfu", cap, tabs, reqs, chunk)
    want = _unpack(t["evlfu_cap%d_hits" % cap], len(reqs))
    assert np.array_equal(hits, want)
    for k in range(26):  # rows are the table rows, bit for bit, hit or miss
        assert np.array_equal(outs[:, k, :], tabs[k][reqs[:, k]])
    np.testing.assert_array_equal(c.dump(), t["evlfu_cap%d_final_buckets" % cap])
    st = c.stats()
    assert [st["min_c1"], st["n_perfect"], st["size"], st["n_flush"]] == list(t["evlfu_cap%d_state" % cap])
    assert st["n_requests"] == len(reqs) and st["n_hits"] == int(want.sum())
    assert st["n_perfect_hits"] == int(want.all(1).sum())


@pytest.mark.parametrize("policy,cap,key", [("evlfu", 64, "evlfu_cap64"), ("evlfu", 768, "evlfu_cap768"), ("evlfu", 80, "evlfu_cap80"),
                                            ("lru", 64, "lru_cap64"), ("lfu", 768, "lfu_cap768")])
def test_resident_server_gives_the_golden_traces(E, orc, policy, cap, key):
    """Round 5: the exact policy as a RESIDENT SERVER (evs_cache_serve_*: a one-wavefront kernel that stays on the device and
    takes one request at a time from a mailbox in pinned host memory; rows into a ring in HBM, hit flags back through a host
    line) against the traces of the imported reference (cache_algo/EvLFU_C1.py, LRU.py, LFU.py): hit flags, rows and final
    list order bit-exact -- through stops in the middle (stats / dump send the server home, the next request starts it
    again), an idle time-out, and a plain launch-per-request call mixed in."""
    import time
    t = load_golden("cache_traces")
    tabs = _tables(orc, t)
    reqs = t["requests_flush"] if cap == 80 else t["requests"]
    want = _unpack(t[key + "_hits"], len(reqs))
    c = E.GpuCache(policy, cap, 26, 36, 32, "python")
    c.set_backing([torch.from_numpy(x).cuda() for x in tabs])
    c.serve_start(n_slots=3, idle_us=300)
    pin_rows = torch.empty((1, 26), dtype=torch.int32).pin_memory()
    pin_out = torch.empty((1, 26, 36), dtype=torch.float32).pin_memory()
    pin_hit = torch.empty((1, 26), dtype=torch.uint8).pin_memory()
    held = []
    for i, rq in enumerate(reqs):
        if i == 400:                       # anything else on the exact state sends the server home first ...
            st = c.stats()
            assert st["n_requests"] == 400 and st["n_hits"] == int(want[:400].sum())
        if i == 700:
            time.sleep(0.01)               # ... it leaves by itself when idle ...
        if i == 900:                       # ... and a launch-per-request call on the same cache fits in between
            pin_rows[0] = torch.from_numpy(rq.astype(np.int32))
            c.request(pin_rows, out=pin_out, hit=pin_hit)
            torch.cuda.synchronize()
            assert np.array_equal(pin_hit[0].numpy().astype(bool), want[i]), i
            continue
        hit, rows = c.serve_request(rq)
        assert np.array_equal(hit.astype(bool), want[i]), i
        if i % 97 == 0 or i < 8:           # rows: device tensors, the table rows bit for bit
            got = rows.cpu().numpy()
            for k in range(26):
                assert np.array_equal(got[k], tabs[k][rq[k]]), (i, k)
        held.append((rows, rq))
        if len(held) == 2:                 # a slot stays valid for n_slots - 1 more requests
            r0, q0 = held.pop(0)
            assert np.array_equal(r0[3].cpu().numpy(), tabs[3][q0[3]])
    c.serve_stop()
    if policy == "evlfu":
        np.testing.assert_array_equal(c.dump(), t[key + "_final_buckets"])
    elif policy == "lru":
        np.testing.assert_array_equal(c.dump()[:, 1:], t[key + "_final_order"])
    else:
        np.testing.assert_array_equal(c.dump(), t[key + "_final_freq"])
    st = c.stats()
    assert st["n_requests"] == len(reqs) and st["n_hits"] == int(want.sum())


@pytest.mark.parametrize("policy,cap,key", [("evlfu", 768, "evlfu_cap768"), ("evlfu", 80, "evlfu_cap80"), ("lru", 64, "lru_cap64")])
def test_resident_server_rows_into_the_callers_buffer(E, orc, policy, cap, key):
    """Round 6: evs_cache_serve_request_to -- the server writes a request's rows into a device buffer of the caller's instead of
    a ring slot (what the plug-in loop wants: a fresh tensor per request, no copy out of the ring), ids by value or by address
    (an int64 device tensor, element 0 of each row).  Mixed with ring requests on one cache over the imported reference's
    traces: hit flags and rows bit-exact whichever way a request is posted."""
    t = load_golden("cache_traces")
    tabs = _tables(orc, t)
    reqs = t["requests_flush"] if cap == 80 else t["requests"]
    want = _unpack(t[key + "_hits"], len(reqs))
    c = E.GpuCache(policy, cap, 26, 36, 32, "python")
    c.set_backing([torch.from_numpy(x).cuda() for x in tabs])
    c.serve_start(n_slots=3, idle_us=300)
    for i, rq in enumerate(reqs):
        way = i % 3
        if way == 0:
            hit, rows = c.serve_request(rq)
        elif way == 1:
            rows = torch.full((26, 36), -7.0, dtype=torch.float32, device="cuda")
            torch.cuda.synchronize()
            hit = c.serve_request_to(rq, rows)
        else:
            ids = torch.from_numpy(np.stack([rq.astype(np.int64), np.full(26, -1, np.int64)], 1)).cuda()   # (T, 2): element 0 is the id
            rows = torch.full((26, 1, 36), -7.0, dtype=torch.float32, device="cuda")
            torch.cuda.synchronize()
            hit = c.serve_request_to(ids, rows)
        assert np.array_equal(hit.astype(bool), want[i]), (i, way)
        if i % 31 < 3 or i < 9:
            got = rows.reshape(26, 36).cpu().numpy()
            for k in range(26):
                assert np.array_equal(got[k], tabs[k][rq[k]]), (i, way, k)
    c.serve_stop()
    st = c.stats()
    assert st["n_requests"] == len(reqs) and st["n_hits"] == int(want.sum())
    c27 = E.GpuCache(policy, cap, 27, 36, 32, "python")     # the address rides in the id words of tables 26 and 27
    c27.set_backing([torch.from_numpy(tabs[k % 26]).cuda() for k in range(27)])
    c27.serve_start(n_slots=2, idle_us=100)
    with pytest.raises(E.EvsError):
        c27.serve_request_to(np.zeros(27, np.int32), torch.zeros((27, 36), dtype=torch.float32, device="cuda"))
    c27.serve_stop()


def test_evlfu_approx_mode(E, orc):
    t = load_golden("cache_traces")
    tabs = _tables(orc, t)
    c, hits, outs = _run(E, "evlfu", 768, tabs, t["requests"], 50, approx=20)
    assert np.array_equal(hits, _unpack(t["evlfu_cap768_approx20_hits"], len(t["requests"])))
    np.testing.assert_array_equal(c.dump(), t["evlfu_cap768_approx20_final_buckets"])


@pytest.mark.parametrize("cap,chunk", [(64, 1), (768, 100), (80, 1200)])
def test_lru_trace(E, orc, cap, chunk):
    t = load_golden("cache_traces")
    tabs = _tables(orc, t)
    reqs = t["requests_flush"] if cap == 80 else t["requests"]
    c, hits, outs = _run(E, "lru", cap, tabs, reqs, chunk)
    assert np.array_equal(hits, _unpack(t["lru_cap%d_hits" % cap], len(reqs)))
    assert np.array_equal(outs[:, 5, :], tabs[5][reqs[:, 5]])
    np.testing.assert_array_equal(c.dump()[:, 1:], t["lru_cap%d_final_order" % cap])


@pytest.mark.parametrize("cap,chunk", [(64, 1), (768, 100), (80, 1200)])
def test_lfu_trace(E, orc, cap, chunk):
    t = load_golden("cache_traces")
    tabs = _tables(orc, t)
    reqs = t["requests_flush"] if cap == 80 else t["requests"]
    c, hits, outs = _run(E, "lfu", cap, tabs, reqs, chunk)
    assert np.array_equal(hits, _unpack(t["lfu_cap%d_hits" % cap], len(reqs)))
    assert np.array_equal(outs[:, 11, :], tabs[11][reqs[:, 11]])
    np.testing.assert_array_equal(c.dump(), t["lfu_cap%d_final_freq" % cap])


@pytest.mark.parametrize("variant", ["cpp", "cython"])
def test_evlfu_variants_vs_oracle(E, orc, variant):
    """mixed_precs_caching (0.3/0.95, n flushed) and Cython (0.4/1.0) constants: oracle vs GPU."""
    t = load_golden("cache_traces")
    tabs = _tables(orc, t)
    reqs = t["requests_flush"]
    o = orc.EvLFU(80, tabs, variant=variant)
    want = np.stack([o.request(rq)[0].copy() for rq in reqs])
    c, hits, _ = _run(E, "evlfu", 80, tabs, reqs, 97, variant=variant)
    assert np.array_equal(hits, want)
    np.testing.assert_array_equal(c.dump(), o.dump())


@pytest.mark.parametrize("stream,cap,chunk", [("main", 64, 1), ("main", 768, 100), ("flush", 52, 1), ("flush", 78, 33),
                                              ("flush", 300, 1200), ("main", 300, 7)])
def test_evlfu_cython_variant_matches_compiled_reference(E, orc, stream, cap, chunk):
    """variant='cython' on the GPU against traces of the reference's COMPILED C++ (EvLFU_C1_Cython/EvLFU.cpp,
    tests/golden/cython_traces.npz): hit flags, rows, final list order and counters bit-exact, flushes included."""
    t = load_golden("cython_traces")
    tabs = _tables(orc, t)
    reqs = t["requests_flush"] if stream == "flush" else t["requests"]
    tag = "cython_%s_cap%d" % (stream, cap)
    c, hits, outs = _run(E, "evlfu", cap, tabs, reqs, chunk, variant="cython")
    want = _unpack(t[tag + "_hits"], len(reqs))
    assert np.array_equal(hits, want)
    for k in range(26):
        assert np.array_equal(outs[:, k, :], tabs[k][reqs[:, k]])
    np.testing.assert_array_equal(c.dump(), t[tag + "_final_buckets"])
    st = c.stats()
    assert [st["min_c1"], st["n_perfect"], st["size"], st["n_flush"]] == list(t[tag + "_state"])


@pytest.mark.parametrize("codec", [8, 4, 16])
def test_cache_over_reduced_precision_rows(E, orc, codec):
    """a cache tier holding 8/4/16-bit rows (the reference's C2 precisions) decodes on output."""
    t = load_golden("cache_traces")
    tabs = _tables(orc, t)
    raws = [orc.encode_table(np.clip(w * 8, -1, 1), codec) for w in tabs]
    reqs = t["requests"][:400]
    c, hits, outs = _run(E, "evlfu", 300, tabs, reqs, 40, codec=codec, raws=raws)
    assert np.array_equal(hits, _unpack(t["evlfu_cap300_hits"], 1500)[:400])  # policy is precision-agnostic
    for k in (0, 2, 8, 25):
        want = orc.decode(raws[k][reqs[:, k]], codec, 36)
        assert np.array_equal(outs[:, k, :].view(np.uint32), want.view(np.uint32))


def test_plugin_surface_evstore(E, orc, tmp_path):
    """apply_emb_evstore + cache module + storage manager, as dlrm_s_pytorch_C1.py drives them."""
    from evstore_dlrm_amd import evstore_ops
    from evstore_dlrm_amd.cache_algo import EvLFU_C1
    from evstore_dlrm_amd.emb_storage import storage_manager as sm
    t = load_golden("cache_traces")
    tabs = _tables(orc, t)
    (tmp_path / "binary").mkdir()
    for k, w in enumerate(tabs):
        w.tofile(tmp_path / "binary" / ("ev-table-%d.bin" % (k + 1)))
    # the reference's file / mmap readers
    for st in (sm.EmbStorage.FILEPY, sm.EmbStorage.MMAPFILEPY):
        sm.storage_type = st
        sm.load_ev_table_into_emb_stor(str(tmp_path))
        for (tb, r), want in zip(t["reader_probe"], t["reader_rows"]):
            assert np.array_equal(np.asarray(sm.get_val_from_storage(int(tb), int(r)), np.float32), want)
        sm.close_any_db_conn()
    for st in (sm.EmbStorage.HBM, sm.EmbStorage.PINNED):
        sm.storage_type = st
        sm.load_ev_table_into_emb_stor(str(tmp_path))
        EvLFU_C1.init(768)
        evstore_ops.cache_algo = "evlfu"
        evstore_ops.perfect_hit = 0
        want = _unpack(t["evlfu_cap768_hits"], 1500)
        n = 300
        for i in range(n):
            lS_i = torch.from_numpy(t["requests"][i].astype(np.int64)).reshape(26, 1)
            ly = evstore_ops.apply_emb_evstore(None, lS_i, None, None, use_gpu=False, use_emb_cache=True)
            assert len(ly) == 26 and ly[0].shape == (1, 36) and ly[0].requires_grad
            if i % 37 == 0:
                for k in range(26):
                    assert np.array_equal(ly[k].detach().numpy()[0], tabs[k][t["requests"][i][k]])
        assert evstore_ops.perfect_hit == int(want[:n].all(1).sum())
        assert EvLFU_C1.stats()["n_hits"] == int(want[:n].sum())
        # ... and on with use_gpu=True as dlrm_wrap hands the ids over (on the device): with the tables in HBM the GPU engine's
        # resident server answers (one extension call: ids through pinned staging, the mailbox, 26 views over one copy of the
        # answer's ring slot); same trace, rows on the device
        m = 120
        for i in range(n, n + m):
            lS_i = torch.from_numpy(t["requests"][i].astype(np.int64)).reshape(26, 1).cuda()
            ly = evstore_ops.apply_emb_evstore(None, lS_i, None, None, use_gpu=True, use_emb_cache=True)
            assert len(ly) == 26 and ly[0].shape == (1, 36) and ly[0].requires_grad and ly[0].is_cuda
            if i % 7 == 0:
                got = torch.cat([v.detach() for v in ly]).cpu().numpy()
                assert np.array_equal(got, np.stack([tabs[k][t["requests"][i][k]] for k in range(26)]))
        assert evstore_ops.perfect_hit == int(want[:n + m].all(1).sum())
        assert EvLFU_C1.stats()["n_hits"] == int(want[:n + m].sum())
        _, ly = sm.request_to_emb_storage([int(v) for v in t["requests"][0]])
        assert np.array_equal(ly[3].detach().numpy()[0], tabs[3][t["requests"][0][3]])
        sm.close_any_db_conn()


def test_plugin_surface_evstore_cython_branch(E, orc):
    """a5, the `evlfu_cython` branch (dlrm_s_pytorch_C1.py:250-253): apply_emb_evstore -> EvLFU.crequest -> 26 CPU
    FloatTensors; hit pattern = the trace of the reference's EvLFU.cpp compiled in place (constants 0.4 / 1.0)."""
    from evstore_dlrm_amd import evstore_ops
    from evstore_dlrm_amd.cache_algo import EvLFU
    from evstore_dlrm_amd.emb_storage import storage_manager as sm
    t = load_golden("cython_traces")
    tabs = _tables(orc, t)
    sm.use_device_tables([torch.from_numpy(np.ascontiguousarray(w)).cuda() for w in tabs], 32)
    reqs = t["requests_flush"]
    want = _unpack(t["cython_flush_cap78_hits"], len(reqs))
    EvLFU.cinit(78)
    EvLFU.cload_ev_tables()
    evstore_ops.cache_algo = "evlfu_cython"
    evstore_ops.perfect_hit = 0
    n = 400
    for i in range(n):
        lS_i = torch.from_numpy(reqs[i].astype(np.int64)).reshape(26, 1)
        ly = evstore_ops.apply_emb_evstore(None, lS_i, None, None, use_gpu=False, use_emb_cache=True)
        assert len(ly) == 26 and ly[0].shape == (1, 36) and ly[0].dtype == torch.float32 and not ly[0].is_cuda
        if i % 41 == 0:
            for k in range(26):
                assert np.array_equal(ly[k].numpy()[0], tabs[k][reqs[i][k]])
    assert evstore_ops.perfect_hit == int(want[:n].all(1).sum())
    assert EvLFU.stats()["n_hits"] == int(want[:n].sum())
    hit, rows = EvLFU.crequest([int(v) for v in reqs[n]])
    assert hit == [bool(v) for v in want[n]] and len(rows) == 26 and len(rows[0]) == 36 and isinstance(rows[0][0], float)
    EvLFU.cclose_ev_tables()
    evstore_ops.cache_algo = "evlfu"
    sm.close_any_db_conn()


def test_plugin_surface_evstore_unknown_algo_exits(E):
    from evstore_dlrm_amd import evstore_ops
    evstore_ops.cache_algo = "arc"
    with pytest.raises(SystemExit):
        evstore_ops.apply_emb_evstore(None, torch.zeros((26, 1), dtype=torch.int64), None, None, use_emb_cache=True)
    evstore_ops.cache_algo = "cpp_algo_socket"   # the loopback-socket transport is out of scope: prints and exits
    with pytest.raises(SystemExit):
        evstore_ops.apply_emb_evstore(None, torch.zeros((26, 1), dtype=torch.int64), None, None, use_emb_cache=True)
    evstore_ops.cache_algo = "evlfu"


@pytest.mark.parametrize("prec,layers", [(8, 1), (32, 1), (8, 2), (8, 3)])
def test_reference_cabi_ev_lookup(E, orc, tmp_path, prec, layers):
    """ev_lookup / get_ev_values / print_perfect_hit through ctypes, as cpp_socket_client.py binds them."""
    import json
    import os
    import subprocess
    import sys
    t = load_golden("cache_traces")
    tabs = _tables(orc, t)
    sub = {32: "ev-table", 16: "ev-table-16", 8: "ev-table-8", 4: "ev-table-4"}[prec]
    (tmp_path / sub / "binary").mkdir(parents=True)
    for k, w in enumerate(tabs):
        orc.encode_table(np.clip(w * 8, -1, 1), prec).tofile(tmp_path / sub / "binary" / ("ev-table-%d.bin" % (k + 1)))
    if layers == 3:
        (tmp_path / "altkeys").mkdir()
        rs = np.random.RandomState(4)
        for k, w in enumerate(tabs):
            ((rs.randint(0, len(w), size=len(w)) * 100 + (k + 1)).astype(">u4")).tofile(tmp_path / "altkeys" / ("ev-table-%d.bin" % (k + 1)))
    if layers >= 2:
        (tmp_path / "ev-table-4" / "binary").mkdir(parents=True)
        for k, w in enumerate(tabs):
            orc.encode_table(np.clip(w * 8, -1, 1), 4).tofile(tmp_path / "ev-table-4" / "binary" / ("ev-table-%d.bin" % (k + 1)))
    np.save(tmp_path / "reqs.npy", t["requests"][:1200] if layers == 3 else t["requests"][:250])
    child = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_ev_lookup_child.py")
    out = subprocess.run([sys.executable, child, str(tmp_path), str(prec), "40" if layers == 3 else "100", str(layers)], capture_output=True,
                         text=True, timeout=300)
    line = [l for l in out.stdout.splitlines() if l.startswith("RESULT ")]
    assert line, out.stdout[-2000:] + out.stderr[-2000:]
    r = json.loads(line[0][7:])
    assert r["ok"] and r["same_buf"] and r["rc_dead"] == -1
    # ev_lookup was called once more after the loop (same_buf probe): counter >= the oracle's count
    assert r["perfect_oracle"] <= r["counter"] <= r["perfect_oracle"] + 1 and r["after_print"] == 0
    assert "Perfect hit" in out.stdout
    if layers == 3:
        # the child's extra ev_lookup probe may add alt-key hits of its own
        assert r["aprx"][1] <= r["aprx"][0] <= r["aprx"][1] + 26 and "C3 Indiv-Hit" in out.stdout


@pytest.mark.parametrize("var", ["2-32-16-4000", "2-32-8-4000", "2-32-4-4000", "2-16-8-4000", "2-16-4-4000", "2-8-4-4000",
                                 "1-32-4-3000", "1-16-4-3000", "1-4-4-3000"])
def test_cabi_precision_builds_vs_compiled_reference(E, orc, tmp_path, var):
    """a9's sibling builds (MAIN_PRECISION 32 / 16 with a 16 / 8 / 4-bit C2, and the single-tier builds): ev_lookup of
    libevstore_hip configured like the build == what the reference COMPILED with those #defines served
    (tests/golden/mgr_variants.npz: serving precision of every key identical until C1 fills, >= 99 % per block after,
    perfect-hit counter identical while nothing was evicted), every row bit-identical to the oracle, and the tier
    capacities equal the reference constructors' (incl. the x16 of an 8-bit C2)."""
    import json
    import os
    import subprocess
    import sys
    child = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_ev_lookup_variant_child.py")
    out = subprocess.run([sys.executable, child, str(tmp_path), var], capture_output=True, text=True, timeout=600)
    line = [l for l in out.stdout.splitlines() if l.startswith("RESULT ")]
    assert line, out.stdout[-2000:] + out.stderr[-2000:]
    r = json.loads(line[0][7:])
    assert r["exact_vs_oracle"] and r["no_garbage"], r
    assert r["caps"][:2] == r["caps_ref"], r
    assert r["perfect_prefix_equal"] and r["nb"] >= 1, r
    if var.startswith("2-"):
        assert r["first_mine"] == r["first_ref"] and r["prefill_equal"] and r["min_block_agreement"] >= 0.99, r
    else:
        assert abs(r["perfect"][0] - r["perfect"][1]) <= max(8, 0.4 * r["perfect"][1]), r


def test_two_tier_c1c2_vs_oracle(E, orc):
    """a9: request_to_c1_c2 on the GPU == the oracle restatement (itself pinned to the compiled
    reference in tests/test_oracle_golden.py): serving tier, values, both tiers' final state."""
    from evstore_dlrm_amd import gpu_cache
    rs = np.random.RandomState(3)
    n_rows = [400] * 26
    ws = [rs.uniform(-1, 1, size=(n, 36)).astype(np.float32) for n in n_rows]
    raw8 = [orc.encode_table(w, 8) for w in ws]
    raw4 = [orc.encode_table(w, 4) for w in ws]
    dec8 = [orc.decode(r, 8, 36) for r in raw8]
    dec4 = [orc.decode(r, 4, 36) for r in raw4]
    cap1, cap2 = 600, 1200
    reqs = np.zeros((900, 26), np.int32)
    for i in range(len(reqs)):
        reqs[i] = rs.randint(0, 400, 26)
        if i > 20 and rs.rand() < 0.4:
            reqs[i] = reqs[i - 1 - rs.randint(15)]
            reqs[i] = np.where(rs.rand(26) < 0.07, rs.randint(0, 400, 26), reqs[i])
    o = orc.C1C2(cap1, cap2, dec8, dec4)
    want_tier, want_out, perfect = [], [], 0
    for rq in reqs:
        t, out, p = o.request(rq)
        want_tier.append(t.copy()); want_out.append(out.copy()); perfect += p
    c1 = E.GpuCache("evlfu", cap1, 26, 36, 8, "cpp")
    c2 = E.GpuCache("evlfu", cap2, 26, 36, 4, "cpp")
    c1.set_backing([torch.from_numpy(r).cuda() for r in raw8])
    c2.set_backing([torch.from_numpy(r).cuda() for r in raw4])
    r = torch.from_numpy(reqs).cuda()
    tiers, outs = [], []
    for s in range(0, len(reqs), 111):
        t, out = gpu_cache.request_c1c2(c1, c2, r[s:s + 111].contiguous())
        tiers.append(t.cpu().numpy()); outs.append(out.cpu().numpy())
    tiers, outs = np.concatenate(tiers), np.concatenate(outs)
    assert np.array_equal(tiers, np.stack(want_tier))
    assert np.array_equal(outs.view(np.uint32), np.stack(want_out).view(np.uint32))
    np.testing.assert_array_equal(c1.dump(), o.c1.dump())
    np.testing.assert_array_equal(c2.dump(), o.c2.dump())
    assert c1.stats()["n_perfect_hits"] == perfect
    assert (tiers == 2).sum() > 100 and (tiers == 1).sum() > 100  # both tiers actually serve


def _zipf_requests(n_rows, n_req, seed, alpha=1.15):
    rs = np.random.RandomState(seed)
    perms = [rs.permutation(n) for n in n_rows]
    reqs = np.zeros((n_req, len(n_rows)), np.int32)
    for k, n in enumerate(n_rows):
        reqs[:, k] = perms[k][np.minimum(rs.zipf(alpha, n_req) - 1, n - 1)]
    hot = reqs[rs.randint(0, n_req, 64)]
    rep = rs.rand(n_req) < 0.3
    reqs[rep] = hot[rs.randint(0, 64, rep.sum())]
    return reqs


POLICIES = ["sampled", "plan"]   # evs_cache_set_batch_policy: one update kernel with sampled victims / insert-plan-evict-assign
POLICIES1 = POLICIES + ["setassoc"]   # ... / 8-way set-associative (single tier, tables in HBM)
SA_WAYS = 8


def _sa_geom(cap, n_rows, cap2=None, n_rows2=None):
    """Geometry of the set-associative policy restated (csrc/evs_cache.hip: sa_single_feasible / sa_pair_geometry):
    -> (nset, ways1, ways2, bits).  One tier: cap // 8 sets of 8 ways.  A pair that starts out together: nset =
    max(cap1 // 8, ceil(cap2 / 16)) sets, min(cap // nset, 16) ways per tier -- unless a tier would get fewer than 4 ways
    (then each tier keeps sets of its own).  The key universe is the dense row number over all tables."""
    rows = [max(a, b) for a, b in zip(n_rows, n_rows2 or n_rows)]
    total = sum(rows)
    bits = 1
    while (1 << bits) < total:
        bits += 1
    if cap2 is None:
        return cap // SA_WAYS, SA_WAYS, None, bits
    nset = max(cap // 8, (cap2 + 15) // 16, 1)
    w1, w2 = min(cap // nset, 16), min(cap2 // nset, 16)
    if w1 < 4 or w2 < 4:
        return None
    return nset, w1, w2, bits


def _sa_sets(keys_tr, nset, n_rows, bits):
    """Set of each (table_1based, row) key under the set-associative policy (csrc/evs_hash.h: sa_perm / sa_split): the dense
    row number through two rounds of odd multiply + xorshift on `bits` bits, modulo the number of sets."""
    base = np.concatenate([[0], np.cumsum(np.asarray(n_rows, np.uint64))]).astype(np.uint64)
    t = np.asarray([t for t, _ in keys_tr], np.int64) - 1
    x = base[t] + np.asarray([r for _, r in keys_tr], np.uint64)
    mask, half = np.uint64((1 << bits) - 1), np.uint64((bits + 1) // 2)
    x = (x * np.uint64(0x9E3779B1)) & mask
    x ^= x >> half
    x = (x * np.uint64(0x85EBCA6B)) & mask
    x ^= x >> half
    return (x % np.uint64(nset)).astype(np.int64)


@pytest.mark.parametrize("cap_frac,batch,codec", [(0.10, 512, 32), (0.02, 160, 32), (0.3, 4100, 32),
                                                  (0.10, 512, 8), (0.02, 160, 4), (0.3, 1300, 16), (0.05, 700, 4)])
def test_update_inside_the_probe_launch_of_the_set_associative_tier(E, orc, cap_frac, batch, codec):
    """Round 5: evs_cache_lookup_interact on a set-associative tier makes the batch's policy update INSIDE the probe +
    interaction launch (csrc/evs_fused_rf.hip, ProbeArgs::arena_w: the thread that misses a key claims a way of its set, the
    lanes that gather the row store it into the arena; two arena rows per way, evs_hash.h; a reduced-precision tier does the
    same in its consumer, csrc/evs_fused_rfq.hip -- its lanes hold the raw bytes).  Lookups back to back: R of every
    batch = the interaction over the TRUE table rows whatever is being replaced underneath (rtol 1e-5 vs the oracle).  A hit
    flag says the row came from the cache: flag => the key was resident when the batch arrived; a resident key flagged as a
    miss was retired by this very batch's inserts (at most as many as it evicted).  Behind every batch: no duplicate keys,
    every key in its own set, no set above its ways, size and histogram = the statistics, hits counted = flags seen;
    missed keys are resident afterwards unless their set is full.  EVS_CACHE_INLINE=0 (the update as a launch of its own,
    strict snapshot flags) is what the other tests of this file pin through lookup_batch."""
    n_rows = [3000, 40, 20000, 700, 5, 9000, 1500, 12, 26000, 300, 8000, 64, 2200, 17000, 3, 450, 5000, 90, 13000,
              2, 7000, 30, 1000, 11000, 150, 4000]
    tabs = orc.kaggle_tables(n_rows, 21)
    if codec == 32:
        dev_tabs = [torch.from_numpy(t).cuda() for t in tabs]
    else:   # (the rows the interaction sees: the decoded ones; spread over the codec's range so that rows differ)
        raws = [orc.encode_table(np.clip(t * np.sqrt(len(t)), -1, 1), codec) for t in tabs]
        dev_tabs = [torch.from_numpy(a).cuda() for a in raws]
        tabs = [orc.decode(a, codec, 36) for a in raws]
    cap = int(cap_frac * sum(n_rows))
    reqs = _zipf_requests(n_rows, 9 * batch, 5)
    rs = np.random.RandomState(6)
    x_np = rs.uniform(-1, 1, size=(batch, 36)).astype(np.float32)
    x = torch.from_numpy(x_np).cuda()
    c = E.GpuCache("evlfu", cap, 26, 36, codec, "python").set_batch_policy("setassoc")
    c.set_backing(dev_tabs)
    r = torch.from_numpy(reqs).cuda()
    nset, ways, _, bits = _sa_geom(cap, n_rows)

    def residents():
        d = c.batch_dump()
        keys = [(int(t), int(rw)) for _, t, rw in d]
        st = c.batch_stats()
        assert len(set(keys)) == len(keys) and len(keys) == st["size"] <= cap
        assert np.array_equal(np.bincount(d[:, 0], minlength=27), np.array(st["hist"]))
        if keys:
            assert np.bincount(_sa_sets(keys, nset, n_rows, bits), minlength=nset).max() <= ways
        return {(int(t), int(rw)): int(p) for p, t, rw in d}, st

    flags_seen = 0
    res, st0 = {}, {"n_evict": 0}      # (nothing resident, nothing counted: the batched path has not run yet)
    k = 0
    for run in (1, 3, 1, 2, 2):       # lookups back to back, the state looked at between the runs
        before, evict0 = res, st0["n_evict"]
        Rs, hits = [], []
        for j in range(run):
            rq = r[(k + j) * batch:(k + j + 1) * batch].contiguous()
            hit, R = c.lookup_interact(rq, x)
            Rs.append(R.clone()); hits.append(hit.clone())
        res, st0 = residents()
        asked = set()                  # keys the earlier batches of the run asked for: what can be new for the later ones
        lost = set()
        for j in range(run):
            rq = reqs[(k + j) * batch:(k + j + 1) * batch]
            hit = hits[j].cpu().numpy().astype(bool)
            flags_seen += int(hit.sum())
            ly = [tabs[t][rq[:, t]] for t in range(26)]
            np.testing.assert_allclose(Rs[j].cpu().numpy(), orc.interact_features(x_np, ly), rtol=1e-5, atol=2e-6)
            keys = [[(t + 1, int(rq[b, t])) for t in range(26)] for b in range(batch)]
            was = np.array([[key in before for key in row] for row in keys])
            new = np.array([[key in asked for key in row] for row in keys])
            assert not (hit & ~was & ~new).any()                  # a hit was resident when the batch arrived
            if j == 0:
                lost |= {keys[b][t] for b in range(batch) for t in range(26) if was[b, t] and not hit[b, t]}
            asked |= {key for row in keys for key in row}
        # resident keys reported as misses: retired by the batch's own inserts (each such insert is an eviction)
        assert len(lost) <= st0["n_evict"] - evict0
        if run == 1:                   # a missed key is resident afterwards unless its set is full
            rq = reqs[k * batch:(k + 1) * batch]
            hit = hits[0].cpu().numpy().astype(bool)
            missed = sorted({(t + 1, int(rq[b, t])) for b in range(batch) for t in range(26) if not hit[b, t]})
            gone = [key for key in missed if key not in res]
            if gone:
                per_set = np.bincount(_sa_sets(list(res), nset, n_rows, bits), minlength=nset)
                assert (per_set[_sa_sets(gone, nset, n_rows, bits)] == ways).all()
        k += run
    assert st0["n_hits"] == flags_seen and st0["n_requests"] == k * batch
    # what the launches left in the arena, bit for bit: the last batches again through the rows-out path (the update as a launch
    # of its own: snapshot flags) -- every resident key is served from the arena
    for j in (k - 2, k - 1):
        rq = reqs[j * batch:(j + 1) * batch]
        hit, out = c.lookup_batch(r[j * batch:(j + 1) * batch].contiguous())
        hit, out = hit.cpu().numpy().astype(bool), out.cpu().numpy()
        assert np.array_equal(hit, np.array([[(t + 1, int(rq[b, t])) in res for t in range(26)] for b in range(batch)])) and hit.any()
        for t in range(26):
            assert np.array_equal(out[:, t, :].view(np.uint32), tabs[t][rq[:, t]].view(np.uint32))
        res, _ = residents()


@pytest.mark.parametrize("policy", POLICIES1)
@pytest.mark.parametrize("cap_frac,batch", [(0.10, 256), (0.02, 64), (0.5, 1024)])
def test_batched_cache_invariants_and_hit_rate(E, orc, cap_frac, batch, policy):
    """Batched (snapshot) EvLFU under both policy updates: rows exact, hit flags = residency at batch start, no duplicate
    keys, size <= capacity, histogram consistent, priorities monotone; hit rate tracks the sequential oracle."""
    n_rows = [3000, 40, 20000, 700, 5, 9000, 1500, 12, 26000, 300, 8000, 64, 2200, 17000, 3, 450, 5000, 90, 13000,
              2, 7000, 30, 1000, 11000, 150, 4000]
    tabs = orc.kaggle_tables(n_rows, 21)
    cap = int(cap_frac * sum(n_rows))
    reqs = _zipf_requests(n_rows, 4096, 2)
    c = E.GpuCache("evlfu", cap, 26, 36, 32, "python").set_batch_policy(policy)
    c.set_backing([torch.from_numpy(t).cuda() for t in tabs])
    r = torch.from_numpy(reqs).cuda()
    resident = {}
    hits_total = 0
    for s in range(0, len(reqs), batch):
        rq = reqs[s:s + batch]
        hit, out = c.lookup_batch(r[s:s + batch].contiguous())
        hit, out = hit.cpu().numpy().astype(bool), out.cpu().numpy()
        for k in range(26):
            assert np.array_equal(out[:, k, :], tabs[k][rq[:, k]])
        want_hit = np.array([[(k + 1, int(rq[b, k])) in resident for k in range(26)] for b in range(len(rq))])
        assert np.array_equal(hit, want_hit)
        hits_total += int(hit.sum())
        d = c.batch_dump()
        st = c.batch_stats()
        keys = [(int(t), int(rw)) for _, t, rw in d]
        assert len(set(keys)) == len(keys) and len(keys) == st["size"] <= cap
        assert np.array_equal(np.bincount(d[:, 0], minlength=27), np.array(st["hist"]))
        new_res = {(int(t), int(rw)): int(p) for p, t, rw in d}
        for key, p in new_res.items():
            if key in resident:
                assert p >= resident[key]
        # every key of the batch that could be kept is resident afterwards when there was room (the sampled update may
        # pick a key that was HIT in this batch as a victim while free entries remain elsewhere: missed keys only there)
        if policy == "setassoc":
            # every resident key sits in its own set, no set holds more than its ways, and a missed key is resident afterwards
            # unless its set is full (it can only have been turned away by SA_WAYS other new keys of this batch)
            kl = list(new_res)
            nset, ways, _, bits = _sa_geom(cap, n_rows)
            per_set = np.bincount(_sa_sets(kl, nset, n_rows, bits), minlength=nset) if kl else np.zeros(1, int)
            assert per_set.max() <= ways
            missed = sorted({(k + 1, int(rq[b, k])) for b in range(len(rq)) for k in range(26) if not hit[b, k]})
            gone = [key for key in missed if key not in new_res]
            if gone:
                assert (per_set[_sa_sets(gone, nset, n_rows, bits)] == ways).all()
        elif st["size"] < cap:
            assert all((k + 1, int(rq[b, k])) in new_res for b in range(len(rq)) for k in range(26)
                       if policy == "plan" or not hit[b, k])
        resident = new_res
    st = c.batch_stats()
    assert st["n_hits"] == hits_total and st["n_requests"] == len(reqs)
    o = orc.EvLFU(cap, tabs)
    seq_hits = sum(int(o.request(rq)[0].sum()) for rq in reqs)
    rate_b, rate_s = hits_total / reqs.size, seq_hits / reqs.size
    # the snapshot cannot hit keys first inserted inside the same batch: allow that much plus 5 points
    first_seen_in_batch = 0
    seen = set()
    for s in range(0, len(reqs), batch):
        local = set()
        for rq in reqs[s:s + batch]:
            for k in range(26):
                key = (k, int(rq[k]))
                if key in local and key not in seen:
                    first_seen_in_batch += 1
                local.add(key)
        seen |= local
    assert rate_b >= rate_s - first_seen_in_batch / reqs.size - 0.05, (rate_b, rate_s)
    assert rate_b <= rate_s + 0.05, (rate_b, rate_s)


@pytest.mark.parametrize("policy", POLICIES)
def test_batched_cache_over_host_memory_backing(E, orc, policy):
    """the miss tier in pinned HOST memory (the reference's C3 / mmap miss path): the batched
    lookup serves hits from the HBM arena and misses straight from host rows (pointer table -> fused kernel),
    fills the arena from host rows, and gives the bits of the all-HBM path."""
    n_rows = [500, 7, 9000, 40, 2500, 3] + [100] * 20
    tabs = orc.kaggle_tables(n_rows, 9)
    host = [torch.from_numpy(np.ascontiguousarray(t)).pin_memory() for t in tabs]
    dev = [torch.from_numpy(t).cuda() for t in tabs]
    ch = E.GpuCache("evlfu", 2500, 26, 36, 32).set_batch_policy(policy)
    cd = E.GpuCache("evlfu", 2500, 26, 36, 32).set_batch_policy(policy)
    ch.set_backing(host)
    cd.set_backing(dev)
    reqs = _zipf_requests(n_rows, 1200, 6)
    r = torch.from_numpy(reqs).cuda()
    for s in range(0, 1200, 300):
        x = torch.rand(300, 36, device="cuda")
        hit_h, R_h = ch.lookup_interact(r[s:s + 300].contiguous(), x)
        hit_d, R_d = cd.lookup_interact(r[s:s + 300].contiguous(), x)
        assert torch.equal(R_h, R_d)   # the rows are the table rows wherever they are served from
        if policy == "plan":           # (the sampled update's victims depend on thread timing: residency may differ)
            assert torch.equal(hit_h, hit_d)
        hb, rows = ch.lookup_batch(r[s:s + 300].contiguous())
        for k in range(26):
            assert np.array_equal(rows[:, k, :].cpu().numpy(), tabs[k][reqs[s:s + 300, k]])
    assert ch.batch_stats()["n_hits"] > 0 and ch.batch_stats()["size"] > 0


@pytest.mark.parametrize("policy", POLICIES1)
def test_batched_cache_smaller_than_one_batch(E, orc, policy):
    """A cache far smaller than the unique keys of one batch: the hash can run out of empty words between
    rebuilds; every walk is bounded, rows stay exact, the cache never exceeds its capacity."""
    n_rows = [400] * 26
    tabs = orc.kaggle_tables(n_rows, 3)
    c = E.GpuCache("evlfu", 60, 26, 36, 32).set_batch_policy(policy)
    c.set_backing([torch.from_numpy(t).cuda() for t in tabs])
    rs = np.random.RandomState(0)
    for it in range(6):
        rq = rs.randint(0, 400, size=(600, 26)).astype(np.int32)
        hit, out = c.lookup_batch(torch.from_numpy(rq).cuda())
        out = out.cpu().numpy()
        for k in range(26):
            assert np.array_equal(out[:, k, :], tabs[k][rq[:, k]])
        st = c.batch_stats()
        d = c.batch_dump()
        assert st["size"] == len(d) <= 60 and len({(int(t), int(r)) for _, t, r in d}) == len(d)


@pytest.mark.parametrize("budget_kb,cap", [(0, 3000), (200, 3000), (10 ** 6, 3000), (200, 40)])
def test_file_backed_miss_tier(E, orc, tmp_path, budget_kb, cap):
    """SURVEY 8(f).1 / mmap_file_read.py:32-40: the miss tier is a directory of ev-table-N.bin files larger than the
    pinned budget.  Tables are registered (zero-copy) smallest first while they fit; the rest are STAGED: the host's
    reader pool copies the batch's de-duplicated new rows out of the mappings.  Rows exact for hits, staged misses,
    registered misses and keys that found no room (cap=40: a cache smaller than one batch); cache invariants; the
    fused consumer gives the same R as the rows; every staged row is read from the file once per batch at most."""
    rs = np.random.RandomState(9)
    n_rows = [3000 if k % 5 == 0 else (40 if k % 3 == 0 else 700) for k in range(26)]
    tabs = [rs.uniform(-1, 1, size=(n, 36)).astype(np.float32) for n in n_rows]
    paths = []
    for k, w in enumerate(tabs):
        p = tmp_path / ("ev-table-%d.bin" % (k + 1))
        w.tofile(p)
        paths.append(str(p))
    total = sum(n * 144 for n in n_rows)
    ft = E.FileTier(paths, 144, budget_kb * 1024)
    assert ft.pinned_bytes <= budget_kb * 1024 and ft.n_rows == n_rows
    if budget_kb == 0:
        assert not any(ft.registered)
    elif budget_kb * 1024 >= total:
        assert all(ft.registered)
    else:   # the file set is larger than the pinned budget: the small tables are registered, the big ones staged
        assert any(ft.registered) and not all(ft.registered)
        assert max(n for n, r in zip(n_rows, ft.registered) if r) <= min(n for n, r in zip(n_rows, ft.registered) if not r)
    c = E.GpuCache("evlfu", cap, 26, 36, 32)
    c.set_file_backing(ft)
    B = 500
    for it in range(8):
        hot = rs.rand(B, 26) < 0.6
        rq = np.where(hot, rs.randint(0, 25, size=(B, 26)), np.stack([rs.randint(0, n, size=B) for n in n_rows], 1)).astype(np.int32)
        rq = np.minimum(rq, np.asarray(n_rows, np.int32) - 1)
        r = torch.from_numpy(rq).cuda()
        before = c.staged_rows()
        hit, out = c.lookup_batch(r)
        out = out.cpu().numpy()
        for k in range(26):
            assert np.array_equal(out[:, k, :], tabs[k][rq[:, k]]), (it, k)
        staged_keys = {(k, int(v)) for k in range(26) if not ft.registered[k] for v in rq[:, k]}
        if cap >= 1000:    # every missing row is read from its file once per batch at most
            assert c.staged_rows() - before <= len(staged_keys)
        else:              # a hash smaller than the batch's misses: the keys it drops are staged per request position
            assert c.staged_rows() - before <= B * sum(1 for k in range(26) if not ft.registered[k])
        x = torch.rand(B, 36, device="cuda")
        _, R = c.lookup_interact(r, x)
        want = orc.interact_features(x.cpu().numpy(), [tabs[k][rq[:, k]] for k in range(26)])
        np.testing.assert_allclose(R.cpu().numpy(), want, rtol=1e-5, atol=2e-6)
        st = c.batch_stats()
        d = c.batch_dump()
        assert st["size"] == len(d) <= cap and len({(int(t), int(rr)) for _, t, rr in d}) == len(d)
    assert c.batch_stats()["n_hits"] > 0
    if not all(ft.registered):
        assert c.staged_rows() > 0
        with pytest.raises(E.EvsError):   # the exact batch-1 machine cannot wait for the host's reader pool
            c2 = E.GpuCache("evlfu", 100, 26, 36, 32)
            c2.set_file_backing(ft)
            c2.request(torch.zeros((1, 26), dtype=torch.int32, device="cuda"))


@pytest.mark.parametrize("policy", POLICIES)
def test_batched_host_tier_flush_with_pinned_hits(E, orc, policy):
    """Found by tools/fuzz_cache.py: one table (every hit has the top priority, so the EvLFU flush fires), a 50-entry
    cache in front of host-memory tables, batches far larger than the cache.  Hits of the running batch are pinned,
    so the flush finds fewer victims than planned; the free stack must stay gap-free or live entries are handed out
    again and rows come back wrong."""
    rs = np.random.RandomState(0)
    n, d = 500, 16
    tab = rs.uniform(-1, 1, size=(n, d)).astype(np.float32)
    for host in (True, False):
        c = E.GpuCache("evlfu", 50, 1, d, 32, "python").set_batch_policy(policy)
        c.set_backing([torch.from_numpy(tab).pin_memory() if host else torch.from_numpy(tab).cuda()])
        for it in range(6):
            rq = rs.randint(0, n, size=(700, 1)).astype(np.int32)
            hit, out = c.lookup_batch(torch.from_numpy(rq).cuda())
            assert np.array_equal(out.cpu().numpy()[:, 0, :], tab[rq[:, 0]]), (host, it)
            x = torch.rand(700, d, device="cuda")
            hit, R = c.lookup_interact(torch.from_numpy(rq).cuda(), x)
            want = (torch.from_numpy(tab[rq[:, 0]]).cuda() * x).sum(1)
            torch.testing.assert_close(R[:, d], want, rtol=1e-5, atol=2e-6)
            st = c.batch_stats()
            dmp = c.batch_dump()
            assert st["size"] == len(dmp) <= 50 and len({int(r) for _, _, r in dmp}) == len(dmp)


def test_setassoc_flush_and_refusals(E, orc):
    """Set-associative policy: one table (every hit has the top priority, so the EvLFU flush fires -- EvLFU_C1.py:36-44), a
    small cache, batches far larger than it: rows exact, no duplicate keys, size <= capacity, flushes counted; host-memory
    tables and the two-tier lookup are refused (the policy reads its miss tier in place from HBM, single tier)."""
    rs = np.random.RandomState(0)
    n, d = 500, 16
    tab = rs.uniform(-1, 1, size=(n, d)).astype(np.float32)
    c = E.GpuCache("evlfu", 64, 1, d, 32, "python").set_batch_policy("setassoc")
    c.set_backing([torch.from_numpy(tab).cuda()])
    for it in range(8):
        rq = rs.randint(0, 80 if it % 2 else n, size=(700, 1)).astype(np.int32)
        hit, out = c.lookup_batch(torch.from_numpy(rq).cuda())
        assert np.array_equal(out.cpu().numpy()[:, 0, :], tab[rq[:, 0]]), it
        x = torch.rand(700, d, device="cuda")
        hit, R = c.lookup_interact(torch.from_numpy(rq).cuda(), x)
        want = (torch.from_numpy(tab[rq[:, 0]]).cuda() * x).sum(1)
        torch.testing.assert_close(R[:, d], want, rtol=1e-5, atol=2e-6)
        st, dmp = c.batch_stats(), c.batch_dump()
        assert st["size"] == len(dmp) <= 64 and len({int(r) for _, _, r in dmp}) == len(dmp)
        assert np.array_equal(np.bincount(dmp[:, 0], minlength=2), np.array(st["hist"]))
    # every resident key requested again: every entry reaches the top priority, the top bucket passes max_perfect
    # (0.95 x capacity) and the flush takes flush_n = int(0.3 x 64) + 1 entries away (EvLFU_C1.py:30,36-44)
    res = c.batch_dump()[:, 2].astype(np.int32)
    assert len(res) >= 61
    before = c.batch_stats()
    rq = np.resize(res, (700, 1))
    hit, out = c.lookup_batch(torch.from_numpy(rq).cuda())
    assert hit.all() and np.array_equal(out.cpu().numpy()[:, 0, :], tab[rq[:, 0]])
    st, dmp = c.batch_stats(), c.batch_dump()
    assert st["n_flush"] == before["n_flush"] + 1 and st["size"] == before["size"] - 20 == len(dmp)
    assert set(dmp[:, 2].tolist()) <= set(res.tolist())
    ch = E.GpuCache("evlfu", 64, 1, d, 32, "python").set_batch_policy("setassoc")
    ch.set_backing([torch.from_numpy(tab).pin_memory()])
    with pytest.raises(E.EvsError):
        ch.lookup_batch(torch.zeros((4, 1), dtype=torch.int32, device="cuda"))
    with pytest.raises(E.EvsError):
        E.GpuCache("evlfu", 4, 1, d, 32, "python").set_batch_policy("setassoc")   # fewer entries than one set has ways


def test_cache_wrappers_refuse_wrong_shapes(E, orc):
    """The kernels trust their shapes (a wrong one is an out-of-bounds device access): rows with the wrong number of
    columns, an `out` / `hit` of the wrong size or dtype are refused on the host, by both call paths."""
    tabs = orc.kaggle_tables([50] * 26, 1)
    c = E.GpuCache("evlfu", 100, 26, 36, 32)
    c.set_backing([torch.from_numpy(t).cuda() for t in tabs])
    good = torch.zeros((4, 26), dtype=torch.int32, device="cuda")
    x = torch.rand(4, 36, device="cuda")
    for bad_rows in (torch.zeros((4, 25), dtype=torch.int32, device="cuda"), torch.zeros((4, 26), dtype=torch.int64, device="cuda"),
                     torch.zeros((4, 52), dtype=torch.int32, device="cuda")[:, ::2]):
        for call in (lambda r: c.lookup_batch(r), lambda r: c.lookup_interact(r, x), lambda r: c.request(r)):
            with pytest.raises(ValueError):
                call(bad_rows)
    with pytest.raises(ValueError):
        c.lookup_batch(good, out=torch.empty((4, 26, 35), device="cuda"))
    with pytest.raises(ValueError):
        c.lookup_batch(good, hit=torch.empty((4, 26), dtype=torch.int32, device="cuda"))
    with pytest.raises(ValueError):
        c.lookup_interact(good, x, out=torch.empty((4, 36 + 350), device="cuda"))
    with pytest.raises(ValueError):
        c.lookup_interact(good, torch.rand(4, 32, device="cuda"))
    hit, R = c.lookup_interact(good, x)
    assert R.shape == (4, 36 + 351)


def test_batched_and_exact_paths_do_not_mix(E, orc):
    tabs = orc.kaggle_tables([50] * 26, 1)
    c = E.GpuCache("evlfu", 100, 26, 36, 32)
    c.set_backing([torch.from_numpy(t).cuda() for t in tabs])
    rows = torch.zeros((4, 26), dtype=torch.int32, device="cuda")
    c.lookup_batch(rows)
    with pytest.raises(E.EvsError):
        c.request(rows)


@pytest.mark.parametrize("policy", POLICIES1)
def test_cache_lookup_interact_equals_rows_then_interact(E, orc, policy):
    """evs_cache_lookup_interact (pointer-table + fused MFMA kernel) == interact_features over the table rows."""
    n_rows = [500, 7, 9000, 40, 2500, 3] + [100] * 20
    tabs = orc.kaggle_tables(n_rows, 5)
    dev = [torch.from_numpy(t).cuda() for t in tabs]
    c = E.GpuCache("evlfu", 2000, 26, 36, 32).set_batch_policy(policy)
    c.set_backing(dev)
    reqs = _zipf_requests(n_rows, 900, 4)
    r = torch.from_numpy(reqs).cuda()
    ev = E.EVTables(dev, 36, 32)
    for s in range(0, 900, 300):
        x = torch.rand(300, 36, device="cuda")
        hit, R = c.lookup_interact(r[s:s + 300].contiguous(), x)
        idx = r[s:s + 300].t().contiguous().to(torch.int64)
        off = torch.arange(300, device="cuda").repeat(26, 1)
        want = E.apply_emb_interact(x, off, idx, ev)
        assert torch.equal(R, want)
    assert c.batch_stats()["n_hits"] > 0


@pytest.mark.parametrize("policy", POLICIES1)
@pytest.mark.parametrize("thr", [23, 20])
def test_batched_two_tier_c1c2(E, orc, thr, policy):
    """Batched C1 (u8) + C2 (u4) lookup, snapshot semantics: tier flags = residency when the batch starts; rows at
    the precision of the tier that serves them (hit) or of the destination tier (miss, routed by the reference's
    rule on the snapshot); no key in both tiers; C2 untouched until C1 is full; histograms consistent.
    Set-associative tiers: "C1 is full" is a property of the key's own C1 set (no free way when the batch starts)."""
    from evstore_dlrm_amd import gpu_cache
    rs = np.random.RandomState(31)
    n, T, d = 300, 26, 36
    ws = [rs.uniform(-1, 1, size=(n, d)).astype(np.float32) for _ in range(T)]
    raw8 = [orc.encode_table(w, 8) for w in ws]
    raw4 = [orc.encode_table(w, 4) for w in ws]
    dec8 = [orc.decode(r, 8, d) for r in raw8]
    dec4 = [orc.decode(r, 4, d) for r in raw4]
    cap1, cap2 = 500, 900
    c1 = E.GpuCache("evlfu", cap1, T, d, 8, "cpp").set_batch_policy(policy)
    c2 = E.GpuCache("evlfu", cap2, T, d, 4, "cpp").set_batch_policy(policy)
    c1.set_backing([torch.from_numpy(r).cuda() for r in raw8])
    c2.set_backing([torch.from_numpy(r).cuda() for r in raw4])
    reqs = np.minimum(rs.zipf(1.25, size=(3000, T)) - 1, n - 1).astype(np.int32)
    r = torch.from_numpy(reqs).cuda()
    R1, R2 = {}, {}
    saw_c2 = False
    for s in range(0, len(reqs), 250):
        rq = reqs[s:s + 250]
        tier, out = gpu_cache.lookup_batch_c1c2(c1, c2, r[s:s + 250].contiguous(), threshold=thr)
        tier, out = tier.cpu().numpy(), out.cpu().numpy()
        c1_full = len(R1) >= cap1
        if policy == "setassoc":   # occupancy of every C1 set when the batch starts (the pair shares its set records)
            nset, ways1, ways2, bits = _sa_geom(cap1, [n] * T, cap2, [n] * T)
            occ = np.bincount(_sa_sets(list(R1), nset, [n] * T, bits), minlength=nset) if R1 else np.zeros(nset, int)
            set_of = _sa_sets([(k + 1, int(v)) for k in range(T) for v in range(n)], nset, [n] * T, bits).reshape(T, n)
        for b in range(len(rq)):
            in1 = np.array([(k + 1, int(rq[b, k])) in R1 for k in range(T)])
            in2 = np.array([(k + 1, int(rq[b, k])) in R2 for k in range(T)]) & ~in1
            assert np.array_equal(tier[b] == 1, in1) and np.array_equal(tier[b] == 2, in2), (s, b)
            agg = int(in1.sum() + in2.sum())
            for k in range(T):
                row = int(rq[b, k])
                if in1[k]:
                    want = dec8[k][row]
                elif in2[k]:
                    want = dec4[k][row]
                else:
                    full = c1_full if policy != "setassoc" else occ[set_of[k, row]] >= ways1
                    dest = 1 if not full else ((1 if k % 2 == 1 else 2) if agg < thr else 2)
                    want = dec8[k][row] if dest == 1 else dec4[k][row]
                assert np.array_equal(out[b, k].view(np.uint32), want.view(np.uint32)), (s, b, k)
        d1, d2 = c1.batch_dump(), c2.batch_dump()
        st1, st2 = c1.batch_stats(), c2.batch_stats()
        n1 = {(int(t), int(rw)): int(p) for p, t, rw in d1}
        n2 = {(int(t), int(rw)): int(p) for p, t, rw in d2}
        assert len(n1) == len(d1) == st1["size"] <= cap1 and len(n2) == len(d2) == st2["size"] <= cap2
        assert not (set(n1) & set(n2)), "a key lives in one tier"
        assert np.array_equal(np.bincount(d1[:, 0], minlength=T + 1), np.array(st1["hist"]))
        if len(d2):
            assert np.array_equal(np.bincount(d2[:, 0], minlength=T + 1), np.array(st2["hist"]))
        if not c1_full and policy != "setassoc":
            assert len(n2) == len(R2), "C2 is left alone while C1 has room"
        if policy == "setassoc" and len(R1) == 0:
            assert len(n2) == 0, "C2 is left alone while every C1 set has room"
        for key, p in n1.items():
            if key in R1:
                assert p >= R1[key]
        saw_c2 |= len(n2) > 0
        R1, R2 = n1, n2
    assert saw_c2 and (len(R1) == cap1 if policy != "setassoc" else 0.8 * cap1 < len(R1) <= cap1)
    assert c1.batch_stats()["n_requests"] == len(reqs)
    # the same lookup feeding the interaction: R == interact_features over the rows it served
    x = torch.rand(250, d, device="cuda")
    rows_buf = torch.empty((250, T, d), device="cuda")
    t2, R = gpu_cache.lookup_interact_c1c2(c1, c2, r[:250].contiguous(), x, threshold=thr, out=rows_buf, fused=False)
    assert torch.equal(R, E.interact_features(x, list(rows_buf.unbind(1)))) and torch.equal(R[:, :d], x)
    t2, rb = t2.cpu().numpy(), rows_buf.cpu().numpy()
    for b in range(0, 250, 7):
        for k in range(T):
            if t2[b, k]:
                want = (dec8 if t2[b, k] == 1 else dec4)[k][int(reqs[b, k])]
                assert np.array_equal(rb[b, k].view(np.uint32), want.view(np.uint32))


@pytest.mark.parametrize("policy", POLICIES1)
@pytest.mark.parametrize("codecs,d,T", [((8, 4), 36, 26), ((32, 8), 36, 26), ((32, 4), 32, 26), ((8, 4), 16, 26), ((8, 4), 32, 26),
                                        ((8, 4), 36, 9), ((8, 4), 36, 27)])
def test_two_tier_mixed_codec_interaction_consumer(E, orc, codecs, d, T, policy):
    """configs[4] end to end without the fp32 (B,T,d) rows: evs_cache_lookup_interact_c1c2 decodes every row from the
    precision of the tier that serves it inside the interaction kernel.  Which tier serves a MISS depends on the
    routing, so the tables hold only values every codec represents exactly (-1, 0, 1: u8 codes 0 / 127 / 254, u4 codes
    14 / 7 / 0): whatever tier serves a key, its row must decode to the same fp32 values -- and does only if address,
    row size and decoder all belong to that tier.  R against the oracle over the true rows; tier codes against the
    snapshot (a key reported in C1 / C2 was resident there before the call)."""
    from evstore_dlrm_amd import gpu_cache
    rs = np.random.RandomState(17)
    n_rows = [300] * T
    ws = [rs.randint(-1, 2, size=(n, d)).astype(np.float32) for n in n_rows]
    raws = {c: [orc.encode_table(w, c) for w in ws] for c in codecs}
    for c in codecs:
        assert all(np.array_equal(orc.decode(raws[c][k], c, d), ws[k]) for k in range(T))
    c1 = E.GpuCache("evlfu", 400, T, d, codecs[0], "cpp").set_batch_policy(policy)
    c2 = E.GpuCache("evlfu", 900, T, d, codecs[1], "cpp").set_batch_policy(policy)
    c1.set_backing([torch.from_numpy(a).cuda() for a in raws[codecs[0]]])
    c2.set_backing([torch.from_numpy(a).cuda() for a in raws[codecs[1]]])
    B = 333
    saw = set()
    for it in range(8):
        hot = rs.rand(B, T) < 0.7
        rq = np.where(hot, rs.randint(0, 12, size=(B, T)), rs.randint(0, 300, size=(B, T))).astype(np.int32)
        r = torch.from_numpy(rq).cuda()
        x = torch.from_numpy(rs.uniform(-1, 1, size=(B, d)).astype(np.float32)).cuda()
        in1 = {(int(t), int(rw)) for _, t, rw in c1.batch_dump()} if it else set()
        in2 = {(int(t), int(rw)) for _, t, rw in c2.batch_dump()} if it else set()
        tier, R = gpu_cache.lookup_interact_c1c2(c1, c2, r, x, itself=bool(it & 1))
        want = orc.interact_features(x.cpu().numpy(), [ws[k][rq[:, k]] for k in range(T)], bool(it & 1))
        np.testing.assert_allclose(R.cpu().numpy(), want, rtol=1e-5, atol=2e-6)
        assert torch.equal(R[:, :d], x)
        tn = tier.cpu().numpy()
        for b in range(0, B, 5):
            for k in range(T):
                key = (k + 1, int(rq[b, k]))
                assert tn[b, k] == (1 if key in in1 else (2 if key in in2 else 0)), (it, b, k)
        saw |= set(np.unique(tn).tolist())
    assert saw == {0, 1, 2}
    assert c1.batch_stats()["size"] <= 400 and c2.batch_stats()["size"] <= 900


@pytest.mark.parametrize("policy", POLICIES1)
@pytest.mark.parametrize("codec,d,T", [(8, 36, 26), (4, 36, 26), (16, 36, 26), (8, 16, 9), (4, 32, 17), (16, 16, 27)])
def test_single_tier_reduced_precision_interaction_consumer(E, orc, codec, d, T, policy):
    """A single reduced-precision tier (the reference's one-layer 16 / 8 / 4-bit builds) feeding the interaction: rows
    decoded inside the kernel, hits from the arena and misses from the backing table -- the same decoded row either way,
    so R must equal interact_features over the decoded table rows, whatever the cache holds."""
    rs = np.random.RandomState(31 + codec + d)
    n_rows = [200 + 7 * k for k in range(T)]
    raws = [orc.encode_table(rs.uniform(-1, 1, size=(n, d)).astype(np.float32), codec) for n in n_rows]
    dec = [orc.decode(r, codec, d) for r in raws]
    c = E.GpuCache("evlfu", 500, T, d, codec, "cpp").set_batch_policy(policy)
    c.set_backing([torch.from_numpy(a).cuda() for a in raws])
    B = 211
    saw_hit = saw_miss = False
    for it in range(6):
        hot = rs.rand(B, T) < 0.6
        rq = np.where(hot, rs.randint(0, 10, size=(B, T)), np.stack([rs.randint(0, n, size=B) for n in n_rows], 1)).astype(np.int32)
        x = torch.from_numpy(rs.uniform(-1, 1, size=(B, d)).astype(np.float32)).cuda()
        resident = {(int(t), int(rw)) for _, t, rw in c.batch_dump()} if it else set()
        hit, R = c.lookup_interact(torch.from_numpy(rq).cuda(), x, itself=bool(it & 1))
        want = orc.interact_features(x.cpu().numpy(), [dec[k][rq[:, k]] for k in range(T)], bool(it & 1))
        np.testing.assert_allclose(R.cpu().numpy(), want, rtol=1e-5, atol=2e-6)
        assert torch.equal(R[:, :d], x)
        h = hit.cpu().numpy()
        was = np.array([[(k + 1, int(rq[b, k])) in resident for k in range(T)] for b in range(B)])
        ev0 = ev1 if it else 0
        st = c.batch_stats()
        ev1 = st["n_evict"]
        if policy == "setassoc":
            # the update inside the launch (test_update_inside_the_probe_launch...): a flag says the row came from the cache, so the
            # key was resident when the batch arrived; a resident key read from its table was retired by this batch's own inserts
            assert not (h.astype(bool) & ~was).any()
            assert len({(k, int(rq[b, k])) for b, k in zip(*np.nonzero(was & ~h.astype(bool)))}) <= ev1 - ev0
        else:
            # the flags are the snapshot: residency when the batch arrived (the probe folded into the consumer or not)
            assert np.array_equal(h.astype(bool), was)
        saw_hit |= bool(h.any()); saw_miss |= bool((h == 0).any())
        keys = [(int(t), int(rw)) for _, t, rw in c.batch_dump()]
        assert len(set(keys)) == len(keys) == st["size"] <= 500
    assert saw_hit and saw_miss and c.batch_stats()["size"] <= 500
    assert c.batch_stats()["n_requests"] == 6 * B
    # what the launches left in the arena (the one-launch form copies the gathered rows' raw bytes there), bit for bit: the last
    # requests again through the rows-out path -- every resident key is served from the arena
    resident = {(int(t), int(rw)) for _, t, rw in c.batch_dump()}
    hit, out = c.lookup_batch(torch.from_numpy(rq).cuda())
    hit, out = hit.cpu().numpy().astype(bool), out.cpu().numpy()
    assert np.array_equal(hit, np.array([[(k + 1, int(rq[b, k])) in resident for k in range(T)] for b in range(B)])) and hit.any()
    for k in range(T):
        assert np.array_equal(out[:, k, :].view(np.uint32), dec[k][rq[:, k]].view(np.uint32))


@pytest.mark.parametrize("cap", [50, 64, 257])
def test_altkey_tier_ops_match_reference_driven_single_threaded(E, orc, cap):
    """a12, the pinnable part: APRX_EV's public methods (insert_altkey / get_altkey_str / set_recency_flag_c3 /
    evict_one_key) on the GPU tier against the COMPILED reference driven single-threaded (tests/golden/aprx_ops.npz):
    lookup results, alt keys and the final FIFO order bit-exact."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import make_golden as G
    g = load_golden("aprx_ops")
    alt = G.aprx_inputs()[0]
    t = E.GpuAltKeyTier(cap, [torch.from_numpy(a.view(np.int32).copy()).cuda() for a in alt])
    ops = torch.from_numpy(g["cap%d_ops" % cap].astype(np.int32)).cuda().contiguous()
    half = len(ops) // 2       # two calls: the state carries over
    res = torch.cat([t.apply_ops(ops[:half].contiguous()), t.apply_ops(ops[half:].contiguous())])
    st = t.stats()
    assert st["error"] == 0 and st["size"] <= cap
    np.testing.assert_array_equal(res.cpu().numpy().astype(np.uint32), g["cap%d_res" % cap])
    np.testing.assert_array_equal(t.queue().numpy(), g["cap%d_queue" % cap])


def test_three_tier_c1c2c3_vs_oracle(E, orc):
    """a12: request_to_c1_c2_c3 with the alt-key tier (deterministic re-specification) == the oracle:
    tier codes (incl. 3 = alt-key hit), values, both tiers' final lists, C3 counters."""
    from evstore_dlrm_amd import gpu_cache
    rs = np.random.RandomState(8)
    n = 300
    ws = [rs.uniform(-1, 1, size=(n, 36)).astype(np.float32) for _ in range(26)]
    raw8 = [orc.encode_table(w, 8) for w in ws]
    raw4 = [orc.encode_table(w, 4) for w in ws]
    dec8 = [orc.decode(r, 8, 36) for r in raw8]
    dec4 = [orc.decode(r, 4, 36) for r in raw4]
    alt = [(rs.randint(0, n, size=n) * 100 + (k + 1)).astype(np.uint32) for k in range(26)]
    cap1, cap2, cap3 = 400, 800, 200
    reqs = np.minimum(rs.zipf(1.3, size=(2500, 26)) - 1, n - 1).astype(np.int32)
    o = orc.C1C2C3(cap1, cap2, cap3, dec8, dec4, alt)
    want_tier, want_out = [], []
    for rq in reqs:
        t, out, _ = o.request(rq)
        want_tier.append(t.copy()); want_out.append(out.copy())
    c1 = E.GpuCache("evlfu", cap1, 26, 36, 8, "cpp")
    c2 = E.GpuCache("evlfu", cap2, 26, 36, 4, "cpp")
    c1.set_backing([torch.from_numpy(r).cuda() for r in raw8])
    c2.set_backing([torch.from_numpy(r).cuda() for r in raw4])
    c3 = gpu_cache.GpuAltKeyTier(cap3, [torch.from_numpy(a.view(np.int32)).cuda() for a in alt])
    r = torch.from_numpy(reqs).cuda()
    tiers, outs = [], []
    for s in range(0, len(reqs), 173):
        t, out = gpu_cache.request_c1c2c3(c1, c2, c3, r[s:s + 173].contiguous())
        tiers.append(t.cpu().numpy()); outs.append(out.cpu().numpy())
    tiers, outs = np.concatenate(tiers), np.concatenate(outs)
    assert np.array_equal(tiers, np.stack(want_tier))
    assert np.array_equal(outs.view(np.uint32), np.stack(want_out).view(np.uint32))
    np.testing.assert_array_equal(c1.dump(), o.c1.dump())
    np.testing.assert_array_equal(c2.dump(), o.c2.dump())
    st, so = c3.stats(), o.c3_state()
    assert st == so and st["n_hit"] > 50 and st["error"] == 0
    assert (tiers == 3).sum() == st["n_hit"]


def test_zz_cpp_socket_client_mirror(E, orc, tmp_path):
    """request_to_cpp_cache / print_n_reset_perfect_hit of the ctypes client mirror, in-process
    (the cache manager is a process-wide singleton: this test runs last in this file)."""
    from evstore_dlrm_amd.cache_algo import cpp_socket_client as cli
    t = load_golden("cache_traces")
    tabs = _tables(orc, t)
    (tmp_path / "ev-table" / "binary").mkdir(parents=True)
    for k, w in enumerate(tabs):
        w.tofile(tmp_path / "ev-table" / "binary" / ("ev-table-%d.bin" % (k + 1)))
    cli.init_ctypes_lib(ev_table_root=str(tmp_path), main_precision=32, total_size=300)
    o = orc.EvLFU(300, tabs, variant="cpp")
    perfect = 0
    for rq in t["requests"][:150]:
        ly = cli.request_to_cpp_cache([int(v) for v in rq])
        hit, vals = o.request(rq)
        perfect += int(hit.all())
        assert len(ly) == 26 and ly[0].shape == (1, 36)
        assert np.array_equal(np.stack([v.numpy()[0] for v in ly]), vals)
    assert int(cli.cache_manager_cpp.evs_manager_perfect_hit()) == perfect
    cli.print_n_reset_perfect_hit()
    assert int(cli.cache_manager_cpp.evs_manager_perfect_hit()) == 0
    # a5, the `cpp_algo` branch of the C1_C2 forks (dlrm_s_pytorch_C1_C2.py:247-249): the same client through
    # apply_emb_evstore; the forks count perfect hits in the library only
    from evstore_dlrm_amd import evstore_ops
    evstore_ops.cache_algo, evstore_ops.perfect_hit, evstore_ops.evstore_gpu_id = "cpp_algo", 0, 0
    perfect = 0
    for rq in t["requests"][150:260]:
        lS_i = torch.from_numpy(rq.astype(np.int64)).reshape(26, 1).cuda()
        ly = evstore_ops.apply_emb_evstore(None, lS_i, None, None, use_gpu=True, use_emb_cache=True)
        hit, vals = o.request(rq)
        perfect += int(hit.all())
        assert len(ly) == 26 and ly[0].shape == (1, 36) and ly[0].is_cuda
        assert np.array_equal(torch.cat(list(ly)).cpu().numpy(), vals)
    assert evstore_ops.perfect_hit == 0 and int(cli.cache_manager_cpp.evs_manager_perfect_hit()) == perfect
    evstore_ops.cache_algo = "evlfu"


@pytest.mark.parametrize("policy", POLICIES1)
@pytest.mark.parametrize("codecs,d", [((8, 4), 36), ((32, 8), 16)])
def test_batched_three_tier_c1c2c3(E, orc, codecs, d, policy):
    """f2 / configs[4]: the batched three-tier lookup (evs_cache_lookup_batch_c1c2c3 / _interact_c1c2c3), snapshot
    semantics.  Tier codes against the state before the call: 1 / 2 = resident in C1 / C2; 3 = a double miss whose key is
    a member of C3 and whose ALT row is resident in C1 (else C2) -- that row is served, at the precision of the tier
    holding it; 0 = miss, routed by the reference's rule with the alt hits counted in agg_hit.  C3 fills with keys the
    batch's policy update removed from C1 / C2, never beyond its capacity; recency flags only on keys that were served
    through their alt key.  The tables hold values every codec represents exactly (-1, 0, 1), so a row decoded with the
    wrong tier's codec or row size cannot come out right."""
    from evstore_dlrm_amd import gpu_cache
    rs = np.random.RandomState(41)
    T, n, thr = 26, 300, 23
    ws = [rs.randint(-1, 2, size=(n, d)).astype(np.float32) for _ in range(T)]
    raws = {c: [orc.encode_table(w, c) for w in ws] for c in codecs}
    for c in codecs:
        assert all(np.array_equal(orc.decode(raws[c][k], c, d), ws[k]) for k in range(T))
    # alt key of (table t, row r): the hot row r % 8 of table (t + 1) % T  (alt_row * 100 + alt_table_1based)
    alt = [np.array([(r % 8) * 100 + ((t + 1) % T + 1) for r in range(n)], dtype=np.uint32) for t in range(T)]
    cap1, cap2, cap3 = 400, 700, 800
    c1 = E.GpuCache("evlfu", cap1, T, d, codecs[0], "cpp").set_batch_policy(policy)
    c2 = E.GpuCache("evlfu", cap2, T, d, codecs[1], "cpp").set_batch_policy(policy)
    c1.set_backing([torch.from_numpy(a).cuda() for a in raws[codecs[0]]])
    c2.set_backing([torch.from_numpy(a).cuda() for a in raws[codecs[1]]])
    c3 = E.GpuAltKeyTier(cap3, [torch.from_numpy(a.view(np.int32)).cuda() for a in alt])
    B = 250
    R1, R2, M3 = {}, {}, set()
    ever_alt, removed, n3_total = set(), set(), 0
    saw = set()
    for it in range(14):
        hot = rs.rand(B, T) < 0.6
        rq = np.where(hot, rs.randint(0, 8, size=(B, T)), rs.randint(0, n, size=(B, T))).astype(np.int32)
        r = torch.from_numpy(rq).cuda()
        if it % 3 == 2:   # the interaction form: R over the rows the tier codes say were served
            x = torch.from_numpy(rs.uniform(-1, 1, size=(B, d)).astype(np.float32)).cuda()
            tier, Rm = gpu_cache.lookup_interact_c1c2c3(c1, c2, c3, r, x, threshold=thr, itself=bool(it & 1))
            out = None
        else:
            tier, out = gpu_cache.lookup_batch_c1c2c3(c1, c2, c3, r, threshold=thr)
            out = out.cpu().numpy()
        tier = tier.cpu().numpy()
        c1_full = len(R1) >= cap1
        served = np.empty((B, T, d), np.float32)
        for b in range(B):
            in1 = np.array([(k + 1, int(rq[b, k])) in R1 for k in range(T)])
            in2 = np.array([(k + 1, int(rq[b, k])) in R2 for k in range(T)]) & ~in1
            in3 = np.zeros(T, bool)
            for k in range(T):
                key = (k + 1, int(rq[b, k]))
                if not in1[k] and not in2[k] and key in M3:
                    a = int(alt[k][rq[b, k]])
                    akey = (a % 100, a // 100)
                    in3[k] = akey in R1 or akey in R2
            want_tier = np.where(in1, 1, np.where(in2, 2, np.where(in3, 3, 0)))
            assert np.array_equal(tier[b], want_tier), (it, b)
            for k in range(T):
                if in3[k]:
                    a = int(alt[k][rq[b, k]])
                    served[b, k] = ws[a % 100 - 1][a // 100]
                    ever_alt.add((k + 1, int(rq[b, k])))
                else:
                    served[b, k] = ws[k][rq[b, k]]
            n3_total += int(in3.sum())
        if out is not None:
            assert np.array_equal(out, served), it
        else:
            want = orc.interact_features(x.cpu().numpy(), [served[:, k, :] for k in range(T)], bool(it & 1))
            np.testing.assert_allclose(Rm.cpu().numpy(), want, rtol=1e-5, atol=2e-6)
        saw |= set(np.unique(tier).tolist())
        d1, d2 = c1.batch_dump(), c2.batch_dump()
        n1 = {(int(t), int(rw)) for _, t, rw in d1}
        n2 = {(int(t), int(rw)) for _, t, rw in d2}
        assert len(n1) == len(d1) <= cap1 and len(n2) == len(d2) <= cap2 and not (n1 & n2)
        removed |= (set(R1) - n1) | (set(R2) - n2)
        m3, st3 = c3.batch_dump()
        members = {(int(t), int(rw)) for t, rw, _ in m3}
        assert len(members) == len(m3) == st3["members"] <= st3["capacity"] <= cap3
        assert members <= removed, "C3 holds only keys the tiers gave up"
        assert {(int(t), int(rw)) for t, rw, f in m3 if f} <= ever_alt, "a recency flag means the key was served through its alt key"
        assert st3["n_hit"] == n3_total
        R1, R2, M3 = {k: 1 for k in n1}, {k: 1 for k in n2}, members
    assert saw == {0, 1, 2, 3} and len(M3) > 0
    # the two forms of the tier do not mix
    with pytest.raises(E.EvsError):
        gpu_cache.request_c1c2c3(E.GpuCache("evlfu", 50, T, d, codecs[0], "cpp"), E.GpuCache("evlfu", 50, T, d, codecs[1], "cpp"), c3,
                                 torch.zeros((1, T), dtype=torch.int32, device="cuda"))


@pytest.mark.parametrize("policy,cap,B", [("sampled", 9_000_000, 60_000), ("plan", 9_000_000, 60_000), ("sampled", 400_000, 70_001),
                                            ("plan", 400_000, 70_001), ("setassoc", 9_000_000, 60_000), ("setassoc", 400_000, 70_001)])
def test_batched_cache_above_the_slot_hint_range(E, orc, policy, cap, B):
    """A cache whose hash has more than 2^24 slots (9 M entries): the probe's slot hints no longer fit their 24 bits, the
    inserts walk from the key's home slot instead -- and a batch above 65 536 requests (the probe's blocks loop, their
    miss lists hold more than one round): same invariants (rows exact, hit flags = residency at batch start, no
    duplicate keys, size <= capacity, histogram consistent)."""
    T, d, n = 4, 16, 3_000_000
    g = torch.Generator(device="cuda").manual_seed(5)
    tabs = [torch.rand((n, d), generator=g, device="cuda") for _ in range(T)]
    c = E.GpuCache("evlfu", cap, T, d, 32, "python").set_batch_policy(policy)
    c.set_backing(tabs)
    rs = np.random.RandomState(9)
    resident = set()
    for it in range(4):
        hot = rs.rand(B, T) < 0.5
        rq = np.where(hot, rs.randint(0, 50_000, size=(B, T)), rs.randint(0, n, size=(B, T))).astype(np.int32)
        r = torch.from_numpy(rq).cuda()
        hit, out = c.lookup_batch(r)
        for k in range(T):
            assert torch.equal(out[:, k, :], tabs[k][r[:, k].long()]), (it, k)
        hit = hit.cpu().numpy().astype(bool)
        want = np.array([[(k + 1, int(rq[b, k])) in resident for k in range(T)] for b in range(0, B, 37)])
        assert np.array_equal(hit[::37], want), it
        dmp, st = c.batch_dump(), c.batch_stats()
        keys = {(int(t), int(rw)) for _, t, rw in dmp}
        assert len(keys) == len(dmp) == st["size"] <= cap
        assert np.array_equal(np.bincount(dmp[:, 0], minlength=T + 1), np.array(st["hist"]))
        if policy != "setassoc":   # (a set that takes more new keys than it has ways in one batch turns the rest away)
            assert all((k + 1, int(rq[b, k])) in keys for b in range(0, B, 101) for k in range(T) if not hit[b, k])
        resident = keys
