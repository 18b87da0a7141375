"""The host engine of the exact batch-1 policies (csrc/evs_hostcache.hip, through the C ABI) against the SAME golden
traces the GPU exact kernel is held to -- recorded from the reference's Python policies (cache_traces.npz), from its
Cython EvLFU.cpp compiled in place (cython_traces.npz), from its compiled cache manager (c1c2_ref.npz) and from APRX_EV
driven single-threaded (aprx_ops.npz) -- and against the oracle on random streams.  Bit-exact: hit flags, rows, final
list order, counters.  No GPU involved: these run in the CPU suite."""
import os
import sys

import numpy as np
import pytest

from conftest import load_golden
from oracle import oracle as orc

import evstore_dlrm_amd  # noqa: F401  (import shim)
from evstore_dlrm_amd import host_cache as H


def _tables(t):
    return orc.kaggle_tables([int(n) for n in t["n_rows"]], int(t["table_seed"]))


def _unpack(packed, n):
    return np.unpackbits(packed, axis=1)[:, :26].astype(bool)[:n]


def _run(policy, cap, raws, reqs, chunk, approx=-1, variant="python", codec=32):
    c = H.HostCache(policy, cap, 26, 36, codec, variant).set_backing(raws)
    hits, outs = [], []
    for s in range(0, len(reqs), chunk):
        h, o = c.request(reqs[s:s + chunk], approx)
        hits.append(h.astype(bool))
        outs.append(o)
    return c, np.concatenate(hits), np.concatenate(outs)


@pytest.mark.parametrize("cap,chunk", [(64, 1), (300, 7), (768, 100), (2000, 1500), (79, 1), (80, 33), (82, 1200)])
def test_evlfu_trace(cap, chunk):
    t = load_golden("cache_traces")
    tabs = _tables(t)
    reqs = t["requests_flush"] if cap in (79, 80, 82) else t["requests"]
    c, hits, outs = _run("evlfu", cap, tabs, reqs, chunk)
    assert np.array_equal(hits, _unpack(t["evlfu_cap%d_hits" % cap], len(reqs)))
    for k in range(26):
        assert np.array_equal(outs[:, k, :], tabs[k][reqs[:, k]])
    np.testing.assert_array_equal(c.dump(), t["evlfu_cap%d_final_buckets" % cap])
    st = c.stats()
    assert [st["min_c1"], st["n_perfect"], st["size"], st["n_flush"]] == list(t["evlfu_cap%d_state" % cap])
    assert st["n_requests"] == len(reqs) and st["n_hits"] == int(hits.sum()) and st["n_perfect_hits"] == int(hits.all(1).sum())
    if cap in (79, 80, 82):
        assert st["n_flush"] >= 1


@pytest.mark.parametrize("stream,cap,chunk", [("main", 52, 1), ("main", 64, 5), ("main", 78, 1), ("main", 300, 64), ("main", 768, 100),
                                              ("flush", 52, 1), ("flush", 64, 3), ("flush", 78, 33), ("flush", 300, 1200), ("flush", 768, 1)])
def test_evlfu_cython_variant_trace(stream, cap, chunk):
    t = load_golden("cython_traces")
    tabs = _tables(t)
    reqs = t["requests_flush"] if stream == "flush" else t["requests"]
    tag = "cython_%s_cap%d" % (stream, cap)
    c, hits, outs = _run("evlfu", cap, tabs, reqs, chunk, variant="cython")
    assert np.array_equal(hits, _unpack(t[tag + "_hits"], len(reqs)))
    for k in range(26):
        assert np.array_equal(outs[:, k, :], tabs[k][reqs[:, k]])
    np.testing.assert_array_equal(c.dump(), t[tag + "_final_buckets"])
    st = c.stats()
    assert [st["min_c1"], st["n_perfect"], st["size"], st["n_flush"]] == list(t[tag + "_state"])


def test_evlfu_approx_mode_trace_and_rows():
    """approx mode (EvLFU_C1.py:122-125,142-152): hit flags + final lists = the reference's trace; every row = the
    oracle's (a miss above the threshold is served the previous hit's vector, zeros when there was none)."""
    t = load_golden("cache_traces")
    tabs = _tables(t)
    reqs = t["requests"]
    c, hits, outs = _run("evlfu", 768, tabs, reqs, 50, approx=20)
    assert np.array_equal(hits, _unpack(t["evlfu_cap768_approx20_hits"], len(reqs)))
    np.testing.assert_array_equal(c.dump(), t["evlfu_cap768_approx20_final_buckets"])
    o = orc.EvLFU(768, tabs)
    for i, rq in enumerate(reqs):
        _, vals = o.request(rq, approx_thres=20)
        assert np.array_equal(outs[i].view(np.uint32), vals.view(np.uint32)), i


@pytest.mark.parametrize("cap", [64, 300, 768, 2000, 80])
def test_lru_trace(cap):
    t = load_golden("cache_traces")
    tabs = _tables(t)
    reqs = t["requests_flush"] if cap == 80 else t["requests"]
    c, hits, outs = _run("lru", cap, tabs, reqs, 97)
    assert np.array_equal(hits, _unpack(t["lru_cap%d_hits" % cap], len(reqs)))
    for k in range(26):
        assert np.array_equal(outs[:, k, :], tabs[k][reqs[:, k]])
    np.testing.assert_array_equal(c.dump()[:, 1:], t["lru_cap%d_final_order" % cap])


@pytest.mark.parametrize("cap", [64, 300, 768, 2000, 80])
def test_lfu_trace(cap):
    t = load_golden("cache_traces")
    tabs = _tables(t)
    reqs = t["requests_flush"] if cap == 80 else t["requests"]
    c, hits, outs = _run("lfu", cap, tabs, reqs, 1)
    assert np.array_equal(hits, _unpack(t["lfu_cap%d_hits" % cap], len(reqs)))
    for k in range(26):
        assert np.array_equal(outs[:, k, :], tabs[k][reqs[:, k]])
    np.testing.assert_array_equal(c.dump(), t["lfu_cap%d_final_freq" % cap])


@pytest.mark.parametrize("codec", [16, 8, 4])
def test_reduced_precision_tier_decodes_like_the_reference(codec):
    """a tier holding 16 / 8 / 4-bit rows: the policy is precision-agnostic (same hit trace), rows decode bit-exactly
    like the compiled reference decoders (the oracle's are pinned to them exhaustively, codec_tables.npz)."""
    t = load_golden("cache_traces")
    tabs = _tables(t)
    raws = [orc.encode_table(np.clip(w * 8, -1, 1), codec) for w in tabs]
    reqs = t["requests"][:500]
    c, hits, outs = _run("evlfu", 300, raws, reqs, 40, codec=codec)
    assert np.array_equal(hits, _unpack(t["evlfu_cap300_hits"], 1500)[:500])
    for k in range(26):
        want = orc.decode(raws[k][reqs[:, k]], codec, 36)
        assert np.array_equal(outs[:, k, :].view(np.uint32), want.view(np.uint32))


def test_exhaustive_decoders():
    """every u8 code, every packed u4 byte, every u16 code through the host engine's decoders = the compiled reference's"""
    g = load_golden("codec_tables")
    for codec, n_codes in ((8, 256), (4, 256), (16, 65536)):
        if codec == 16:
            raw = np.arange(65536, dtype=np.uint16).reshape(-1, 16).view(np.uint8)   # 4096 rows of d = 16
            d = 16
        elif codec == 8:
            raw = np.arange(256, dtype=np.uint8).reshape(-1, 16)
            d = 16
        else:
            raw = np.arange(256, dtype=np.uint8).reshape(-1, 8)   # 8 bytes = 16 elements per row
            d = 16
        n = raw.shape[0]
        c = H.HostCache("lru", 8, 1, d, codec).set_backing([np.ascontiguousarray(raw)])
        _, out = c.request(np.arange(n, dtype=np.int32).reshape(n, 1))
        got = out.reshape(-1)
        want = orc.decode(np.ascontiguousarray(raw), codec, d).reshape(-1)
        both_nan = np.isnan(got) & np.isnan(want)
        assert np.array_equal(got.view(np.uint32)[~both_nan], want.view(np.uint32)[~both_nan]), codec
        key = {8: "u8", 4: "u4", 16: "u16"}[codec]
        if key in g.files:
            ref = g[key].reshape(-1)
            m = ~(np.isnan(ref) | np.isnan(got[:ref.size]))
            assert np.array_equal(got[:ref.size][m].view(np.uint32), ref[m].view(np.uint32))


def _c1c2_tables():
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import make_golden as G
    _, tabs = G.c1c2_tables(orc)
    return [t[0] for t in tabs], [t[1] for t in tabs]


def test_c1c2_matches_oracle_and_compiled_reference():
    """request_to_c1_c2 on the host: every tier code and every row = the oracle's (bit for bit, 11 000 requests through
    the fill of C1 and beyond), hence the compiled reference's serving precision as the oracle test pins it."""
    g = load_golden("c1c2_ref")
    raw8, raw4 = _c1c2_tables()
    dec8, dec4 = [orc.decode(r, 8, 36) for r in raw8], [orc.decode(r, 4, 36) for r in raw4]
    reqs = g["requests"]
    c1 = H.HostCache("evlfu", int(g["cap_c1"]), 26, 36, 8, "cpp").set_backing(raw8)
    c2 = H.HostCache("evlfu", int(g["cap_c2"]), 26, 36, 4, "cpp").set_backing(raw4)
    o = orc.C1C2(int(g["cap_c1"]), int(g["cap_c2"]), dec8, dec4)
    tier, out = H.request_c1c2(c1, c2, reqs)
    perfect = 0
    for i, rq in enumerate(reqs):
        t_o, v_o, p = o.request(rq)
        perfect += p
        assert np.array_equal(tier[i], t_o), i
        assert np.array_equal(out[i].view(np.uint32), v_o.view(np.uint32)), i
    assert c1.stats()["n_perfect_hits"] == perfect
    np.testing.assert_array_equal(c1.dump(), o.c1.dump())
    np.testing.assert_array_equal(c2.dump(), o.c2.dump())
    ref = g["served_bits"]
    first_ref = int(np.argmax((ref == 4).any(1)))
    served = np.where(tier == 2, 4, 8)
    served[(tier == 0)] = 0
    # until C1 fills every row is served at the main precision, like the compiled reference
    assert (tier[:first_ref] != 2).all() and first_ref > 9000


@pytest.mark.parametrize("caps", [(120, 200, 50), (300, 500, 64), (60, 90, 257)])
def test_c1c2c3_matches_oracle(caps):
    """request_to_c1_c2_c3 with small tiers so that evictions feed the alt-key tier and alt rows are served: tier codes,
    rows, the three tiers' final state = the oracle's."""
    rs = np.random.RandomState(sum(caps))
    T, n, d = 26, 400, 36
    ws = [rs.uniform(-1, 1, size=(n, d)).astype(np.float32) for _ in range(T)]
    raw8, raw4 = [orc.encode_table(w, 8) for w in ws], [orc.encode_table(w, 4) for w in ws]
    dec8, dec4 = [orc.decode(r, 8, d) for r in raw8], [orc.decode(r, 4, d) for r in raw4]
    alt = [((rs.randint(0, 12, size=n)).astype(np.uint32) * 100 + (rs.randint(0, T, size=n) + 1).astype(np.uint32)) for _ in range(T)]
    z = rs.zipf(1.3, size=(3000, T)) - 1
    reqs = np.minimum(z, n - 1).astype(np.int32)
    c1 = H.HostCache("evlfu", caps[0], T, d, 8, "cpp").set_backing(raw8)
    c2 = H.HostCache("evlfu", caps[1], T, d, 4, "cpp").set_backing(raw4)
    c3 = H.HostAltKeyTier(caps[2], alt)
    o = orc.C1C2C3(caps[0], caps[1], caps[2], dec8, dec4, alt)
    for s in range(0, len(reqs), 250):
        tier, out = H.request_c1c2c3(c1, c2, c3, reqs[s:s + 250])
        for i in range(tier.shape[0]):
            t_o, v_o, _ = o.request(reqs[s + i])
            assert np.array_equal(tier[i], t_o), s + i
            assert np.array_equal(out[i].view(np.uint32), v_o.view(np.uint32)), s + i
    assert (tier == 3).any() or c3.stats()["n_hit"] > 0
    assert c3.stats() == o.c3_state()
    np.testing.assert_array_equal(c1.dump(), o.c1.dump())
    np.testing.assert_array_equal(c2.dump(), o.c2.dump())


def _aprx_alt_tables():
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import make_golden as G
    return G.aprx_inputs()[0]


@pytest.mark.parametrize("cap", [50, 64, 257])
def test_altkey_tier_ops_match_reference_driven_single_threaded(cap):
    g = load_golden("aprx_ops")
    t = H.HostAltKeyTier(cap, _aprx_alt_tables())
    res = t.apply_ops(g["cap%d_ops" % cap])
    assert t.stats()["error"] == 0
    np.testing.assert_array_equal(res, g["cap%d_res" % cap])
    np.testing.assert_array_equal(t.queue(), g["cap%d_queue" % cap])
    assert t.stats()["size"] <= cap


@pytest.mark.parametrize("policy", ["evlfu", "lru", "lfu"])
@pytest.mark.parametrize("seed", [1, 2, 3])
def test_random_streams_vs_oracle(policy, seed):
    """tiny caches (capacity below one request's keys, equal to it, a few requests' worth), few tables, all variants"""
    rs = np.random.RandomState(100 * seed + len(policy))
    T = int(rs.choice([1, 3, 26, 40]))
    d = int(rs.choice([4, 16, 36]))
    n_rows = [int(rs.choice([1, 2, 7, 50, 300])) for _ in range(T)]
    tabs = [rs.uniform(-1, 1, size=(n, d)).astype(np.float32) for n in n_rows]
    cap = int(rs.choice([1, 2, T, T + 1, 3 * T, 10 * T]))
    variant = str(rs.choice(["python", "cpp", "cython"])) if policy == "evlfu" else "python"
    reqs = np.stack([np.minimum(rs.zipf(1.4, size=2000) - 1, n - 1) for n in n_rows], 1).astype(np.int32)
    o = {"evlfu": lambda: orc.EvLFU(cap, tabs, d, variant), "lru": lambda: orc.LRU(cap, tabs, d),
         "lfu": lambda: orc.LFU(cap, tabs, d)}[policy]()
    c = H.HostCache(policy, cap, T, d, 32, variant).set_backing(tabs)
    for i, rq in enumerate(reqs):
        try:
            h_o, v_o = o.request(rq)
        except RuntimeError:
            # the Python reference raises here (flush of an empty top bucket with cap < T): the engine reports ESTATE
            with pytest.raises(evstore_dlrm_amd.EvsError):
                c.request(rq.reshape(1, -1))
            return
        h, v = c.request(rq.reshape(1, -1))
        assert np.array_equal(h[0].astype(bool), np.asarray(h_o).astype(bool)), (i, cap, T, variant)
        assert np.array_equal(v[0].view(np.uint32), np.asarray(v_o).view(np.uint32)), i
    d_o = o.dump()
    d_c = c.dump()
    np.testing.assert_array_equal(d_c if policy != "lru" else d_c[:, 1:], d_o)


def test_errors():
    tabs = [np.zeros((5, 36), np.float32)] * 26
    c = H.HostCache("evlfu", 10).set_backing(tabs)
    bad = np.zeros((1, 26), np.int32)
    bad[0, 3] = 5
    with pytest.raises(evstore_dlrm_amd.EvsError) as e:
        c.request(bad)
    assert e.value.code == evstore_dlrm_amd._lib.EVS_EINDEX
    with pytest.raises(evstore_dlrm_amd.EvsError):
        H.HostCache("evlfu", 10).request(np.zeros((1, 26), np.int32))   # no backing
    with pytest.raises(evstore_dlrm_amd.EvsError):
        H.HostAltKeyTier(49, [np.zeros(4, np.uint32)] * 26)           # aprx_embedding.cpp:33 assert
    with pytest.raises(evstore_dlrm_amd.EvsError):
        H.HostCache("evlfu", 0)


@pytest.mark.parametrize("storage", ["DUMMY", "FILEPY", "MMAPFILEPY"])
def test_plugin_surface_on_the_host_engine(storage, tmp_path):
    """apply_emb_evstore + the cache modules + the storage manager as dlrm_s_pytorch_C1.py drives them, tables in host
    storage -> the host engine (engine="auto"): the reference's hit trace, perfect-hit count and rows; all four module
    surfaces (EvLFU_C1, LRU, LFU, the Cython EvLFU)."""
    import torch
    from evstore_dlrm_amd import evstore_ops
    from evstore_dlrm_amd.cache_algo import EvLFU, EvLFU_C1, LFU, LRU
    from evstore_dlrm_amd.emb_storage import storage_manager as sm
    t = load_golden("cache_traces")
    tabs = _tables(t)
    (tmp_path / "binary").mkdir()
    for k, w in enumerate(tabs):
        w.tofile(tmp_path / "binary" / ("ev-table-%d.bin" % (k + 1)))
    sm.storage_type, sm.ev_precs = getattr(sm.EmbStorage, storage), 32
    sm.load_ev_table_into_emb_stor(str(tmp_path))
    n = 400
    reqs = t["requests"]
    for algo, mod, init, tag in (("evlfu", EvLFU_C1, lambda: EvLFU_C1.init(768), "evlfu_cap768_hits"),
                                 ("lru", LRU, lambda: LRU.init(300), "lru_cap300_hits"),
                                 ("lfu", LFU, lambda: LFU.init(64), "lfu_cap64_hits")):
        init()
        evstore_ops.cache_algo, evstore_ops.perfect_hit = algo, 0
        want = _unpack(t[tag], 1500)
        for i in range(n):
            lS_i = torch.from_numpy(reqs[i].astype(np.int64)).reshape(26, 1)
            ly = evstore_ops.apply_emb_evstore(None, lS_i, None, None, use_gpu=False, use_emb_cache=True)
            assert len(ly) == 26 and ly[0].shape == (1, 36) and ly[0].requires_grad and not ly[0].is_cuda
            if i % 37 == 0:
                for k in range(26):
                    assert np.array_equal(ly[k].detach().numpy()[0], tabs[k][reqs[i][k]])
        assert mod._m.engine == "host"
        assert evstore_ops.perfect_hit == int(want[:n].all(1).sum())
        assert mod._m.cache.stats()["n_hits"] == int(want[:n].sum())
    tc = load_golden("cython_traces")
    assert np.array_equal(tc["requests"], reqs)
    EvLFU.cinit(78)
    EvLFU.cload_ev_tables()
    evstore_ops.cache_algo, evstore_ops.perfect_hit = "evlfu_cython", 0
    wantc = _unpack(tc["cython_main_cap78_hits"], 1500)
    for i in range(n):
        lS_i = torch.from_numpy(reqs[i].astype(np.int64)).reshape(26, 1)
        ly = evstore_ops.apply_emb_evstore(None, lS_i, None, None, use_gpu=False, use_emb_cache=True)
        assert len(ly) == 26 and ly[0].shape == (1, 36)
    EvLFU.cclose_ev_tables()
    assert EvLFU.stats()["n_hits"] == int(wantc[:n].sum()) and evstore_ops.perfect_hit == int(wantc[:n].all(1).sum())
    _, ly = sm.request_to_emb_storage([int(v) for v in reqs[0]])
    assert np.array_equal(ly[3].detach().numpy()[0], tabs[3][reqs[0][3]])
    evstore_ops.cache_algo = "evlfu"
    sm.close_any_db_conn()


def test_altkey_tables_are_checked_against_the_tier_tables():
    """the three-tier request indexes alt_tables[t][row] with rows of the TIERS' tables -- a short alt-key table
    (a truncated file) or none at all must be an error at the boundary, not a read past the end."""
    E = evstore_dlrm_amd
    T, n, d = 26, 40, 36
    ws = [np.zeros((n, d), np.float32)] * T
    raw8, raw4 = [orc.encode_table(w, 8) for w in ws], [orc.encode_table(w, 4) for w in ws]
    c1 = H.HostCache("evlfu", 50, T, d, 8, "cpp").set_backing(raw8)
    c2 = H.HostCache("evlfu", 50, T, d, 4, "cpp").set_backing(raw4)
    rq = np.zeros((1, T), np.int32)
    short = [np.ones(n if k != 7 else n - 1, np.uint32) * 101 for k in range(T)]   # table 8 one row short
    with pytest.raises(E.EvsError) as e:
        H.request_c1c2c3(c1, c2, H.HostAltKeyTier(64, short), rq)
    assert e.value.code == E._lib.EVS_EINVAL and "alt-key table 8" in str(e.value)
    # a tier whose alt-key tables were never set (created through the C ABI without evs_hostaprx_set_altkeys)
    import ctypes as C
    h = C.c_void_p()
    E._lib.check(E._lib.lib().evs_hostaprx_create(C.byref(h), 64, T))
    try:
        out, tier = np.empty((1, T, d), np.float32), np.empty((1, T), np.uint8)
        rc = E._lib.lib().evs_hostcache_request_c1c2c3(c1._h, c2._h, h, 1, rq.ctypes.data, out.ctypes.data, tier.ctypes.data, 23)
        assert rc == E._lib.EVS_ESTATE
    finally:
        E._lib.lib().evs_hostaprx_destroy(h)
    full = [np.ones(n, np.uint32) * 101 for _ in range(T)]
    t_, _ = H.request_c1c2c3(c1, c2, H.HostAltKeyTier(64, full), rq)
    assert t_.shape == (1, T)


def test_altkey_fifo_grows_like_the_unbounded_queue_it_stands_for():
    """re-inserting keys that are already members pushes a duplicate queue entry each time and pops nothing while
    the tier has room (aprx_embedding.cpp:278-288: std::queue, unbounded).  The ring used to latch an error once 4 cap + 64
    entries were queued; it doubles now, order kept: every push is in the dump, front to back."""
    cap, T = 60, 26
    alt = [np.arange(50, dtype=np.uint32) * 100 + (k + 1) for k in range(T)]
    t = H.HostAltKeyTier(cap, alt)
    keys = [(1 + i % 5, i % 7) for i in range(1500)]               # 35 distinct keys, 1 500 inserts: > 4 * 60 + 64
    ops = np.array([[0, tb, r] for tb, r in keys], np.int32)
    res = t.apply_ops(ops)
    st = t.stats()
    assert st["error"] == 0 and st["size"] == len(set(keys)) <= cap
    np.testing.assert_array_equal(res, [r * 100 + tb for tb, r in keys])
    q = t.queue()
    assert q.shape == (1500, 2) and [tuple(int(v) for v in row) for row in q] == keys
    # ... and the tier keeps working after the growth: eviction pops from the front (the first entry of a key is its turn)
    t.apply_ops(np.array([[3, 1, 0]] * 3, np.int32))
    assert t.stats()["error"] == 0 and t.stats()["size"] == len(set(keys)) - 3


def test_module_stats_before_the_first_request():
    """init() only records its arguments (the engine is chosen when the tables are known); stats() right after
    cinit() / init() must bind the engine instead of dereferencing None."""
    from evstore_dlrm_amd.cache_algo import EvLFU, EvLFU_C1
    from evstore_dlrm_amd.emb_storage import storage_manager as sm
    import torch
    tabs = [torch.zeros((9, 36)) for _ in range(26)]
    sm.use_device_tables(tabs, 32, storage=sm.EmbStorage.DUMMY)
    try:
        EvLFU.cinit(32, engine="host")
        assert EvLFU.stats()["size"] == 0
        EvLFU_C1.init(32, engine="host")
        assert EvLFU_C1.stats()["n_requests"] == 0
    finally:
        sm.close_any_db_conn()
