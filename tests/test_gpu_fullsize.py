"""GPU, FULL SIZE: exactly what bench.py times for BASELINE configs[2] (EvLFU C1 at 10 % of the Kaggle rows, the batched
snapshot lookup with the probe folded into the interaction kernel) and configs[4] (u8 C1 + u4 C2 + the alt-key tier at the
48-48-4 split), at the bench's shape -- Kaggle cardinalities (33.76 M rows), 3 376 257 entries, B = 16 384, T = 26, d = 36,
both policy updates -- and a Terabyte-cardinality property test (d = 64 and 128, 40 M-row tables, byte offsets > 4 GB).

No CPU reference finishes at these sizes row by row, so the checks are: residency-at-batch-start hit / tier flags (from
the cache's own dump before the call), R bit-equal to the uncached fused launch over the same tables (a cache serves exact
copies), R of sampled samples against the oracle, no duplicate keys, size <= capacity -- and the hit RATE against the
sequential oracle (cache_algo/EvLFU_C1.py restated) replaying the same Zipf stream one request at a time."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

RTOL = 1e-5
B = 16384
D = 36
FILL, CHECKED = 60, 10       # batches that fill the 10 % cache (bench: warmup = 60), batches checked afterwards
RATE_BAND = 0.01             # |batched hit rate - sequential oracle hit rate| on the same batches (measured gaps: <= 0.006)


@pytest.fixture(scope="module")
def E():
    import evstore_dlrm_amd as E
    assert torch.cuda.is_available(), "gpu tests need a GPU"
    E._lib.lib()
    return E


@pytest.fixture(scope="module")
def orc():
    from oracle import oracle
    return oracle


@pytest.fixture(scope="module")
def kaggle(E):
    """the bench's tables and its configs[2] request stream (bench.cache_tier_section: seed 3, Zipf 0.75)"""
    import bench
    ln = bench.KAGGLE_LN
    ev = bench.make_tables(ln, D)
    batches = bench.make_batches(ln, B, FILL + CHECKED, seed=3, device="cuda", dist="zipf", alpha=0.75)
    rows = [b[1].t().contiguous().to(torch.int32) for b in batches]
    yield {"ln": ln, "ev": ev, "batches": batches, "rows": rows, "cap": int(0.10 * sum(ln))}
    del ev
    torch.cuda.empty_cache()


@pytest.fixture(scope="module")
def oracle_rates(kaggle, orc):
    """the sequential oracle over the same stream: per-batch hit counts of the CHECKED batches (policy-independent)"""
    ev, ln = kaggle["ev"], kaggle["ln"]
    tabs = [ev.fp32_view(k).cpu().numpy() for k in range(len(ln))]
    o = orc.EvLFU(kaggle["cap"], tabs, D, "python")
    hits = []
    for i, r in enumerate(kaggle["rows"]):
        hr = r.cpu().numpy()
        n = 0
        for q in hr:
            n += int(o.request(q)[0].sum())
        if i >= FILL:
            hits.append(n)
    st = o.state()
    del o, tabs
    return {"hits": hits, "size": st["size"]}


def _keys(dump):
    return (dump[:, 1].astype(np.int64) << 32) | dump[:, 2].astype(np.int64)


def _query_keys(rows_np):
    t = np.arange(1, rows_np.shape[1] + 1, dtype=np.int64)[None, :]
    return (t << 32) | rows_np.astype(np.int64)


@pytest.mark.parametrize("policy", ["sampled", "plan", "setassoc", "setassoc-two-launches", "setassoc-u8"])
def test_bench_cache_tier_workload_at_full_size(E, orc, kaggle, oracle_rates, policy, monkeypatch):
    """"setassoc" = the library's default for this workload: the policy update inside the probe + interaction launch (round 5);
    "setassoc-two-launches" = EVS_CACHE_INLINE=0, the update as a launch of its own (strict snapshot flags); "setassoc-u8" = the
    same one-launch form of a single u8 tier over the tables encoded to 8 bits (the reference's one-layer evlfu_8 build)."""
    ev, ln, cap = kaggle["ev"], kaggle["ln"], kaggle["cap"]
    T = len(ln)
    assert cap == 3376257
    inline = policy in ("setassoc", "setassoc-u8")
    codec = 8 if policy.endswith("-u8") else 32
    if codec != 32:
        ev = ev.encode(codec)
    monkeypatch.setenv("EVS_CACHE_INLINE", "1" if inline else "0")   # (read when the cache takes its first batch)
    policy = policy.split("-")[0]
    cache = E.GpuCache("evlfu", cap, T, D, codec, "python", "cuda").set_batch_policy(policy)
    cache.set_backing(ev)
    g = torch.Generator(device="cuda").manual_seed(17)
    x = torch.rand((B, D), device="cuda", generator=g)
    F = T + 1
    out = torch.empty((B, D + F * (F - 1) // 2), device="cuda")
    hit = torch.empty((B, T), dtype=torch.uint8, device="cuda")
    for i in range(FILL):
        cache.lookup_interact(kaggle["rows"][i], x, out=out, hit=hit)      # the bench's step
    st = cache.batch_stats()
    # (the set-associative policy fills set by set: 94 % after the 60 fill batches, evicting in the sets that are full)
    assert st["size"] > (0.90 if policy == "setassoc" else 0.95) * cap, "the fill phase must leave the cache at capacity (the bench times it evicting)"
    rs = np.random.RandomState(5)
    hits_batched = []
    for i in range(FILL, FILL + CHECKED):
        before = np.sort(_keys(cache.batch_dump()))
        assert before.size == np.unique(before).size                       # no duplicate keys
        rows = kaggle["rows"][i]
        h, R = cache.lookup_interact(rows, x, out=out, hit=hit)
        torch.cuda.synchronize()
        rows_np = rows.cpu().numpy()
        want_hit = np.isin(_query_keys(rows_np), before)                  # residency when the batch starts
        got_hit = h.cpu().numpy().astype(bool)
        if inline:
            # the update runs inside the launch: a flag says "served from the cache" -- never for a key that was not resident
            # when the batch arrived; a resident key reported as a miss was retired by one of this batch's own inserts
            assert not (got_hit & ~want_hit).any()
            lost = np.unique(_query_keys(rows_np)[want_hit & ~got_hit]).size
            ev_before = st["n_evict"]
            assert lost <= cache.batch_stats()["n_evict"] - ev_before, "more resident keys reported as misses than this batch evicted"
            assert lost <= 0.002 * B * T
        else:
            assert np.array_equal(got_hit, want_hit), "hit flags differ from the snapshot at %d positions" % int((got_hit != want_hit).sum())
        hits_batched.append(int(got_hit.sum()))
        # a cache serves exact copies of the table rows: R = the uncached fused launch over the same tables, bit for bit
        off, idx = kaggle["batches"][i]
        R_ref = E.apply_emb_interact(x, off, idx, ev, one_index_per_bag=True)
        assert torch.equal(R, R_ref)
        # ... and sampled samples against the oracle
        sel = np.sort(rs.choice(B, 48, replace=False))
        sel_t = torch.from_numpy(sel).cuda()
        ly = [ev.fp32_view(k)[idx[k][sel_t]].cpu().numpy() if codec == 32 else
              orc.decode(ev.raw[k][idx[k][sel_t]].cpu().numpy(), codec, D) for k in range(T)]
        want = orc.interact_features(x[sel_t].cpu().numpy(), ly)
        np.testing.assert_allclose(R[sel_t].cpu().numpy(), want, rtol=RTOL, atol=2e-6)
        st = cache.batch_stats()
        after = _keys(cache.batch_dump())
        assert after.size == st["size"] <= cap and np.unique(after).size == after.size
    rate_b = sum(hits_batched) / (CHECKED * B * T)
    rate_o = sum(oracle_rates["hits"]) / (CHECKED * B * T)
    print("configs[2] full size, %s: batched hit rate %.4f, sequential oracle %.4f" % (policy, rate_b, rate_o))
    assert abs(rate_b - rate_o) <= RATE_BAND, (policy, rate_b, rate_o)
    del cache
    torch.cuda.empty_cache()


def _exact_tables(ln, d, seed):
    """tables whose values every codec represents exactly (-1, 0, 1): what a key is served does not depend on which
    tier the batched (racy) routing put it in"""
    g = torch.Generator(device="cuda").manual_seed(seed)
    return [(torch.randint(0, 3, (n, d), device="cuda", generator=g, dtype=torch.int8) - 1).to(torch.float32) for n in ln]


PAIR_FILL, PAIR_CHECKED = 100, 2
PAIR_RATE_BAND = 0.02        # |batched pair's hit rate - sequential C1 + C2 oracle's| on the checked batches (tier 1 or 2 = a hit)


@pytest.fixture(scope="module")
def pair_stream():
    """configs[4]'s request stream (bench.mixed_tiers_section's shape: seed 21, Zipf 0.75)"""
    import bench
    ln = bench.KAGGLE_LN
    return [b[1].t().contiguous().to(torch.int32)
            for b in bench.make_batches(ln, B, PAIR_FILL + 2 * PAIR_CHECKED + 20, seed=21, device="cuda", dist="zipf", alpha=0.75)]


@pytest.fixture(scope="module")
def pair_oracle_rates(pair_stream, orc):
    """the SEQUENTIAL two-tier oracle (orc.C1C2 = request_to_c1_c2, mixed_precs_caching/evlfu_8.cpp:669-796 with the routing of
    :570-601) over the same stream, one request at a time: hits (tier 1 or 2) of the checked batches; policy-independent"""
    import bench
    ln = bench.KAGGLE_LN
    budget = int(0.02 * sum(ln))
    tabs = [np.zeros((n, D), np.float32) for n in ln]     # (hit / miss does not depend on the values)
    o = orc.C1C2(int(0.48 * budget) * 4, int(0.48 * budget) * 8, tabs, tabs, D, 23)
    hits = 0
    for i, r in enumerate(pair_stream[:PAIR_FILL + PAIR_CHECKED]):
        for q in r.cpu().numpy():
            t_ = o.request(q)[0]
            if i >= PAIR_FILL:
                hits += int((t_ != 0).sum())
    del o, tabs
    return hits / (PAIR_CHECKED * B * len(ln))


@pytest.mark.parametrize("policy", ["sampled", "plan", "setassoc"])
def test_bench_mixed_precision_tiers_at_full_size(E, orc, policy, pair_stream, pair_oracle_rates):
    """configs[4] as bench.mixed_tiers_section builds it: u8 C1 + u4 C2 at the 48-48-4 split of 2 % of the Kaggle rows
    (1 296 480 + 2 592 960 entries) and the alt-key tier, B = 16 384, probe + mixed-precision interaction in one launch.
    Round 5: the pair's hit rate (tier 1 or 2) on the checked batches against the sequential two-tier oracle's on the same
    stream (PAIR_RATE_BAND) -- what ties the set-associative / sampled / plan pair to request_to_c1_c2's semantics at bench size."""
    import bench
    from evstore_dlrm_amd import gpu_cache
    ln = bench.KAGGLE_LN
    T = len(ln)
    ws = _exact_tables(ln, D, 9)
    ev = E.EVTables(ws, D, 32)
    ev8, ev4 = ev.encode(8), ev.encode(4)
    budget = int(0.02 * sum(ln))
    c1 = E.GpuCache("evlfu", int(0.48 * budget) * 4, T, D, 8, "cpp", "cuda").set_batch_policy(policy)
    c2 = E.GpuCache("evlfu", int(0.48 * budget) * 8, T, D, 4, "cpp", "cuda").set_batch_policy(policy)
    assert (c1.capacity, c2.capacity) == (1296480, 2592960)
    c1.set_backing(ev8)
    c2.set_backing(ev4)
    fill, checked = PAIR_FILL, PAIR_CHECKED
    rq = pair_stream
    g = torch.Generator(device="cuda").manual_seed(3)
    x = torch.rand((B, D), device="cuda", generator=g)
    tier = torch.empty((B, T), dtype=torch.uint8, device="cuda")
    off = torch.arange(B, device="cuda", dtype=torch.int64).repeat(T, 1).contiguous()
    for r in rq[:fill]:
        gpu_cache.lookup_interact_c1c2(c1, c2, r, x, tier=tier)
    assert c1.batch_stats()["size"] > 0.9 * c1.capacity
    rs = np.random.RandomState(8)

    def check(r, R, tier_np, k1, k2, alt_rows=None):
        q = _query_keys(r.cpu().numpy())
        in1, in2 = np.isin(q, k1), np.isin(q, k2)
        assert np.array_equal(tier_np == 1, in1), "tier 1 flags differ from C1's residency at the batch start"
        assert np.array_equal(tier_np == 2, in2 & ~in1), "tier 2 flags differ from C2's residency at the batch start"
        idx = r.t().contiguous().to(torch.int64)
        if alt_rows is not None:   # tier 3: the ALT row is served (same table here), everything else the key's own row
            t3 = torch.from_numpy((tier_np == 3).T.copy()).cuda()
            idx = torch.where(t3, alt_rows, idx)
        R_ref = E.apply_emb_interact(x, off, idx, ev, one_index_per_bag=True)   # fp32 rows of the same values
        torch.testing.assert_close(R, R_ref, rtol=RTOL, atol=2e-6)
        sel = torch.from_numpy(np.sort(rs.choice(B, 32, replace=False))).cuda()
        ly = [ws[k][idx[k][sel]].cpu().numpy() for k in range(T)]
        np.testing.assert_allclose(R[sel].cpu().numpy(), orc.interact_features(x[sel].cpu().numpy(), ly), rtol=RTOL, atol=2e-6)

    pair_hits = 0
    for r in rq[fill:fill + checked]:
        k1, k2 = np.sort(_keys(c1.batch_dump())), np.sort(_keys(c2.batch_dump()))
        assert np.unique(k1).size == k1.size and np.unique(k2).size == k2.size
        t_, R = gpu_cache.lookup_interact_c1c2(c1, c2, r, x, tier=tier)
        torch.cuda.synchronize()
        check(r, R, t_.cpu().numpy(), k1, k2)
        pair_hits += int((t_ != 0).sum())
        s1, s2 = c1.batch_stats(), c2.batch_stats()
        assert s1["size"] <= c1.capacity and s2["size"] <= c2.capacity
    rate_b = pair_hits / (checked * B * T)
    print("configs[4] full size, %s: batched pair hit rate %.4f, sequential C1 + C2 oracle %.4f" % (policy, rate_b, pair_oracle_rates))
    assert abs(rate_b - pair_oracle_rates) <= PAIR_RATE_BAND, (policy, rate_b, pair_oracle_rates)
    # the alt-key tier as the bench attaches it: alt key of (t, r) = row r % 4096 of the same table
    alt = [torch.from_numpy(((np.arange(n, dtype=np.int64) % min(n, 4096)) * 100 + (t + 1)).astype(np.uint32).view(np.int32)).cuda()
           for t, n in enumerate(ln)]
    c3 = E.GpuAltKeyTier(int(0.04 * budget) * 8 + 64, alt, "cuda")
    for r in rq[fill + checked:fill + checked + 20]:
        gpu_cache.lookup_interact_c1c2c3(c1, c2, c3, r, x, tier=tier)
    n3 = 0
    for r in rq[fill + checked + 20:fill + 2 * checked + 20]:
        k1, k2 = np.sort(_keys(c1.batch_dump())), np.sort(_keys(c2.batch_dump()))
        members, _ = c3.batch_dump()
        mk = np.sort((members[:, 0].astype(np.int64) << 32) | members[:, 1].astype(np.int64))
        t_, R = gpu_cache.lookup_interact_c1c2c3(c1, c2, c3, r, x, tier=tier)
        torch.cuda.synchronize()
        tier_np = t_.cpu().numpy()
        q = _query_keys(r.cpu().numpy())
        is3 = tier_np == 3
        n3 += int(is3.sum())
        # a tier-3 key was a double miss, a member of C3, and its alt row was resident in C1 or C2 when the batch started
        assert not (is3 & (np.isin(q, k1) | np.isin(q, k2))).any() and np.isin(q[is3], mk).all()
        rows_np = r.cpu().numpy().astype(np.int64)
        alt_np = np.stack([rows_np[:, t] % min(n, 4096) for t, n in enumerate(ln)], 1)
        aq = _query_keys(alt_np)
        assert (np.isin(aq[is3], k1) | np.isin(aq[is3], k2)).all()
        check(r, R, np.where(is3, 0, tier_np) if False else tier_np, k1, k2, alt_rows=torch.from_numpy(alt_np.T.copy()).cuda())
    assert n3 > 0, "the alt-key tier never served a row: the three-tier path was not exercised"
    del c1, c2, c3, ev, ev8, ev4, ws
    torch.cuda.empty_cache()


@pytest.mark.parametrize("bits,Bq", [(8, 16384 + 3), (4, 16384 + 3), (16, 16384 + 3), (8, 40000 + 7), (4, 40000 + 7)])
def test_reduced_precision_tables_at_full_size(E, orc, bits, Bq):
    """What bench.py's reduced-precision lines time: the 26 Kaggle tables (33.76 M rows) in the reference's u16 / u8 / u4 row
    layouts, random codes (every code decodes; u16 with its tail codes), d = 36 -- the rows-in-registers kernel of
    evs_fused_rfq.hip, whose d = 36 u8 / u4 rows travel as ONE load per lane with the tail chunk folded in (round 3): the
    LAST rows of every table among the indices (a load that ran past a row would run past the table there), the declared
    form, lS_o given and the two-call path give the same bits, and sampled samples agree with the oracle's decoders."""
    import bench
    ln = bench.KAGGLE_LN
    T = len(ln)
    ev = bench.make_tables(ln, D, seed=5, bits=bits, codes="random")
    g = torch.Generator(device="cuda").manual_seed(17)
    idx = torch.stack([torch.randint(0, n, (Bq,), device="cuda", generator=g) for n in ln])
    for k, n in enumerate(ln):
        idx[k, :4] = torch.tensor([n - 1, 0, max(n - 2, 0), n - 1], device="cuda")
        idx[k, -1] = n - 1
    off = torch.arange(Bq, device="cuda").repeat(T, 1)
    x = torch.randn(Bq, D, device="cuda")
    a = E.apply_emb_interact(x, off, idx, ev, one_index_per_bag=True, check_indices=True)
    b = E.apply_emb_interact(x, off, idx, ev, check_indices=True)
    assert torch.equal(a, b) and torch.equal(a[:, :D], x)
    if Bq <= 20000:
        c = E.interact_features(x, E.apply_emb(off, idx, ev, None, lazy=False))
        if bits == 8:   # (round 5: the fused u8 launch multiplies on the integer matrix pipe, the two-call path runs fp32 chains)
            torch.testing.assert_close(a, c, rtol=RTOL, atol=2e-6 * max(1.0, float(c.abs().max())))
        else:
            assert torch.equal(a, c)
    sel = np.array([0, 1, 2, 3, Bq // 2, Bq - 2, Bq - 1])
    sel_t = torch.from_numpy(sel).cuda()
    raws = [ev.raw[k][idx[k][sel_t]].cpu().numpy() for k in range(T)]
    ly = [orc.decode(r, bits, D) for r in raws]
    want = orc.interact_features(x[sel_t].cpu().numpy(), ly)
    np.testing.assert_allclose(a[sel_t].cpu().numpy(), want, rtol=RTOL, atol=2e-6 * max(1.0, float(np.abs(want).max())))
    del ev
    torch.cuda.empty_cache()


# MLPerf DLRM (Criteo Terabyte) cardinalities with --max-ind-range=40000000 (bench/run_and_time.sh:17): external, a
# synthetic shape only (bench.py --shape terabyte, tools/sweep.py)
TERABYTE_LN = [39884406, 39043, 17289, 7420, 20263, 3, 7120, 1543, 63, 38532951, 2953546, 403346, 10, 2208, 11938, 155,
               4, 976, 14, 39979771, 25641295, 39664984, 585935, 12972, 108, 36]


@pytest.mark.parametrize("d", [64, 128])
def test_terabyte_cardinalities_properties(E, d):
    """BASELINE configs[3]'s tables on ONE GPU (58 GB at d = 64, 116 GB at d = 128 of the 288 GB): 40 M-row tables of 10 to
    20 GB each, row byte offsets far beyond 2^32.  bag = 1 -> every pooled row IS the addressed row (exact copy), indices
    in the LAST rows of the giant tables included; pairs -> the fp32 sum of the two rows; the fused launch = the two-call
    path bit for bit; x passthrough; one column of the triangle recomputed from the rows."""
    import bench
    ln = TERABYTE_LN
    T = len(ln)
    ev = bench.make_tables(ln, d, seed=2)
    Bt = 4096
    g = torch.Generator(device="cuda").manual_seed(11)
    idx = torch.stack([torch.randint(0, n, (Bt,), device="cuda", generator=g) for n in ln])
    for k, n in enumerate(ln):   # the last rows (highest byte offsets) and the first
        idx[k, :8] = torch.arange(n - 1, max(n - 9, -1), -1, device="cuda")[:8].clamp_(0) if n >= 8 else torch.zeros(8, dtype=torch.int64, device="cuda")
        idx[k, 8] = 0
    assert int(idx[0, 0]) * d * 4 > 2 ** 32
    off = torch.arange(Bt, device="cuda").repeat(T, 1)
    ly = E.apply_emb(off, idx, ev, check_indices=True)
    for k in (0, 5, 9, 19, 20, 21, 25):
        assert torch.equal(ly[k], ev.fp32_view(k)[idx[k]]), k
    ly1 = E.apply_emb(off, idx, ev, one_index_per_bag=True)
    for k in range(T):
        assert torch.equal(ly1[k], ly[k])
    off2 = (torch.arange(Bt // 2, device="cuda") * 2).repeat(T, 1)
    ly2 = E.apply_emb(off2, idx, ev)
    for k in (0, 19, 21):
        rows = ev.fp32_view(k)[idx[k]]
        assert torch.equal(ly2[k], rows[0::2] + rows[1::2])
    x = torch.randn(Bt, d, device="cuda")
    a = E.apply_emb_interact(x, off, idx, ev, check_indices=True)
    b = E.apply_emb_interact(x, None, idx, ev, one_index_per_bag=True)
    c = E.interact_features(x, ly)
    assert torch.equal(a, b) and torch.equal(a, c) and torch.equal(a[:, :d], x)
    for k in (0, 9, 21, 25):
        f = k + 1
        col = d + f * (f - 1) // 2
        want = (ev.fp32_view(k)[idx[k]].double() * x.double()).sum(1)
        torch.testing.assert_close(a[:, col].double(), want, rtol=1e-5, atol=1e-6)
    # a bad index one past the end of a giant table is caught, not read
    bad = idx.clone()
    bad[19, 5] = ln[19]
    with pytest.raises(E.EvsError):
        E.apply_emb_interact(x, off, bad, ev, check_indices=True)
    del ev, ly, ly1, ly2
    torch.cuda.empty_cache()


@pytest.mark.parametrize("policy", ["count", "rows", "rows+replicate", "rowsplit"])
def test_sharded_op_at_terabyte_cardinalities_eight_virtual_ranks(E, policy):
    """BASELINE configs[3] through the SHARDED op (dlrm_s_pytorch.py:543-570 distributed_forward,
    bench/dlrm_s_criteo_terabyte.sh:24: d = 64, --max-ind-range=40000000): 8 virtual ranks on one GPU over the 58 GB of
    Terabyte-cardinality tables (every rank's tables are VIEWS of the one model: whole tables, or its row range of the
    40 M-row ones under `rowsplit`), the exchange done by hand as all_to_all_single lays the blocks out
    (extend_distributed.py:389-426).  One index per bag, indices in the LAST rows of the giant tables (byte offsets
    beyond 2^32 inside a rank's shard): every rank's R slice equals the single-process fused launch BIT FOR BIT, through
    the one-off and the planned step.  `rows+replicate` leaves ranks that own no table (empty send blocks); `count` is the
    reference's 4-4-3-3-3-3-3-3 split."""
    import bench
    from evstore_dlrm_amd import sharded
    ln, d, world, Bl = TERABYTE_LN, 64, 8, 512
    T, Bg = len(ln), 8 * 512
    ev = bench.make_tables(ln, d, seed=2)
    g = torch.Generator(device="cuda").manual_seed(23)
    idx = torch.stack([torch.randint(0, n, (Bg,), device="cuda", generator=g) for n in ln])
    for k, n in enumerate(ln):
        # the last and the first rows of every table, and -- for the row-split tables -- both ends of every rank's range
        edge = [n - 1, 0, max(n - 2, 0)]
        for r in range(world):
            lo, hi = sharded.row_range(n, r, world)
            if hi > lo:
                edge += [lo, hi - 1]
        e = torch.tensor(edge, device="cuda", dtype=torch.int64)
        idx[k, :e.numel()] = e
        idx[k, -1] = n - 1
    assert int(idx[19, 0]) * d * 4 > 2 ** 32
    off = torch.arange(Bg, device="cuda").repeat(T, 1)
    x = torch.randn(Bg, d, device="cuda")
    R_ref = E.apply_emb_interact(x, off, idx, ev, one_index_per_bag=True, check_indices=True)
    lS_o, lS_i = [off[k] for k in range(T)], [idx[k] for k in range(T)]
    owner = sharded.plan_placement(ln, world, policy)
    ops = []
    for r in range(world):
        held = {}
        for t in range(T):
            if owner[t] in (r, -1):
                held[t] = ev.fp32_view(t)
            elif owner[t] == -2:
                lo, hi = sharded.row_range(ln[t], r, world)
                held[t] = ev.fp32_view(t)[lo:hi]
        op = sharded.ShardedEmbeddingInteract(ln, d, r, world, held, sharded.HipBackend(torch.device("cuda")), policy=policy,
                                              one_index_per_bag=True)
        for t, k in op.local_id.items():    # views, not copies: 8 ranks x 58 GB would not fit
            lo = sharded.row_range(ln[t], r, world)[0] if owner[t] == -2 else 0
            assert op.ev.raw[k].data_ptr() == ev.raw[t].data_ptr() + lo * d * 4
        ops.append(op)
    n_own = sorted(len(op.my_own) for op in ops)
    if policy == "count":
        assert [len(op.my_own) for op in ops] == [4, 4, 3, 3, 3, 3, 3, 3]
    if policy == "rows":
        assert sum(n_own) == T and n_own[0] >= 1
    if policy == "rows+replicate":
        assert n_own[0] == 0 and sum(n_own) == sum(1 for n in ln if n > 1_000_000), n_own   # ranks that own nothing
    if policy == "rowsplit":
        assert all(not op.my_own and len(op.split) == sum(1 for n in ln if n > 1_000_000) for op in ops)
    sends = [op.pool(lS_o, lS_i)[0] if op.any_sharded else None for op in ops]
    assert E._lib.lib().evs_check_index_errors(None) == 0
    for r, op in enumerate(ops):
        _, _, out_splits = op._splits(Bg)
        recv = torch.cat([sends[p].reshape(world, sends[p].numel() // world)[r] for p in range(world)])
        assert recv.numel() == sum(out_splits)
        sl = slice(r * Bl, (r + 1) * Bl)
        R = op.finish((None, recv, Bg, Bl, out_splits), x[sl], lS_o, lS_i)
        assert torch.equal(R, R_ref[sl]), (policy, r, float((R - R_ref[sl]).abs().max()))
        out = torch.empty_like(R)
        pl = op.plan(x[sl], lS_o, lS_i, out=out)
        pl["recv"].copy_(recv)
        op.run_finish(pl, None)
        assert torch.equal(out, R_ref[sl]), (policy, r)
    del ops, sends, ev
    torch.cuda.empty_cache()
