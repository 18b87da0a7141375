"""CPU: host-side logic of the product package that needs no GPU -- storage manager readers over the
reference's on-disk format, the host row decoders, placement planning, module surfaces."""
import numpy as np
import pytest

from conftest import load_golden
from oracle import oracle as orc


def test_storage_manager_file_and_mmap_readers(tmp_path):
    from evstore_dlrm_amd.emb_storage import storage_manager as sm
    t = load_golden("cache_traces")
    tabs = orc.kaggle_tables([int(n) for n in t["n_rows"]], int(t["table_seed"]))
    (tmp_path / "binary").mkdir()
    for k, w in enumerate(tabs):
        w.tofile(tmp_path / "binary" / ("ev-table-%d.bin" % (k + 1)))
    for st in (sm.EmbStorage.FILEPY, sm.EmbStorage.MMAPFILEPY, sm.EmbStorage.DUMMY):
        sm.storage_type, sm.ev_precs = st, 32
        sm.load_ev_table_into_emb_stor(str(tmp_path))
        for (tb, r), want in zip(t["reader_probe"], t["reader_rows"]):
            assert np.array_equal(np.asarray(sm.get_val_from_storage(int(tb), int(r)), np.float32), want)
        vals = sm.get_arr_val_from_storage([[1, 0], [26, 17]])
        assert len(vals) == 2 and len(vals[0]) == 36
        _, ly = sm.request_to_emb_storage([int(v) for v in t["requests"][0]])
        assert len(ly) == 26 and tuple(ly[0].shape) == (1, 36) and ly[0].requires_grad
        assert np.array_equal(ly[5].detach().numpy()[0], tabs[5][t["requests"][0][5]])
        sm.close_any_db_conn()


@pytest.mark.parametrize("bits", [16, 8, 4])
def test_host_row_decoders_match_oracle(tmp_path, bits):
    from evstore_dlrm_amd import codecs_host
    from evstore_dlrm_amd.emb_storage import storage_manager as sm
    rs = np.random.RandomState(bits)
    w = rs.uniform(-1, 1, size=(40, 36)).astype(np.float32)
    raw = orc.encode_table(w, bits)
    want = orc.decode(raw, bits, 36)
    for r in range(40):
        got = codecs_host.decode_row(raw[r].tobytes(), bits, 36)
        assert np.array_equal(got.view(np.uint32), want[r].view(np.uint32))
    # and through the storage manager with a reduced-precision store
    (tmp_path / "binary").mkdir()
    for k in range(26):
        raw.tofile(tmp_path / "binary" / ("ev-table-%d.bin" % (k + 1)))
    sm.storage_type, sm.ev_precs = sm.EmbStorage.FILEPY, bits
    sm.load_ev_table_into_emb_stor(str(tmp_path))
    assert np.array_equal(np.asarray(sm.get_val_from_storage(3, 7), np.float32).view(np.uint32), want[7].view(np.uint32))
    sm.close_any_db_conn()
    sm.ev_precs = 32


def test_unsupported_storage_type_exits():
    from evstore_dlrm_amd.emb_storage import storage_manager as sm
    sm.storage_type = sm.EmbStorage.ROCKSDB
    with pytest.raises(SystemExit):
        sm.load_ev_table_into_emb_stor("/nonexistent")
    sm.storage_type = sm.EmbStorage.DUMMY


def test_fused_supported_shapes_and_module_surface():
    import evstore_dlrm_amd as E
    from evstore_dlrm_amd import dlrm_ops, evstore_ops
    from evstore_dlrm_amd.cache_algo import EvLFU_C1, LRU, LFU, cpp_socket_client
    assert dlrm_ops.fused_supported(27, 36) and dlrm_ops.fused_supported(9, 16) and not dlrm_ops.fused_supported(33, 36)
    assert not dlrm_ops.fused_supported(27, 20)
    for mod, fn in ((EvLFU_C1, "request_to_ev_lfu"), (LRU, "request_to_lru"), (LFU, "request_to_lfu")):
        assert callable(getattr(mod, "init")) and callable(getattr(mod, fn))
    assert callable(evstore_ops.apply_emb_evstore) and callable(cpp_socket_client.request_to_cpp_cache)
    assert E._lib.lib().evs_fused_dim_supported(36) == 1 and E._lib.lib().evs_fused_dim_supported(40) == 0


def test_cdf_writer_and_latency_definition_match_the_reference(tmp_path):
    """a16: calculate_and_write_cdf (dlrm_s_pytorch_C1.py:299-326) -- byte-identical CSV to what the reference's own
    function wrote for the same time stamps (tests/golden/cdf.npz, make_golden.py gen_cdf)."""
    import os
    import sys
    import numpy as np
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import make_golden as G
    from evstore_dlrm_amd import inference_loop as IL
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "cdf.npz"))
    for n, seed in ((5000, 3), (2345, 4)):
        stamps = G.cdf_stamps(n, seed)
        out = IL.calculate_and_write_cdf(str(tmp_path / ("cdf%d" % n)), "evlfu", stamps)
        assert out.endswith("evlfu-cdf.csv")
        assert open(out).read() == bytes(g["n%d_seed%d_csv" % (n, seed)]).decode()
        assert len(IL.latencies(stamps)) == n - 1     # the reference's loop bound drops the last request
    # short runs (below 1000 requests the reference divides by zero): every point kept
    out = IL.calculate_and_write_cdf(str(tmp_path / "short"), "lru", G.cdf_stamps(40, 1))
    assert open(out).read().count("\n") == 1 + 39


def test_inference_loop_stamps_and_dlrm_wrap_order():
    """inference(): one stamp per request at the top of the loop plus one at the end; dlrm_wrap hands the forward the
    batch in the reference's argument order (dlrm_s_pytorch.py:131-147)."""
    import torch
    from evstore_dlrm_amd import inference_loop as IL
    seen = []

    def fwd(X, lS_o, lS_i):
        seen.append((tuple(X.shape), tuple(lS_o.shape), len(lS_i)))
        return X.sum()

    ld = [(torch.zeros(4, 13), torch.zeros(3, 4, dtype=torch.int64), [torch.zeros(4, dtype=torch.int64)] * 3)] * 5
    got = []
    stamps = IL.inference(ld, fwd, use_gpu=False, device="cpu", consume=got.append)
    assert len(stamps) == 6 and stamps == sorted(stamps) and len(got) == 5
    assert seen[0] == ((4, 13), (3, 4), 3)


def test_file_tier_reader_pool_reads_the_reference_file_format(tmp_path):
    """evs_filetier_open / _fetch without a GPU (budget 0: nothing is registered): rows come back exactly as the
    reference's mmap reader returns them (emb_storage/mmap_file_read.py:32-40: seek(144 * row), read(144))."""
    import numpy as np
    import evstore_dlrm_amd as E
    rs = np.random.RandomState(2)
    n_rows = [50, 3, 4000, 17]
    tabs = [rs.uniform(-1, 1, size=(n, 36)).astype(np.float32) for n in n_rows]
    paths = []
    for k, w in enumerate(tabs):
        p = tmp_path / ("ev-table-%d.bin" % (k + 1))
        w.tofile(p)
        paths.append(str(p))
    ft = E.FileTier(paths, 144, 0)
    assert ft.n_rows == n_rows and not any(ft.registered) and ft.pinned_bytes == 0
    keys = np.array([((t + 1) << 32) | r for t, r in ((0, 0), (2, 3999), (1, 2), (3, 16), (2, 17), (0, 49))] * 1500, np.uint64)
    got = ft.fetch(keys).view(np.float32)      # 9000 keys: the threaded path
    for i in range(0, len(keys), 997):
        t, r = int(keys[i] >> 32) - 1, int(keys[i] & 0xffffffff)
        assert np.array_equal(got[i], tabs[t][r])
    ft.close()
    with pytest.raises(E.EvsError):
        E.FileTier([paths[0], str(tmp_path / "missing.bin")], 144, 0)
    with pytest.raises(E.EvsError):
        E.FileTier(paths, 100, 0)   # not a whole number of rows


def test_deferred_rows_see_keyword_and_nested_arguments_and_pool_entries_know_when_they_are_free():
    """CPU side of the deferred default (dlrm_ops.py): the argument walk of _DeferredRow.__torch_function__ (positional,
    keyword, nested lists / tuples / dicts) and the three "free again" conditions of a pool entry (the handed-out list died,
    no row escaped it, no view of the buffer is alive) -- the classes run on CPU tensors; the launches need a GPU."""
    import torch
    import evstore_dlrm_amd  # noqa: F401
    from evstore_dlrm_amd import dlrm_ops as D

    class Count:
        def __init__(self):
            self.n = 0

        def materialize(self):
            self.n += 1

    def rows(st, n=3):
        out = []
        for v in torch.zeros(n, 4, 2).unbind(0):
            r = v.as_subclass(D._DeferredRow)
            r._evs_state = st
            out.append(r)
        return out

    for call in (lambda r: torch.cat(r), lambda r: torch.cat(tensors=r), lambda r: torch.stack(tensors=r, dim=1),
                 lambda r: torch.cat(tensors=tuple(r)), lambda r: torch.einsum("ij,ij->i", [r[0], r[1]]),
                 lambda r: torch.add(r[0], other=r[1]), lambda r: r[0] + 1, lambda r: r[2].sum()):
        st = Count()
        out = call(rows(st))
        assert st.n >= 1, call
        assert type(out) is not D._DeferredRow or True
    st = Count()
    r = rows(st)
    assert r[0].shape == (4, 2) and r[0].dtype == torch.float32 and r[0].size(0) == 4 and r[0].is_contiguous() and st.n == 0
    D._touch({"a": [(r[0],)], "b": 3}, D._DeferredRow)
    assert st.n == 1

    e = D._PoolEntry(3, 4, 2, "cpu")
    assert e.free()
    ly = e.hand_out()
    assert not e.free()
    keep = ly[1]
    del ly
    assert e.out is None and not e.free()          # the list died, one of its rows escaped
    del keep
    assert e.free()
    ly = e.hand_out()
    with torch._C.DisableTorchFunctionSubclass():
        view = ly[2][1:3]
    del ly
    assert not e.free()                            # a view of the buffer is alive
    del view
    assert e.free()

    t = torch.zeros(3)
    assert D._version_of(t) == t._version
    with torch.inference_mode():
        u = torch.zeros(3)
    assert D._version_of(u) is None
