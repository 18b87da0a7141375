"""evstore_dlrm_amd.evstore_utils against what the reference's own evstore_utils.py functions wrote and returned
(tests/golden/evstore_utils.npz, made by tests/golden/make_golden_utils.py in the build container): training_config.txt and
the 26 workload-trace files byte for byte, the CSV tables loaded to the same fp32 values, the traces read back as request rows."""
import contextlib
import io
import os

import numpy as np
import pytest

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "evstore_utils.npz")
KAGGLE = [1460, 583, 10131227, 2202608, 305, 24, 12517, 633, 3, 93145, 5683, 8351593, 3194, 27, 14992, 5461306, 10, 5652, 2173, 4,
          7046547, 18, 15, 286181, 105, 142572]


@pytest.fixture(scope="module")
def U():
    import evstore_dlrm_amd  # noqa: F401  (the import shim)
    from evstore_dlrm_amd import evstore_utils
    return evstore_utils


def test_training_config_round_trip_and_bytes(U, tmp_path):
    g = np.load(GOLDEN)
    p = str(tmp_path / U.TRAINING_CONFIG_FILE)
    tfm = {i: i for i in range(1, 27)}
    with contextlib.redirect_stdout(io.StringIO()) as out:
        U.store_training_config(p, tfm, 306969, 51162, np.array(KAGGLE), 13)
        a, b, c, d, e = U.read_training_config(p)
    assert open(p, "rb").read() == g["config_bytes"].tobytes()
    assert a == tfm and (b, c, e) == (306969, 51162, 13) and np.array_equal(d, np.array(KAGGLE)) and isinstance(d, np.ndarray)
    assert out.getvalue() == ""   # the formats are the contract, not console chatter
    assert U.parse_training_config(U.format_training_config(table_feature_map=tfm, nbatches=1, nbatches_test=2, ln_emb=[3, 4], m_den=5))[1:3] == (1, 2)
    with pytest.raises(ValueError):
        U.parse_training_config("header only\n")
    # ... and the reference's own file parses to the same values
    q = str(tmp_path / "ref.txt")
    open(q, "wb").write(g["config_bytes"].tobytes())
    with contextlib.redirect_stdout(io.StringIO()):
        assert U.read_training_config(q)[0] == tfm


def test_workload_traces_bytes_and_read_back(U, tmp_path):
    g = np.load(GOLDEN)
    rows = g["rows"]
    work = [[str(k + 1) + "-" + str(int(rows[i, k])) for k in range(26)] for i in range(len(rows))]
    with contextlib.redirect_stdout(io.StringIO()) as out:
        U.write_inf_workload_to_file(str(tmp_path), work)
    assert out.getvalue() == ""
    for k in range(26):
        assert open(tmp_path / ("workload-group-%d.csv" % (k + 1)), "rb").read() == g["trace_%d" % (k + 1)].tobytes(), k
    back = U.read_inf_workload(str(tmp_path))
    assert back.dtype == np.int32 and np.array_equal(back, rows)
    cwd = os.getcwd()
    os.chdir(tmp_path)
    try:
        with contextlib.redirect_stdout(io.StringIO()):
            d = U.prepare_inference_trace_folder("kaggle", 10)
        assert d == os.path.join("logs", "inf-workload-traces", "kaggle", "inference=10") and os.path.isdir(d)
    finally:
        os.chdir(cwd)


def test_load_new_ev_table_equals_reference(U, tmp_path):
    g = np.load(GOLDEN)
    for k in range(26):
        open(tmp_path / ("ev-table-%d.csv" % (k + 1)), "wb").write(g["csv_%d" % (k + 1)].tobytes())
    ld = {"state_dict": {}}
    with contextlib.redirect_stdout(io.StringIO()):
        U.load_new_ev_table(ld, str(tmp_path))
    for k in range(26):
        t = ld["state_dict"]["emb_l.%d.weight" % k]
        assert t.dtype.is_floating_point and np.array_equal(t.numpy(), g["loaded_%d" % (k + 1)]), k


@pytest.mark.gpu
def test_csv_tables_into_hbm(U, tmp_path):
    import torch
    g = np.load(GOLDEN)
    for k in range(26):
        open(tmp_path / ("ev-table-%d.csv" % (k + 1)), "wb").write(g["csv_%d" % (k + 1)].tobytes())
    ev = U.ev_tables_from_csv_dir(str(tmp_path))
    for k in range(26):
        assert torch.equal(ev.fp32_view(k).cpu(), torch.from_numpy(g["loaded_%d" % (k + 1)]))


def test_replay_of_a_recorded_workload(U, tmp_path):
    """tools/replay_workload.py = the reference's manual check cache_algo/EvLFU_C1_Cython/test.py on this package's modules: the
    golden request stream written as the 26 trace files (write_inf_workload_to_file), read back, replayed through the Cython
    EvLFU surface and through EvLFU_C1 over mmap'ed .bin tables -> the perfect-hit counts of the reference's own traces."""
    import importlib.util
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from conftest import load_golden
    t, tc = load_golden("cache_traces"), load_golden("cython_traces")
    reqs = t["requests"][:400]
    (tmp_path / "w").mkdir()
    (tmp_path / "ev" / "binary").mkdir(parents=True)
    with contextlib.redirect_stdout(io.StringIO()):
        U.write_inf_workload_to_file(str(tmp_path / "w"), [[str(k + 1) + "-" + str(int(r[k])) for k in range(26)] for r in reqs])
    from oracle import oracle as orc
    tabs = orc.kaggle_tables([int(n) for n in t["n_rows"]], int(t["table_seed"]))   # (the tables the traces were recorded on)
    for k, w in enumerate(tabs):
        np.ascontiguousarray(w, np.float32).tofile(tmp_path / "ev" / "binary" / ("ev-table-%d.bin" % (k + 1)))
    spec = importlib.util.spec_from_file_location("replay_workload", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "replay_workload.py"))
    rw = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(rw)

    def unpack(packed, n):
        return np.unpackbits(packed, axis=1)[:, :26].astype(bool)[:n]
    with contextlib.redirect_stdout(io.StringIO()) as out:
        got = rw.main([str(tmp_path / "w"), str(tmp_path / "ev"), "--cache-size", "78", "--algo", "evlfu_cython"])
    assert got == int(unpack(tc["cython_main_cap78_hits"], 1500)[:400].all(1).sum())
    assert "perfect hit: %d" % got in out.getvalue() and "(400, 26)" in out.getvalue()
    with contextlib.redirect_stdout(io.StringIO()):
        got = rw.main([str(tmp_path / "w"), str(tmp_path / "ev"), "--cache-size", "768", "--algo", "evlfu"])
    assert got == int(unpack(t["evlfu_cap768_hits"], 1500)[:400].all(1).sum())


def test_ragged_and_overlong_requests_are_written_as_the_reference_writes_them(U, tmp_path):
    """evstore_utils.py:68-73: key j of a request goes to file j -- a short request leaves the later files shorter, a request
    with more keys than files raises (the reference indexes past its 26 files); nothing is truncated silently."""
    out = str(tmp_path)
    reqs = [["%d-%d" % (t + 1, 10 + t) for t in range(26)], ["%d-%d" % (t + 1, 20 + t) for t in range(3)]]
    with contextlib.redirect_stdout(io.StringIO()):
        U.write_inf_workload_to_file(out, reqs)
    for k in range(26):
        lines = open(os.path.join(out, "workload-group-%d.csv" % (k + 1))).read().splitlines()
        assert lines == ["G%d_key" % (k + 1), "%d-%d" % (k + 1, 10 + k)] + (["%d-%d" % (k + 1, 20 + k)] if k < 3 else [])
    with pytest.raises(IndexError):
        U.write_inf_workload_to_file(out, [["1-1"] * 27])
