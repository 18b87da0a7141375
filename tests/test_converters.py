"""f4: the .bin / alt-key writers against files written by the reference's own converter scripts
(tests/golden/make_golden_converters.py ran script/convert_ev_to_binary.py and convert_altkeys_to_binary.py)."""
import os

import numpy as np
import pytest

import evstore_dlrm_amd as E
from evstore_dlrm_amd import converters

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "converters.npz"))
CASES = bytes(G["case_names"]).decode().split(",")


@pytest.mark.parametrize("name", CASES)
def test_convert_ev_to_binary_matches_reference_bytes_and_placement(name, tmp_path):
    sub = tmp_path / "root" / "tabs"
    sub.mkdir(parents=True)
    p = sub / "ev-table-3.csv"
    p.write_bytes(bytes(G[name + "_csv"]))
    out = converters.convert_ev_to_binary(str(p), bytes(G[name + "_readas"]).decode(), verbose=False)
    assert os.path.normpath(os.path.relpath(out, sub)) == os.path.normpath(bytes(G[name + "_relout"]).decode())
    assert open(out, "rb").read() == bytes(G[name + "_bin"])


def test_convert_ev_to_binary_errors(tmp_path, capsys):
    p = tmp_path / "t.csv"
    p.write_text("0,1\n1,2\n")
    for bad in ("fp16", "nonsense"):
        with pytest.raises(SystemExit):
            converters.convert_ev_to_binary(str(p), bad, verbose=False)
    p.write_text("0,1\n1,300\n")
    with pytest.raises(SystemExit):
        converters.convert_ev_to_binary(str(p), "u_char", verbose=False)  # struct.pack('>B', 300) raises in the reference
    assert "ERROR" in capsys.readouterr().out


def test_altkeys_folder_matches_reference(tmp_path):
    for t in (1, 3, 26):
        (tmp_path / ("ev-table-%d.csv" % t)).write_bytes(bytes(G["alt%d_txt" % t]))
    (tmp_path / "notes.txt").write_text("ignored\n")
    outs = converters.convert_altkeys_folder(str(tmp_path), verbose=False)
    assert sorted(os.path.basename(o) for o in outs) == sorted("ev-table-%d.bin" % t for t in (1, 3, 26))
    for t in (1, 3, 26):
        assert (tmp_path / "binary" / ("ev-table-%d.bin" % t)).read_bytes() == bytes(G["alt%d_bin" % t])


def test_write_altkeys_from_arrays_and_oracle_reader(tmp_path):
    """write_altkeys(tables, rows) = the text path; the tier's loader (aprx_embedding.cpp:243-251: 4 B big-endian,
    alt_key = row * 100 + table) reads back what was written."""
    txt = bytes(G["alt3_txt"]).decode().split()
    tids = [int(l.split("-")[0]) for l in txt]
    rids = [int(l.split("-")[1]) for l in txt]
    p = converters.write_altkeys(tids, rids, str(tmp_path / "a.bin"))
    assert open(p, "rb").read() == bytes(G["alt3_bin"])
    words = np.fromfile(p, dtype=">u4").astype(np.int64)
    assert np.array_equal(words % 100, tids) and np.array_equal(words // 100, rids)
    with pytest.raises(SystemExit):
        converters.altkey_words([1], [1 << 31])


def test_empty_folder_errors(tmp_path):
    with pytest.raises(SystemExit):
        converters.convert_altkeys_folder(str(tmp_path), verbose=False)


@pytest.mark.gpu
@pytest.mark.parametrize("bits", [32, 16, 8, 4])
def test_to_bin_dir_round_trip_and_oracle_bytes(bits, tmp_path):
    """EVTables.encode(bits).to_bin_dir() writes what reduce_precision.py + convert_ev_to_binary.py would (the oracle
    encoders are pinned by encoders.npz), and from_bin_dir reads the same tables back."""
    import torch
    from oracle import oracle as orc
    rs = np.random.RandomState(bits)
    n_rows, d = [300, 7, 1, 2049], 36
    tabs = [rs.uniform(-1, 1, size=(n, d)).astype(np.float32) for n in n_rows]
    ev = E.EVTables.from_fp32([torch.from_numpy(t) for t in tabs], device="cuda:0")
    enc = ev if bits == 32 else ev.encode(bits)
    paths = enc.to_bin_dir(str(tmp_path / "binary"))
    assert [os.path.basename(p) for p in paths] == ["ev-table-%d.bin" % (k + 1) for k in range(len(n_rows))]
    for k, p in enumerate(paths):
        want = tabs[k].tobytes() if bits == 32 else orc.encode_table(tabs[k], bits).tobytes()
        assert open(p, "rb").read() == want
    back = E.EVTables.from_bin_dir(str(tmp_path / "binary"), n_tables=len(n_rows), d=d, codec=bits, device="cuda:0")
    for a, b in zip(enc.raw, back.raw):
        assert torch.equal(a.reshape(-1), b.reshape(-1))
