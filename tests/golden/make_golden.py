#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by IMPORTING the reference.

Runs only in the build container (needs /root/reference).  Nothing from the
reference travels: this script records inputs and the reference's outputs as
plain numpy arrays (.npz).  Re-run with:

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

What is pinned:
  * dlrm_s_pytorch.DLRM_Net.apply_emb / interact_features  (dlrm_s_pytorch.py:407, :483)
  * dlrm_data_pytorch.generate_dist_input_batch            (dlrm_data_pytorch.py:1011)
  * cache_algo/EvLFU_C1.py, LRU.py, LFU.py hit traces       (EvLFU_C1.py:97, LRU.py:38, LFU.py:69)
  * emb_storage/file_read.py + mmap_file_read.py row reads  (file_read.py:27, mmap_file_read.py:32)
  * script/reduce_precision.py encoders                     (reduce_precision.py:26,:140,:270)
  * dlrm_data_pytorch.collate_wrapper_criteo_offset         (dlrm_data_pytorch.py:397-410; `collate` mode)
"""
import os
import sys
import types
import struct
import tempfile
import hashlib
import random

import numpy as np

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True


def _stub(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def import_reference():
    import torch  # noqa: F401

    class _SW:  # torch.utils.tensorboard.SummaryWriter stand-in (dlrm_s_pytorch.py:101)
        def __init__(self, *a, **k):
            pass

    _stub("torch.utils.tensorboard", SummaryWriter=_SW)
    _stub("pyrocksdb")   # storage_rocksdb.py:1 (third-party, absent)
    _stub("EvLFU")       # storage_manager.py:18 (cython .so is cpython-36 only)
    for p in (REF, os.path.join(REF, "script"), os.path.join(REF, "emb_storage"),
              os.path.join(REF, "cache_algo")):
        if p not in sys.path:
            sys.path.insert(0, p)
    cwd = os.getcwd()
    os.chdir(REF)  # EvLFU_C1.py appends the relative path 'emb_storage'
    try:
        import dlrm_s_pytorch as D
        import dlrm_data_pytorch as DP
        import storage_manager as SM
        import file_read as FR
        import mmap_file_read as MFR
        import EvLFU_C1, LRU, LFU
    finally:
        os.chdir(cwd)
    return D, DP, SM, FR, MFR, EvLFU_C1, LRU, LFU


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


# --------------------------------------------------------------------------
# a1/a3: apply_emb + interact_features
# --------------------------------------------------------------------------
def gen_dlrm_case(D, DP, name, ln_emb, m_spa, B, n_idx, seed, itself=False,
                  weighted=False, criteo_layout=False, store_tables=True, n_idx_fixed=False):
    import torch
    np.random.seed(seed)
    torch.manual_seed(seed)
    ln_emb = np.asarray(ln_emb)
    T = len(ln_emb)
    m_den = 13
    ln_bot = np.array([m_den, 32, m_spa])
    F = T + 1
    n_int = (F * (F + 1)) // 2 if itself else (F * (F - 1)) // 2
    ln_top = np.array([n_int + m_spa, 16, 1])
    # data first, then the model: same RNG draw order as run() (dlrm_s_pytorch.py:1178 then :1326)
    if criteo_layout:
        # collate_wrapper_criteo_offset (dlrm_data_pytorch.py:397-410): one index per (table, sample)
        X = torch.tensor(np.random.rand(B, m_den).astype(np.float32))
        lS_i = torch.stack([torch.tensor(np.random.randint(0, n, size=B), dtype=torch.long)
                            for n in ln_emb])
        lS_o = torch.stack([torch.tensor(range(B)) for _ in range(T)])
    else:
        X, lS_o_l, lS_i = DP.generate_dist_input_batch(
            m_den, ln_emb, B, n_idx, n_idx_fixed, "uniform", 0, 1, -1, 1)
        lS_o = torch.stack(lS_o_l)  # collate_wrapper_random_offset (dlrm_data_pytorch.py:791)
    dlrm = D.DLRM_Net(
        m_spa, ln_emb, ln_bot, ln_top,
        arch_interaction_op="dot", arch_interaction_itself=itself,
        sigmoid_bot=-1, sigmoid_top=ln_top.size - 2, ndevices=-1,
        weighted_pooling=("fixed" if weighted else None))
    if weighted:
        # reference initialises v_W_l to ones (dlrm_s_pytorch.py:293); use non-trivial
        # per-row weights so the gather of weights is actually exercised
        for k in range(T):
            dlrm.v_W_l[k] = torch.tensor(
                np.random.uniform(0.5, 1.5, size=int(ln_emb[k])).astype(np.float32))
    with torch.no_grad():
        x = dlrm.apply_mlp(X, dlrm.bot_l)
        ly = dlrm.apply_emb(lS_o, lS_i, dlrm.emb_l, dlrm.v_W_l)
        R = dlrm.interact_features(x, ly)
        Z = dlrm.apply_mlp(R, dlrm.top_l)
    out = {
        "ln_emb": ln_emb.astype(np.int64), "m_spa": np.int64(m_spa), "B": np.int64(B),
        "itself": np.int64(itself), "seed": np.int64(seed),
        "X": X.numpy(), "x": x.numpy(), "R": R.numpy(), "Z": Z.numpy(),
        "lS_o": lS_o.numpy().astype(np.int64),
        "ly": np.stack([v.numpy() for v in ly]),
    }
    if criteo_layout:
        out["lS_i_stacked"] = lS_i.numpy().astype(np.int64)
    else:
        out["lS_i_cat"] = np.concatenate([v.numpy() for v in lS_i]).astype(np.int64)
        out["lS_i_nnz"] = np.array([v.numel() for v in lS_i], dtype=np.int64)
    tables = [e.weight.detach().numpy() for e in dlrm.emb_l]
    out["tables_sha256"] = np.array([sha(t) for t in tables])
    if store_tables:
        out["tables_cat"] = np.concatenate([t.reshape(-1) for t in tables])
    if weighted:
        out["vW_cat"] = np.concatenate([w.numpy() for w in dlrm.v_W_l])
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print("wrote", name, {k: getattr(v, "shape", None) for k, v in out.items()})


# --------------------------------------------------------------------------
# a6/a7/a13: cache policies over the file/mmap readers
# --------------------------------------------------------------------------
KAGGLE_LN = [1460, 583, 10131227, 2202608, 305, 24, 12517, 633, 3, 93145, 5683, 8351593,
             3194, 27, 14992, 5461306, 10, 5652, 2173, 4, 7046547, 18, 15, 286181, 105, 142572]


def make_bin_tables(root, n_rows, seed):
    """Write ev-table-{1..26}.bin fp32 LE raw (script/convert_ev_to_binary.py:58-69)."""
    rs = np.random.RandomState(seed)
    os.makedirs(os.path.join(root, "binary"), exist_ok=True)
    tabs = []
    for k, n in enumerate(n_rows):
        w = rs.uniform(-np.sqrt(1.0 / n), np.sqrt(1.0 / n), size=(n, 36)).astype(np.float32)
        w.tofile(os.path.join(root, "binary", "ev-table-%d.bin" % (k + 1)))
        tabs.append(w)
    return tabs


def zipf_stream(n_rows, n_req, seed, alpha=1.2, n_hot=40, p_hot=0.35, distinct_hot=False):
    """Request stream: Zipf per table + a set of hot whole requests that repeat
    (whole-request repeats are what drives keys into bucket 26 and triggers flush)."""
    rs = np.random.RandomState(seed)
    perms = [rs.permutation(n) for n in n_rows]
    def one():
        return [int(perms[k][min(rs.zipf(alpha) - 1, n - 1)]) for k, n in enumerate(n_rows)]
    hot = [one() for _ in range(n_hot)]
    if distinct_hot:  # no key shared between hot requests (tables must have >= n_hot rows)
        hot = [[int(perms[k][j]) for k in range(len(n_rows))] for j in range(n_hot)]
    reqs = []
    if distinct_hot:  # cold start: each hot request twice -> all its keys reach bucket 26
        for h in hot:
            reqs += [h, h]
    for _ in range(n_req - len(reqs)):
        if rs.rand() < p_hot:
            reqs.append(hot[rs.randint(n_hot)])
        else:
            reqs.append(one())
    return np.asarray(reqs, dtype=np.int32)


FLUSH_CAPS = (79, 80, 82)


def reset_evlfu(E):
    E.cap_C1 = -1; E.min_C1 = 0; E.vals_C1 = dict(); E.lists_C1 = dict()
    E.n_perfect_item_C1 = 0; E.max_perfect_item_C1 = 0


def reset_lru(L):
    import collections
    L.cap = -1; L.LRUCache = collections.OrderedDict()


def reset_lfu(L):
    L.cap = -1; L.least_freq = 1; L.node_for_freq.clear(); L.node_for_key.clear()


def key_to_pair(key):
    t, r = key.split("-")
    return (int(t), int(r))


def gen_cache_traces(SM, FR, MFR, EvLFU_C1, LRU, LFU):
    import io, contextlib
    n_rows = [min(n, 600) for n in KAGGLE_LN]
    tmp = tempfile.mkdtemp(prefix="evs_golden_")
    tabs = make_bin_tables(tmp, n_rows, seed=7)
    reqs_main = zipf_stream(n_rows, 1500, seed=11)
    # few hot whole-requests, repeated often: bucket 26 fills past 0.95*cap -> flush path
    reqs_flush = zipf_stream(n_rows, 1200, seed=13, n_hot=3, p_hot=0.7, distinct_hot=True)
    out = {"n_rows": np.asarray(n_rows, np.int64), "table_seed": np.int64(7),
           "requests": reqs_main, "requests_flush": reqs_flush}

    # a13: file and mmap readers return the same 36 floats as the table rows
    with contextlib.redirect_stdout(io.StringIO()):
        FR.open_files_as_binary(tmp)
        MFR.open_files_as_binary(tmp)
    probe = [(1, 0), (3, 599), (9, 2), (26, 17), (12, 333)]
    rows_f = np.array([FR.get(t, r) for t, r in probe], dtype=np.float32)
    rows_m = np.array([MFR.get(t, r) for t, r in probe], dtype=np.float32)
    for (t, r), a, b in zip(probe, rows_f, rows_m):
        assert np.array_equal(a, tabs[t - 1][r]) and np.array_equal(b, tabs[t - 1][r])
    out["reader_probe"] = np.asarray(probe, np.int64)
    out["reader_rows"] = rows_f

    SM.storage_type = SM.EmbStorage.FILEPY
    for algo, mod, reset, req_fn in (
            ("evlfu", EvLFU_C1, reset_evlfu, "request_to_ev_lfu"),
            ("lru", LRU, reset_lru, "request_to_lru"),
            ("lfu", LFU, reset_lfu, "request_to_lfu")):
        for cap in (64, 300, 768, 2000) + FLUSH_CAPS:
            reqs = reqs_flush if cap in FLUSH_CAPS else reqs_main
            reset(mod)
            random.seed(0)
            with contextlib.redirect_stdout(io.StringIO()):
                mod.init(cap)
            hits = np.zeros((len(reqs), 26), dtype=np.bool_)
            nflush = 0
            for i, rq in enumerate(reqs):
                buf = io.StringIO()
                with contextlib.redirect_stdout(buf):
                    h, vals = getattr(mod, req_fn)([int(v) for v in rq], False)
                nflush += buf.getvalue().count("flushing!")
                hits[i] = h
                for k in range(26):  # values are always the true rows (bit-exact pass-through)
                    v = vals[k].detach().numpy().reshape(-1)
                    assert np.array_equal(v, tabs[k][rq[k]]), (algo, cap, i, k)
            tag = "%s_cap%d" % (algo, cap)
            out[tag + "_hits"] = np.packbits(hits, axis=1)
            if algo == "evlfu":
                buckets = []
                for b in range(27):
                    for key in mod.lists_C1[b]:
                        buckets.append((b,) + key_to_pair(key))
                out[tag + "_final_buckets"] = np.asarray(buckets, np.int64).reshape(-1, 3)
                out[tag + "_state"] = np.asarray(
                    [mod.min_C1, mod.n_perfect_item_C1, len(mod.vals_C1), nflush], np.int64)
            elif algo == "lru":
                out[tag + "_final_order"] = np.asarray(
                    [key_to_pair(k) for k in mod.LRUCache.keys()], np.int64).reshape(-1, 2)
            else:
                fin = []
                for f, lst in enumerate(mod.node_for_freq):
                    if f == 0:
                        continue
                    for key in lst:
                        fin.append((f,) + key_to_pair(key))
                out[tag + "_final_freq"] = np.asarray(fin, np.int64).reshape(-1, 3)
            print(tag, "hits", int(hits.sum()), "perfect", int(hits.all(1).sum()),
                  "flushes", nflush)

    # approximate-embedding mode (EvLFU_C1.py:122-125,142-152): misses turned into hits
    reset_evlfu(EvLFU_C1)
    random.seed(0)
    with contextlib.redirect_stdout(io.StringIO()):
        EvLFU_C1.init(768)
    reqs = reqs_main
    hits = np.zeros((len(reqs), 26), dtype=np.bool_)
    with contextlib.redirect_stdout(io.StringIO()):
        for i, rq in enumerate(reqs):
            h, _ = EvLFU_C1.request_to_ev_lfu([int(v) for v in rq], False, 20)
            hits[i] = h
    out["evlfu_cap768_approx20_hits"] = np.packbits(hits, axis=1)
    buckets = []
    for b in range(27):
        for key in EvLFU_C1.lists_C1[b]:
            buckets.append((b,) + key_to_pair(key))
    out["evlfu_cap768_approx20_final_buckets"] = np.asarray(buckets, np.int64).reshape(-1, 3)
    with contextlib.redirect_stdout(io.StringIO()):
        FR.close(); MFR.close()
    np.savez_compressed(os.path.join(HERE, "cache_traces.npz"), **out)
    print("wrote cache_traces")


# --------------------------------------------------------------------------
# a11: encoders (script/reduce_precision.py) on a fixed grid of values
# --------------------------------------------------------------------------
def gen_encoder_vectors():
    sys.argv = ["reduce_precision.py"]
    import importlib.util
    spec = importlib.util.spec_from_file_location(
        "ref_reduce_precision", os.path.join(REF, "script", "reduce_precision.py"))
    src = open(os.path.join(REF, "script", "reduce_precision.py")).read()
    # import only the function definitions (the file's tail is an argparse main)
    head = src.split("if __name__")[0]
    ns = {}
    exec(compile(head, "reduce_precision_head", "exec"), ns)
    rs = np.random.RandomState(3)
    vals = np.concatenate([
        np.linspace(-1.0, 1.0, 4001),
        rs.uniform(-1, 1, 3000),
        rs.uniform(-0.02, 0.02, 3000),
        np.array([0.0, 1.0, -1.0, 0.65, -0.65, 0.6500001, -0.6500001, 0.25, -0.25, 0.8, -0.8,
                  0.015, -0.015, 0.00025, -0.00025, 1e-9, -1e-9]),
    ]).astype(np.float32).astype(np.float64)
    u16 = np.array([ns["convert_ev_float_to_ushort"](float(v)) for v in vals], np.int64)
    u4 = np.array([ns["convert_to_4bit_int_posit"](float(v)) for v in vals], np.int64)
    u8 = np.array([round(((float(v) + 1) / 2) * 254) for v in vals], np.int64)  # reduce_precision.py:270
    dec4 = np.array([ns["convert_from_4bit_int_posit"](int(c)) for c in range(15)], np.float64)
    np.savez_compressed(os.path.join(HERE, "encoders.npz"), values=vals, u16=u16, u8=u8, u4=u4,
                        u4_decode_table=dec4)
    print("wrote encoders", len(vals))


# --------------------------------------------------------------------------
# a10: decode tables from the COMPILED reference C++ (oracle/_ref/ref_codec_dump)
# --------------------------------------------------------------------------
def gen_codec_tables():
    import subprocess
    orc = os.path.join(os.path.dirname(os.path.dirname(HERE)), "oracle")
    subprocess.check_call(["make", "-s", "-C", orc, "ref"])
    tmp = tempfile.mkdtemp(prefix="evs_codec_")
    outp = os.path.join(tmp, "codec_tables.bin")
    subprocess.check_call([os.path.join(orc, "_ref", "ref_codec_dump"), outp],
                          stdout=subprocess.DEVNULL)
    t = np.fromfile(outp, np.float32)
    assert t.size == 256 + 512 + 65536
    np.savez_compressed(os.path.join(HERE, "codec_tables.npz"), u8=t[:256],
                        u4=t[256:768].reshape(256, 2), u16=t[768:])
    print("wrote codec_tables")


# --------------------------------------------------------------------------
# a9: C1/C2 routing of the COMPILED reference cache manager (oracle/_ref/libcachemanager_ref.so,
# the as-shipped configuration: N_CACHING_LAYER 3, 8-bit C1 + 4-bit C2, TOTAL_SIZE 75425, "48-48-4")
# --------------------------------------------------------------------------
C1C2_ROWS = 12000
C1C2_SEED = 5


def c1c2_tables(orc):
    """Synthetic tables for the two-tier fixture: (raw8, raw4, alt-keys) per table, one RandomState stream."""
    rs = np.random.RandomState(C1C2_SEED)
    out = []
    for k in range(26):
        w = rs.uniform(-1, 1, size=(C1C2_ROWS, 36)).astype(np.float32)
        alt = (rs.randint(0, C1C2_ROWS, size=C1C2_ROWS) * 100 + (k + 1)).astype('>u4')
        out.append((orc.encode_table(w, 8), orc.encode_table(w, 4), alt))
    return rs, out


def gen_c1c2():
    """Runs in a child process (the reference spawns threads that never join)."""
    import subprocess
    code = r"""
import os, sys, ctypes, tempfile, numpy as np
sys.path.insert(0, %r); sys.path.insert(0, %r)
from oracle import oracle as orc
import make_golden as G
root = tempfile.mkdtemp(prefix='evs_c1c2_')
base = root + '/stored_model/criteo_kaggle_all_mmap/epoch-00/'
alt = root + '/stored_model/criteo_kaggle_all/alternative-keys/1000n-euclid003-newrank/binary/'
for d in (base + 'ev-table-8/binary', base + 'ev-table-4/binary', alt):
    os.makedirs(d)
rs, tabs = G.c1c2_tables(orc)
for k, (r8, r4, a) in enumerate(tabs):
    r8.tofile(base + 'ev-table-8/binary/ev-table-%%d.bin' %% (k + 1))
    r4.tofile(base + 'ev-table-4/binary/ev-table-%%d.bin' %% (k + 1))
    a.tofile(alt + 'ev-table-%%d.bin' %% (k + 1))
os.environ['EVS_REF_ROOT'] = root
L = ctypes.CDLL(%r)
L.ev_lookup.argtypes = [ctypes.POINTER(ctypes.c_int)]
L.ev_lookup.restype = ctypes.POINTER(ctypes.c_float)
dec8 = [orc.decode(t[0], 8, 36) for t in tabs]
dec4 = [orc.decode(t[1], 4, 36) for t in tabs]
N, nreq = G.C1C2_ROWS, 11000
reqs = np.zeros((nreq, 26), np.int32)
for i in range(nreq):
    fresh = (i + rs.randint(0, 2, 26)) %% N
    back = rs.randint(0, max(1, min(i, N)), 26)
    reqs[i] = np.where(rs.rand(26) < 0.9, fresh, back)
    if i > 60 and rs.rand() < 0.25:
        reqs[i] = reqs[i - 1 - rs.randint(50)]
        reqs[i] = np.where(rs.rand(26) < 0.08, rs.randint(0, N, 26), reqs[i])
served = np.zeros((nreq, 26), np.uint8)
for i in range(nreq):
    p = L.ev_lookup((ctypes.c_int * 26)(*[int(v) for v in reqs[i]]))
    out = np.ctypeslib.as_array(p, shape=(26, 36)).copy()
    for k in range(26):
        r = reqs[i, k]
        served[i, k] = 8 if np.array_equal(out[k], dec8[k][r]) else (4 if np.array_equal(out[k], dec4[k][r]) else 0)
np.savez_compressed(%r, requests=reqs, served_bits=served, n_rows=np.int64(N), seed=np.int64(G.C1C2_SEED),
                    cap_c1=np.int64((48 * 75425 // 100) * 4), cap_c2=np.int64((48 * 75425 // 100) * 8))
print('c1c2: served 8bit', int((served == 8).sum()), '4bit', int((served == 4).sum()), 'other', int((served == 0).sum()))
sys.stdout.flush(); os._exit(0)
""" % (os.path.dirname(os.path.dirname(HERE)), HERE,
       os.path.join(os.path.dirname(os.path.dirname(HERE)), "oracle", "_ref", "libcachemanager_ref.so"),
       os.path.join(HERE, "c1c2_ref.npz"))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=1800)
    print([l for l in out.stdout.splitlines() if l.startswith("c1c2")], out.stderr[-500:])


# --------------------------------------------------------------------------
# a8: the C++ EvLFU of the Cython build (flush 0.4 / perfect cap 1.0), COMPILED from the reference
# (oracle/_ref/ref_cython_evlfu = cache_algo/EvLFU_C1_Cython/EvLFU.cpp + oracle/ref/ref_cython_evlfu_driver.cpp)
# --------------------------------------------------------------------------
CYTHON_CAPS = (52, 64, 78, 300, 768)


def gen_cython_traces():
    import subprocess
    orc = os.path.join(os.path.dirname(os.path.dirname(HERE)), "oracle")
    subprocess.check_call(["make", "-s", "-C", orc, "ref"])
    n_rows = [min(n, 600) for n in KAGGLE_LN]
    root = tempfile.mkdtemp(prefix="evs_cython_")
    tdir = os.path.join(root, "stored_model", "criteo_kaggle_all_mmap", "epoch-00", "ev-table")
    tabs = make_bin_tables(tdir, n_rows, seed=7)          # the same tables as cache_traces.npz
    streams = {"main": zipf_stream(n_rows, 1500, seed=11),
               "flush": zipf_stream(n_rows, 1200, seed=13, n_hot=3, p_hot=0.7, distinct_hot=True)}
    out = {"n_rows": np.asarray(n_rows, np.int64), "table_seed": np.int64(7),
           "requests": streams["main"], "requests_flush": streams["flush"]}
    for sname, reqs in streams.items():
        for cap in CYTHON_CAPS:
            rq = os.path.join(root, "req.bin")
            with open(rq, "wb") as f:
                f.write(struct.pack("<ii", len(reqs), cap))
                f.write(np.ascontiguousarray(reqs, np.int32).tobytes())
            res = os.path.join(root, "out.bin")
            p = subprocess.run([os.path.join(orc, "_ref", "ref_cython_evlfu"), rq, res], capture_output=True, text=True,
                               env=dict(os.environ, EVS_REF_ROOT=root), timeout=600)
            assert p.returncode == 0, p.stderr
            nflush = p.stdout.count("flushing!")
            raw = open(res, "rb").read()
            n = len(reqs)
            hits = np.frombuffer(raw, np.uint8, n * 26).reshape(n, 26).astype(bool)
            o = n * 26
            rows = np.frombuffer(raw, np.float32, n * 26 * 36, o).reshape(n, 26, 36)
            o += n * 26 * 36 * 4
            state = np.frombuffer(raw, np.int64, 3, o)
            m = int(np.frombuffer(raw, np.int64, 1, o + 24)[0])
            tri = np.frombuffer(raw, np.int64, 3 * m, o + 32).reshape(m, 3)
            for i in range(n):      # the rows are always the true table rows
                for k in range(26):
                    assert np.array_equal(rows[i, k], tabs[k][reqs[i, k]]), (sname, cap, i, k)
            tag = "cython_%s_cap%d" % (sname, cap)
            out[tag + "_hits"] = np.packbits(hits, axis=1)
            out[tag + "_final_buckets"] = tri.copy()
            out[tag + "_state"] = np.asarray(list(state) + [nflush], np.int64)
            print(tag, "hits", int(hits.sum()), "perfect", int(hits.all(1).sum()), "flushes", nflush, "size", int(state[2]))
    np.savez_compressed(os.path.join(HERE, "cython_traces.npz"), **out)
    print("wrote cython_traces")


# --------------------------------------------------------------------------
# a9 siblings: the other precision builds of the COMPILED reference cache manager
# (oracle/_ref/libcachemanager_ref_<layers>-<main>-<secondary>-<TOTAL_SIZE>.so, oracle/Makefile MGR_VARIANTS)
# --------------------------------------------------------------------------
MGR_VARIANTS = ["2-32-16-4000", "2-32-8-4000", "2-32-4-4000", "2-16-8-4000", "2-16-4-4000", "2-8-4-4000",
                "1-32-4-3000", "1-16-4-3000", "1-8-4-3000", "1-4-4-3000"]
VAR_ROWS, VAR_SEED, VAR_NREQ, VAR_BLOCK = 2500, 17, 2000, 100


def variant_tables(orc):
    """(fp32 rows, raw16, raw8, raw4) per table; one RandomState stream; then the request stream."""
    rs = np.random.RandomState(VAR_SEED)
    tabs = []
    for k in range(26):
        w = rs.uniform(-1, 1, size=(VAR_ROWS, 36)).astype(np.float32)
        tabs.append((w, orc.encode_table(w, 16), orc.encode_table(w, 8), orc.encode_table(w, 4)))
    N = VAR_ROWS
    reqs = np.zeros((VAR_NREQ, 26), np.int32)
    for i in range(VAR_NREQ):
        fresh = (i + rs.randint(0, 2, 26)) % N
        back = rs.randint(0, max(1, min(i, N)), 26)
        reqs[i] = np.where(rs.rand(26) < 0.85, fresh, back)
        if i > 30 and rs.rand() < 0.3:
            reqs[i] = reqs[i - 1 - rs.randint(25)]
            reqs[i] = np.where(rs.rand(26) < 0.06, rs.randint(0, N, 26), reqs[i])
    return tabs, reqs


def gen_mgr_variants():
    """One child process per build (global constructors, reader threads that never join)."""
    import subprocess
    root_repo = os.path.dirname(os.path.dirname(HERE))
    subprocess.check_call(["make", "-s", "-C", os.path.join(root_repo, "oracle"), "ref"])
    out = {"n_rows": np.int64(VAR_ROWS), "seed": np.int64(VAR_SEED), "block": np.int64(VAR_BLOCK)}
    for var in MGR_VARIANTS:
        code = r"""
import os, sys, ctypes, re, tempfile, numpy as np
sys.path.insert(0, %r); sys.path.insert(0, %r)
from oracle import oracle as orc
import make_golden as G
root = tempfile.mkdtemp(prefix='evs_var_')
base = root + '/stored_model/criteo_kaggle_all_mmap/epoch-00/'
for sub in ('ev-table', 'ev-table-16', 'ev-table-8', 'ev-table-4'):
    os.makedirs(base + sub + '/binary')
tabs, reqs = G.variant_tables(orc)
for k, (w, r16, r8, r4) in enumerate(tabs):
    for sub, a in (('ev-table', w), ('ev-table-16', r16), ('ev-table-8', r8), ('ev-table-4', r4)):
        a.tofile(base + sub + '/binary/ev-table-%%d.bin' %% (k + 1))
os.environ['EVS_REF_ROOT'] = root
log = root + '/stdout.txt'
fd = os.open(log, os.O_WRONLY | os.O_CREAT)
os.dup2(fd, 1)
L = ctypes.CDLL(%r)
L.ev_lookup.argtypes = [ctypes.POINTER(ctypes.c_int)]
L.ev_lookup.restype = ctypes.POINTER(ctypes.c_float)
dec = {32: [t[0] for t in tabs], 16: [orc.decode(t[1], 16, 36) for t in tabs],
       8: [orc.decode(t[2], 8, 36) for t in tabs], 4: [orc.decode(t[3], 4, 36) for t in tabs]}
order = (%d, %d) if %d == 2 else (%d,)
served = np.zeros(reqs.shape, np.uint8)
for i in range(len(reqs)):
    p = L.ev_lookup((ctypes.c_int * 26)(*[int(v) for v in reqs[i]]))
    o = np.ctypeslib.as_array(p, shape=(26, 36)).copy()
    for k in range(26):
        for b in order:
            if np.array_equal(o[k], dec[b][k][reqs[i, k]]):
                served[i, k] = b
                break
    if (i + 1) %% G.VAR_BLOCK == 0:
        L.print_perfect_hit()
ctypes.CDLL(None).fflush(None)
perfect = [int(m) for m in re.findall(r'Perfect hit\s+= (\d+)', open(log).read())]
np.savez(%r, served=served, perfect=np.asarray(perfect, np.int64), requests=reqs)
os._exit(0)
"""
        L_, M_, S_, T_ = [int(v) for v in var.split("-")]
        tmp_out = os.path.join(tempfile.mkdtemp(prefix="evs_varout_"), "o.npz")
        src = code % (root_repo, HERE, os.path.join(root_repo, "oracle", "_ref", "libcachemanager_ref_%s.so" % var),
                      M_, S_, L_, M_, tmp_out)
        r = subprocess.run([sys.executable, "-c", src], capture_output=True, text=True, timeout=1800)
        assert os.path.exists(tmp_out), (var, r.stderr[-2000:])
        g = np.load(tmp_out)
        tag = "v" + var.replace("-", "_")
        out[tag + "_served"] = g["served"]
        out[tag + "_perfect"] = g["perfect"]
        out["requests"] = g["requests"]
        print(var, "served", {int(b): int((g["served"] == b).sum()) for b in np.unique(g["served"])},
              "perfect/block", g["perfect"][:8].tolist(), "...", int(g["perfect"].sum()))
    np.savez_compressed(os.path.join(HERE, "mgr_variants.npz"), **out)
    print("wrote mgr_variants")


# --------------------------------------------------------------------------
# a12: the alt-key tier APRX_EV driven SINGLE-THREADED through its public methods
# (oracle/_ref/ref_aprx_driver = mixed_precs_caching/aprx_embedding.cpp + oracle/ref/ref_aprx_driver.cpp)
# --------------------------------------------------------------------------
APRX_ROWS, APRX_SEED = 400, 23
APRX_CASES = [(50, 2500), (64, 3000), (257, 6000)]   # (capacity, number of ops)


def aprx_inputs():
    """alt-key tables (uint32: alt_row*100 + alt_table_1based) and one op stream per case, from the seed."""
    rs = np.random.RandomState(APRX_SEED)
    alt = [(rs.randint(0, APRX_ROWS, size=APRX_ROWS) * 100 + rs.randint(1, 27, size=APRX_ROWS)).astype(np.uint32)
           for _ in range(26)]
    cases = []
    for cap, n in APRX_CASES:
        ops = np.zeros((n, 3), np.int32)
        hot = [(int(rs.randint(1, 27)), int(rs.randint(0, APRX_ROWS))) for _ in range(cap)]
        for i in range(n):
            u = rs.rand()
            op = 0 if u < 0.45 else (1 if u < 0.75 else (2 if u < 0.97 else 3))
            if rs.rand() < 0.6:
                t, r = hot[rs.randint(len(hot))]
            else:
                t, r = int(rs.randint(1, 27)), int(rs.randint(0, APRX_ROWS))
            ops[i] = (op, t, r)
        cases.append((cap, ops))
    return alt, cases


def gen_aprx_ops():
    import subprocess
    orc = os.path.join(os.path.dirname(os.path.dirname(HERE)), "oracle")
    subprocess.check_call(["make", "-s", "-C", orc, "ref"])
    alt, cases = aprx_inputs()
    root = tempfile.mkdtemp(prefix="evs_aprx_")
    adir = os.path.join(root, "stored_model", "criteo_kaggle_all", "alternative-keys", "1000n-euclid003-newrank", "binary")
    os.makedirs(adir)
    for k, a in enumerate(alt):
        a.astype(">u4").tofile(os.path.join(adir, "ev-table-%d.bin" % (k + 1)))   # convert_altkeys_to_binary.py:34,49
    out = {"n_rows": np.int64(APRX_ROWS), "seed": np.int64(APRX_SEED)}
    for cap, ops in cases:
        fo = os.path.join(root, "ops.bin")
        with open(fo, "wb") as f:
            f.write(struct.pack("<ii", len(ops), cap))
            f.write(ops.tobytes())
        res = os.path.join(root, "res.bin")
        p = subprocess.run([os.path.join(orc, "_ref", "ref_aprx_driver"), fo, res], capture_output=True, text=True,
                           env=dict(os.environ, EVS_REF_ROOT=root), timeout=600)
        assert p.returncode == 0, (p.returncode, p.stderr[-2000:])
        r = np.fromfile(res, np.uint32)
        qline = p.stdout.split("=" * 52)[-1].split()
        q = np.asarray([[int(v) for v in k.split("-")] for k in qline], np.int64).reshape(-1, 2)
        out["cap%d_ops" % cap] = ops
        out["cap%d_res" % cap] = r
        out["cap%d_queue" % cap] = q
        print("aprx cap", cap, "ops", len(ops), "lookup hits", int(((ops[:, 0] == 1) & (r != 0xffffffff)).sum()),
              "of", int((ops[:, 0] == 1).sum()), "queue", len(q))
    np.savez_compressed(os.path.join(HERE, "aprx_ops.npz"), **out)
    print("wrote aprx_ops")


# --------------------------------------------------------------------------
# a16: calculate_and_write_cdf (dlrm_s_pytorch_C1.py:299-326) on fixed time stamps
# --------------------------------------------------------------------------
def cdf_stamps(n, seed):
    rs = np.random.RandomState(seed)
    lat = rs.gamma(2.0, 0.0004, size=n) + 0.0002
    return [float(v) for v in np.concatenate([[1700000000.0], 1700000000.0 + np.cumsum(lat)])]


def gen_cdf():
    import io, contextlib
    D = import_reference_c1()
    out = {}
    for n, seed in ((5000, 3), (2345, 4)):
        stamps = cdf_stamps(n, seed)
        tmp = tempfile.mkdtemp(prefix="evs_cdf_")
        cwd = os.getcwd()
        os.chdir(tmp)   # the reference then tries ./script/plot_cdf.py, which is absent here: its failure is ignored
        try:
            with contextlib.redirect_stdout(io.StringIO()):
                D.calculate_and_write_cdf(os.path.join(tmp, "cdf"), "evlfu", stamps)
        finally:
            os.chdir(cwd)
        txt = open(os.path.join(tmp, "cdf", "evlfu-cdf.csv")).read()
        out["n%d_seed%d_csv" % (n, seed)] = np.frombuffer(txt.encode(), np.uint8)
        print("cdf", n, "requests ->", txt.count("\n") - 1, "points")
    np.savez_compressed(os.path.join(HERE, "cdf.npz"), **out)
    print("wrote cdf")


def import_reference_c1():
    """dlrm_s_pytorch_C1 (the EVStore fork) -- needs the same stubs as dlrm_s_pytorch plus its cache modules"""
    import_reference()
    cwd = os.getcwd()
    os.chdir(REF)
    try:
        import dlrm_s_pytorch_C1 as D1
    finally:
        os.chdir(cwd)
    return D1


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "cdf":
        gen_cdf()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "aprx":
        gen_aprx_ops()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "variants":
        gen_mgr_variants()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "c1c2":
        gen_c1c2()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "cython":
        gen_cython_traces()
        return
    D, DP, SM, FR, MFR, EvLFU_C1, LRU, LFU = import_reference()
    if len(sys.argv) > 1 and sys.argv[1] == "bench":
        # the reference's own benchmark shape at a reduced row count (bench/dlrm_s_benchmark.sh:20-45: 8 tables, d = 64,
        # --num-indices-per-lookup=100 --num-indices-per-lookup-fixed=true; rows 1 000 000 -> 600, mb 2 048 -> 40)
        gen_dlrm_case(D, DP, "dlrm_bench_shape", [600] * 8, 64, 40, 100, seed=41, n_idx_fixed=True)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "collate":
        # collate_wrapper_criteo_offset (dlrm_data_pytorch.py:397-410) on a raw batch as CriteoDataset.__getitem__ yields it
        # (:372-395: X_int int32 counts after the negative clip, X_cat int32 ids, y): what the loader hands dlrm_wrap
        rs = np.random.RandomState(77)
        B = 50
        x_int = rs.randint(0, 5000, size=(B, 13)).astype(np.int32)
        x_int[rs.rand(B, 13) < 0.3] = 0
        x_int[0, 0] = 2 ** 31 - 2   # (the largest count the +1 in fp32 still takes)
        x_cat = np.stack([rs.randint(0, n, size=B) for n in KAGGLE_LN], axis=1).astype(np.int32)
        y = rs.randint(0, 2, size=B).astype(np.float32)
        X, lS_o, lS_i, Tt = DP.collate_wrapper_criteo_offset([(x_int[i], x_cat[i], y[i]) for i in range(B)])
        np.savez_compressed(os.path.join(HERE, "collate_criteo.npz"), x_int=x_int, x_cat=x_cat, y=y, X=X.numpy(), lS_o=lS_o.numpy(),
                            lS_i=lS_i.numpy(), T=Tt.numpy())
        print("collate_criteo.npz", X.shape, lS_o.shape, lS_i.shape, X.dtype, lS_i.dtype)
        # the Terabyte binary loader: CriteoBinDataset.__getitem__ (script/data_loader_terabyte.py:226-236) is
        # _transform_features (:68-87) over column views of a (B, 40) int32 block of the file
        import torch
        import data_loader_terabyte as DLT
        rec = np.concatenate([y.reshape(-1, 1).astype(np.int32), x_int, x_cat], axis=1).astype(np.int32)
        t = torch.from_numpy(rec).view((-1, 40))
        out = {}
        for rng in (-1, 1000):
            Xb, lo, li, yb = DLT._transform_features(x_int_batch=t[:, 1:14], x_cat_batch=t[:, 14:], y_batch=t[:, 0], max_ind_range=rng,
                                                     flag_input_torch_tensor=True)
            tag = "r%d" % rng if rng > 0 else "all"
            out.update({"X_" + tag: Xb.numpy(), "lS_o_" + tag: lo.numpy(), "lS_i_" + tag: li.contiguous().numpy()})
        np.savez_compressed(os.path.join(HERE, "collate_terabyte.npz"), rec=rec, **out)
        print("collate_terabyte.npz", rec.shape, sorted(out))
        return
    if len(sys.argv) > 1 and sys.argv[1] == "codec":
        gen_codec_tables()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "cache":
        gen_cache_traces(SM, FR, MFR, EvLFU_C1, LRU, LFU)
        return
    # cfg1 exactly: 8 x 10000 x 16, B=128, <=10 idx/lookup (BASELINE.json configs[0]); tables by sha only
    gen_dlrm_case(D, DP, "dlrm_cfg1", [10000] * 8, 16, 128, 10, seed=123, store_tables=False)
    # ragged small case with stored tables, tiny tables, bags 1..10
    gen_dlrm_case(D, DP, "dlrm_ragged_small", [1000, 37, 3, 500, 64, 2, 129, 4096], 16, 33, 10,
                  seed=5)
    # Kaggle-shaped: 26 tables (row counts clipped), d=36, one index per bag
    gen_dlrm_case(D, DP, "dlrm_kaggle_small", [min(n, 500) for n in KAGGLE_LN], 36, 64, 1,
                  seed=9, criteo_layout=True)
    # weighted pooling + interaction including the diagonal
    gen_dlrm_case(D, DP, "dlrm_weighted_itself", [300, 5, 77, 1024], 32, 17, 6, seed=21,
                  itself=True, weighted=True)
    # d=64 / d=128 geometry (Terabyte-shaped scripts use 64 / 128)
    gen_dlrm_case(D, DP, "dlrm_d64", [200, 3000, 11], 64, 20, 4, seed=31)
    gen_dlrm_case(D, DP, "dlrm_d128", [150, 9], 128, 9, 3, seed=33)
    gen_cache_traces(SM, FR, MFR, EvLFU_C1, LRU, LFU)
    gen_encoder_vectors()
    gen_codec_tables()
    gen_c1c2()
    gen_cython_traces()
    gen_mgr_variants()
    gen_aprx_ops()
    gen_cdf()


if __name__ == "__main__":
    main()
