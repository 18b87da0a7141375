#!/usr/bin/env python3
"""Golden vectors for the .bin / alt-key writers: RUNS the reference's own converter scripts
(script/convert_ev_to_binary.py, script/convert_altkeys_to_binary.py) in the build container on small CSVs made
here, and records {input text, where the output landed, output bytes} in tests/golden/converters.npz.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_converters.py

The scripts run unmodified through runpy as __main__ (so their argparse main bodies decide the output placement).
One shim: numpy 2.x no longer has the alias `np.int` the scripts use (`.astype(np.int)`, :113/:138) -- it is set to
the builtin `int`, which is what the alias was.  Nothing of the reference is stored: inputs are generated below.
"""
import io
import os
import runpy
import sys
import tempfile
import contextlib

import numpy as np

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True


def run_ref(script, argv):
    if not hasattr(np, "int"):
        np.int = int  # removed alias (numpy >= 1.24); identical meaning
    old = sys.argv
    sys.argv = [script] + argv
    try:
        with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
            runpy.run_path(os.path.join(REF, "script", script), run_name="__main__")
    finally:
        sys.argv = old


def csv_fp32(rs, n, d, with_key):
    """What dlrm_s_pytorch.py:1787-1792 writes: header 0..d-1, then str(np.float32) cells."""
    vals = rs.uniform(-1, 1, size=(n, d)).astype(np.float32)
    vals[0, :4] = [0.0, -0.0, 1e-30, 3.4e38]
    vals[1, :3] = [0.1, 1.0 / 3.0, -2.5e-7]
    head = [str(i) for i in range(d)] + (["key"] if with_key else [])
    lines = [",".join(head)]
    for r in range(n):
        cells = [str(x) for x in vals[r]] + ([str(r)] if with_key else [])
        lines.append(",".join(cells))
    return "\n".join(lines) + "\n"


def csv_int(rs, n, d, hi):
    vals = rs.randint(0, hi + 1, size=(n, d))
    vals[0, 0], vals[0, 1] = 0, hi
    lines = [",".join(str(i) for i in range(d))] + [",".join(str(int(v)) for v in row) for row in vals]
    return "\n".join(lines) + "\n"


def altkey_txt(rs, n, table):
    # one "tableId-rowId" line per row (the k-NN notebook's output); row ids up to the 10.1 M-row table
    rows = rs.randint(0, 10131227, size=n)
    rows[0], rows[-1] = 0, 10131226
    tabs = np.full(n, table)
    tabs[1] = 26 if table != 26 else 1  # an alt key may point into another table
    return "\n".join("%d-%d" % (t, r) for t, r in zip(tabs, rows)) + "\n"


def main():
    rs = np.random.RandomState(20260303)
    out = {}
    cases = [("fp32_nokey", "fp32", csv_fp32(rs, 37, 36, False)),
             ("fp32_key", "fp32", csv_fp32(rs, 11, 36, True)),
             ("fp32_d16", "fp32", csv_fp32(rs, 5, 16, False)),
             ("u16", "u_short", csv_int(rs, 29, 36, 65535)),
             ("u8", "u_char", csv_int(rs, 31, 36, 254)),
             ("u4_packed", "u_char", csv_int(rs, 23, 18, 238))]
    for name, read_as, text in cases:
        with tempfile.TemporaryDirectory() as td:
            # the reference's layout: <root>/<precision dir>/ev-table-3.csv
            sub = os.path.join(td, "root", "tabs")
            os.makedirs(sub)
            p = os.path.join(sub, "ev-table-3.csv")
            open(p, "w").write(text)
            run_ref("convert_ev_to_binary.py", ["-file", p, "-read_as", read_as])
            found = []
            for dp, _, fs in os.walk(td):
                for f in fs:
                    if f.endswith(".bin"):
                        found.append(os.path.join(dp, f))
            assert len(found) == 1, found
            out[name + "_csv"] = np.frombuffer(text.encode(), np.uint8)
            out[name + "_bin"] = np.frombuffer(open(found[0], "rb").read(), np.uint8)
            out[name + "_relout"] = np.frombuffer(os.path.relpath(found[0], sub).encode(), np.uint8)
            out[name + "_readas"] = np.frombuffer(read_as.encode(), np.uint8)
            print(name, read_as, os.path.relpath(found[0], sub), out[name + "_bin"].size, "bytes")
    with tempfile.TemporaryDirectory() as td:
        names = []
        for t in (1, 3, 26):
            text = altkey_txt(rs, 19 + t, t)
            open(os.path.join(td, "ev-table-%d.csv" % t), "w").write(text)
            out["alt%d_txt" % t] = np.frombuffer(text.encode(), np.uint8)
            names.append(t)
        open(os.path.join(td, "notes.txt"), "w").write("ignored\n")  # no 'ev-table' in its name: skipped
        run_ref("convert_altkeys_to_binary.py", ["-input_folder", td])
        got = sorted(os.listdir(os.path.join(td, "binary")))
        assert got == sorted("ev-table-%d.bin" % t for t in names), got
        for t in names:
            out["alt%d_bin" % t] = np.frombuffer(open(os.path.join(td, "binary", "ev-table-%d.bin" % t), "rb").read(), np.uint8)
            print("altkeys table", t, out["alt%d_bin" % t].size, "bytes")
    out["case_names"] = np.frombuffer(",".join(c[0] for c in cases).encode(), np.uint8)
    np.savez_compressed(os.path.join(HERE, "converters.npz"), **out)
    print("wrote converters.npz")


if __name__ == "__main__":
    main()
