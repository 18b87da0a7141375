#!/usr/bin/env python3
"""Golden vectors for evstore_utils: RUNS the reference's own evstore_utils.py functions in the build container and records
what they wrote (file bytes) and returned in tests/golden/evstore_utils.npz.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_utils.py

Nothing of the reference is stored: the inputs are made here, the outputs are its files' bytes."""
import contextlib
import io
import os
import sys
import tempfile

import numpy as np

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True
sys.path.insert(0, REF)
import evstore_utils as R  # noqa: E402


def main():
    rs = np.random.RandomState(4)
    ln_emb = np.array([1460, 583, 10131227, 2202608, 305, 24, 12517, 633, 3, 93145, 5683, 8351593, 3194, 27, 14992, 5461306, 10, 5652, 2173, 4,
                       7046547, 18, 15, 286181, 105, 142572])
    tfm = {i: i for i in range(1, 27)}
    out = {}
    with tempfile.TemporaryDirectory() as td, contextlib.redirect_stdout(io.StringIO()):
        p = os.path.join(td, R.TRAINING_CONFIG_FILE)
        R.store_training_config(p, tfm, 306969, 51162, ln_emb, 13)
        out["config_bytes"] = np.frombuffer(open(p, "rb").read(), dtype=np.uint8)
        a, b, c, d, e = R.read_training_config(p)
        assert a == tfm and (b, c, e) == (306969, 51162, 13) and np.array_equal(d, ln_emb)
        # workload traces: 7 requests of 26 "table-row" keys
        rows = np.stack([rs.randint(0, n, size=7) for n in ln_emb], axis=1).astype(np.int32)
        work = [[str(k + 1) + "-" + str(int(rows[i, k])) for k in range(26)] for i in range(7)]
        wd = os.path.join(td, "w")
        os.makedirs(wd)
        R.write_inf_workload_to_file(wd, work)
        import gc
        gc.collect()   # (the reference never closes its 26 files)
        for k in range(26):
            out["trace_%d" % (k + 1)] = np.frombuffer(open(os.path.join(wd, "workload-group-%d.csv" % (k + 1)), "rb").read(), dtype=np.uint8)
        out["rows"] = rows
        # load_new_ev_table: 26 small CSV tables (a header line, as the training loop's dump has it: dlrm_s_pytorch.py:1786-1792)
        ed = os.path.join(td, "ev")
        os.makedirs(ed)
        tabs = []
        for k in range(26):
            w = rs.uniform(-1, 1, size=(3 + k % 4, 36)).astype(np.float32)
            tabs.append(w)
            with open(os.path.join(ed, "ev-table-%d.csv" % (k + 1)), "w") as f:
                f.write(",".join(str(i) for i in range(36)) + "\n")
                for r in w:
                    f.write(",".join(str(x) for x in r) + "\n")
            out["csv_%d" % (k + 1)] = np.frombuffer(open(os.path.join(ed, "ev-table-%d.csv" % (k + 1)), "rb").read(), dtype=np.uint8)
        ld = {"state_dict": {}}
        R.load_new_ev_table(ld, ed)
        for k in range(26):
            out["loaded_%d" % (k + 1)] = ld["state_dict"]["emb_l.%d.weight" % k].numpy()
    np.savez_compressed(os.path.join(HERE, "evstore_utils.npz"), **out)
    print("evstore_utils.npz:", len(out), "arrays; trace file 1 holds", int((out["trace_1"] == 10).sum()), "lines")


if __name__ == "__main__":
    main()
