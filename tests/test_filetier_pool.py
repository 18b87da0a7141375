"""The file tier's reader pool (csrc/evs_filetier.hip, round 3: persistent prefetching threads) on the CPU: rows fetched out of the
mmap'ed ev-table-N.bin files (emb_storage/mmap_file_read.py:32-40: row r at byte row_bytes * r) equal the file bytes,
invalid keys leave their slots untouched, repeated fetches and several thread counts give the same bytes.  No GPU: a
tier with no pinned budget registers nothing."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import sys, numpy as np
sys.path.insert(0, %r)
import evstore_dlrm_amd as E
d, rb = 36, 144
rs = np.random.RandomState(4)
n_rows = [5000, 3, 70000, 1, 900, 12000]
tabs = [rs.randint(0, 256, size=(n, rb)).astype(np.uint8) for n in n_rows]
paths = []
for k, t in enumerate(tabs):
    p = sys.argv[1] + "/ev-table-%%d.bin" %% (k + 1)
    t.tofile(p); paths.append(p)
tier = E.FileTier(paths, rb, 0)
assert tier.n_rows == n_rows and not any(tier.registered)
for n in (1, 100, 2047, 2048, 5000, 60000):
    t = rs.randint(0, len(n_rows), size=n)
    r = np.array([rs.randint(0, n_rows[k]) for k in t], dtype=np.uint64)
    keys = ((t.astype(np.uint64) + np.uint64(1)) << np.uint64(32)) | r
    bad = rs.rand(n) < 0.05
    keys[bad] = (np.uint64(99) << np.uint64(32)) | np.uint64(5)            # no such table
    oob = (~bad) & (rs.rand(n) < 0.03)
    keys[oob] = ((t[oob].astype(np.uint64) + np.uint64(1)) << np.uint64(32)) | np.uint64(10**9)   # row past the end
    for rep in range(3):
        got = tier.fetch(keys)
        want = np.zeros((n, rb), np.uint8)
        ok = ~(bad | oob)
        for i in np.flatnonzero(ok):
            want[i] = tabs[t[i]][int(r[i])]
        assert np.array_equal(got, want), (n, rep)
tier.close()
print("POOL_OK")
'''


@pytest.mark.parametrize("threads", ["1", "3", "16"])
def test_reader_pool_fetches_the_file_rows(tmp_path, threads):
    env = dict(os.environ, EVS_FILETIER_THREADS=threads)
    r = subprocess.run([sys.executable, "-c", CHILD % ROOT, str(tmp_path)], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and "POOL_OK" in r.stdout, (r.stdout[-1000:], r.stderr[-3000:])
