"""Child process of tests/test_gpu_cache.py::test_reference_cabi_ev_lookup: drives the cache-manager
C ABI exactly as cache_algo/cpp_socket_client.py does (the manager is a process-wide singleton)."""
import ctypes
import json
import os
import sys

import numpy as np

root, prec, total = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
layers = int(sys.argv[4]) if len(sys.argv) > 4 else 1
backing = sys.argv[5] if len(sys.argv) > 5 else "pinned"   # host (the default engine) | hbm | pinned (GPU engine)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as orc  # noqa: E402

# plain ctypes, exactly the binding of cache_algo/cpp_socket_client.py:69-83 (no torch import: fast child)
_root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
L = ctypes.CDLL(os.path.join(_root, "ev-store-dlrm_amd", "lib", "libevstore_hip.so"))
L.ev_lookup.argtypes = [ctypes.POINTER(ctypes.c_int)]
L.ev_lookup.restype = ctypes.POINTER(ctypes.c_float)
L.get_ev_values.argtypes = [ctypes.POINTER(ctypes.c_int)]
L.get_ev_values.restype = ctypes.POINTER(ctypes.c_float)
L.evs_manager_perfect_hit.restype = ctypes.c_longlong
L.evs_manager_aprx_hit.restype = ctypes.c_longlong

reqs = np.load(os.path.join(root, "reqs.npy"))
raws = [np.fromfile(os.path.join(root, {32: "ev-table", 16: "ev-table-16", 8: "ev-table-8", 4: "ev-table-4"}[prec],
                                 "binary", "ev-table-%d.bin" % (k + 1)), np.uint8).reshape(-1, 36 * prec // 8)
        for k in range(26)]
os.environ["EVS_EV_TABLE_ROOT"] = root            # zero-argument path: configuration from the environment
os.environ["EVS_MAIN_PRECISION"] = str(prec)
os.environ["EVS_TOTAL_SIZE"] = str(total)
if backing != "default":
    os.environ["EVS_BACKING"] = backing
os.environ["EVS_N_CACHING_LAYER"] = str(layers)
os.environ["EVS_SECONDARY_PRECISION"] = "4"
if layers == 3:
    os.environ["EVS_ALTKEY_DIR"] = os.path.join(root, "altkeys")
    os.environ["EVS_SIZE_PROPORTION"] = "40-40-20"
fp32_tabs = [orc.decode(r, prec, 36) for r in raws]
if layers >= 2:
    raws4 = [np.fromfile(os.path.join(root, "ev-table-4", "binary", "ev-table-%d.bin" % (k + 1)), np.uint8).reshape(-1, 18)
             for k in range(26)]
    dec4 = [orc.decode(r, 4, 36) for r in raws4]
    if layers == 3:
        alt = [np.fromfile(os.path.join(root, "altkeys", "ev-table-%d.bin" % (k + 1)), ">u4").astype(np.uint32) for k in range(26)]
        o = orc.C1C2C3((40 * total // 100) * (32 // prec), (40 * total // 100) * 8, (20 * total // 100) * 36, fp32_tabs, dec4, alt)
    else:
        o = orc.C1C2((total // 2) * (32 // prec), (total // 2) * 8, fp32_tabs, dec4)
else:
    o = orc.EvLFU(total * (32 // prec), fp32_tabs, variant="cpp")
perfect = 0
ok = True
for i, rq in enumerate(reqs):
    ptr = L.ev_lookup((ctypes.c_int * 26)(*[int(v) for v in rq]))
    if not ptr:
        print("ev_lookup returned NULL"); break
    got = np.ctypeslib.as_array(ptr, shape=(26, 36)).copy()
    if layers >= 2:
        _, vals, p = o.request(rq)
        perfect += p
    else:
        hit, vals = o.request(rq)
        perfect += int(hit.all())
    if not np.array_equal(got.view(np.uint32), vals.view(np.uint32)):
        ok = False
        break
same_buf = ctypes.addressof(L.get_ev_values(None).contents) == ctypes.addressof(
    L.ev_lookup((ctypes.c_int * 26)(*[int(v) for v in reqs[0]])).contents)
counter = int(L.evs_manager_perfect_hit())
aprx = (int(L.evs_manager_aprx_hit()), o.c3_state()["n_hit"] if layers == 3 else 0)
L.print_perfect_hit()
after = int(L.evs_manager_perfect_hit())
rc_dead = L.ev_lookup_based_on_list_keys((ctypes.c_int * 26)())
print("RESULT " + json.dumps({"ok": ok, "perfect_oracle": perfect, "counter": counter, "after_print": after,
                              "same_buf": same_buf, "rc_dead": rc_dead, "aprx": aprx}))
