"""Miss-path storage manager -- mirror of the reference's emb_storage/storage_manager.py.

Same module-level surface (storage_type, ev_precs, training_config_path, EmbStorage,
get_val_from_storage, get_arr_val_from_storage, request_to_emb_storage,
load_ev_table_into_emb_stor, close_any_db_conn).  Back-ends kept: the plain-file and mmap
readers over the reference's on-disk format (emb_storage/file_read.py:27-33,
mmap_file_read.py:32-40: ev-table-{1..26}.bin, row r at byte 144*r), plus two device-side
tiers the GPU cache reads misses from directly: HBM (tables resident on the GPU) and
PINNED (tables in pinned host memory mapped into the GPU = the host-mmap miss tier).
RocksDB / SQLite / socket back-ends are out of scope (third-party engines).
"""
import mmap
import os
import struct

import numpy as np
import torch

BINARY_DIR_NAME = "binary/"
N_EV_TABLE = 26
EV_DIMENSION = 36


class EmbStorage:
    DUMMY = 1
    ROCKSDB = 2
    FILEPY = 3
    MMAPFILEPY = 4
    SQLITE = 5
    FILEC = 6
    CPP_CACHING_LAYER = 7
    HBM = 8      # MI355X-native: tables resident in HBM
    PINNED = 9   # MI355X-native: tables in pinned host memory, read by the GPU over PCIe on a miss


storage_type = EmbStorage.DUMMY
ev_precs = 32
training_config_path = "/should/point/to/training_config_path"

_files = []     # FILEPY: index 0 unused, as in the reference
_mmaps = []     # MMAPFILEPY
_tables = None  # DUMMY / HBM / PINNED: list of uint8 tensors (raw rows)
_device = "cuda"


def _row_bytes():
    return EV_DIMENSION * ev_precs // 8


def _bin_path(ev_path, k):
    p = os.path.join(ev_path, BINARY_DIR_NAME, "ev-table-%d.bin" % (k + 1))
    if not os.path.exists(p):
        p = os.path.join(ev_path, "ev-table-%d.bin" % (k + 1))
    return p


def _decode(blob):
    if ev_precs == 32:
        return struct.unpack('f' * EV_DIMENSION, blob)  # file_read.py:33
    from .. import codecs_host
    return tuple(codecs_host.decode_row(blob, ev_precs, EV_DIMENSION))


def load_ev_table_into_emb_stor(ev_path_c1, overwrite_db=True):
    """storage_manager.py:141-170."""
    global _files, _mmaps, _tables
    if storage_type == EmbStorage.FILEPY:
        _files = ["ID Zero is not being used!"] + [open(_bin_path(ev_path_c1, k), "rb") for k in range(N_EV_TABLE)]
    elif storage_type == EmbStorage.MMAPFILEPY:
        _files = ["ID Zero is not being used!"] + [open(_bin_path(ev_path_c1, k), "rb") for k in range(N_EV_TABLE)]
        _mmaps = ["ID Zero is not being used!"] + [mmap.mmap(f.fileno(), 0, prot=mmap.PROT_READ) for f in _files[1:]]
    elif storage_type in (EmbStorage.DUMMY, EmbStorage.HBM, EmbStorage.PINNED):
        tabs = []
        for k in range(N_EV_TABLE):
            a = np.fromfile(_bin_path(ev_path_c1, k), dtype=np.uint8)
            if a.size % _row_bytes():
                print("ERROR: %s is not a whole number of rows" % _bin_path(ev_path_c1, k))
                exit(-1)
            t = torch.from_numpy(a)
            if storage_type == EmbStorage.HBM:
                t = t.to(_device)
            elif storage_type == EmbStorage.PINNED:
                t = t.pin_memory()
            tabs.append(t)
        _tables = tabs
    else:
        print("ERROR: Type of Embedding Storage is invalid! (RocksDB/SQLite/FILEC are not part of this build)")
        exit(-1)


def use_device_tables(tables, precs=32, storage=None):
    """MI355X-native shortcut for callers that already hold the tables (EVTables.raw, or raw uint8 tensors in HBM / pinned
    host memory): the same state load_ev_table_into_emb_stor leaves behind, without going through .bin files."""
    global storage_type, ev_precs, _tables
    raws = tables.raw if hasattr(tables, "raw") else list(tables)
    assert len(raws) == N_EV_TABLE
    ev_precs = precs
    storage_type = storage if storage is not None else (EmbStorage.HBM if raws[0].is_cuda else EmbStorage.PINNED)
    _tables = [t.reshape(-1) if t.dtype == torch.uint8 else t.view(torch.uint8).reshape(-1) for t in raws]


def device_tables():
    """Raw row tensors the GPU cache can read misses from (HBM or PINNED storage only)."""
    if storage_type not in (EmbStorage.HBM, EmbStorage.PINNED) or _tables is None:
        return None
    return _tables


def host_tables():
    """Host-readable row arrays for the host cache engine (csrc/evs_hostcache.hip): the DUMMY / PINNED tensors as they
    are, the FILEPY / MMAPFILEPY files mapped read-only (np.memmap: a miss reads one row through the page cache, what
    file_read.py:27-33 / mmap_file_read.py:32-40 do per row).  None when the tables live in HBM only."""
    if storage_type in (EmbStorage.DUMMY, EmbStorage.PINNED) and _tables is not None:
        return [t.numpy() for t in _tables]
    if storage_type in (EmbStorage.FILEPY, EmbStorage.MMAPFILEPY) and len(_files) > 1:
        return [np.memmap(f.name, dtype=np.uint8, mode="r") if os.path.getsize(f.name) else np.zeros(0, np.uint8)
                for f in _files[1:]]
    return None


def get_val_from_storage(tableId, rowId):
    """storage_manager.py:73-94; tableId is 1-based."""
    rb = _row_bytes()
    if storage_type == EmbStorage.FILEPY:
        f = _files[tableId]
        f.seek(rb * rowId)
        return _decode(f.read(rb))
    if storage_type == EmbStorage.MMAPFILEPY:
        m = _mmaps[tableId]
        m.seek(rb * rowId)
        return _decode(m.read(rb))
    if storage_type in (EmbStorage.DUMMY, EmbStorage.HBM, EmbStorage.PINNED):
        t = _tables[tableId - 1]
        return _decode(bytes(t[rb * rowId: rb * (rowId + 1)].cpu().numpy()))
    print("ERROR: Type of Embedding Storage is invalid!")
    exit(-1)


def get_arr_val_from_storage(keys):
    """storage_manager.py:96-123."""
    return [get_val_from_storage(tableId, rowId) for tableId, rowId in keys]


def request_to_emb_storage(group_rowIds, use_gpu=False):
    """storage_manager.py:125-139: bypass the cache."""
    emb_weights = []
    for i, rowId in enumerate(group_rowIds):
        val = get_val_from_storage(i + 1, int(rowId))
        ev_tensor = torch.FloatTensor([val])
        ev_tensor.requires_grad = True
        if use_gpu:
            ev_tensor = ev_tensor.to(torch.device("cuda:0"))
        emb_weights.append(ev_tensor)
    return -1, emb_weights


def close_any_db_conn():
    """storage_manager.py:172-192."""
    global _files, _mmaps, _tables
    for f in _files[1:]:
        f.close()
    _files, _mmaps, _tables = [], [], None
    print("All db connections are closed!")
