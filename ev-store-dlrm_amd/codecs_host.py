"""Host-side (numpy) decode of ONE row for the storage manager's direct reads.
Restates mixed_precs_caching/evlfu_8.cpp:370-378, evlfu_4.cpp:319-341 (+ evlfu_4.hpp:46),
evlfu_16.cpp:332-356.  The bulk paths decode on the GPU (csrc/evs_common.h)."""
import numpy as np

_U4 = np.array([1, 0.8, 0.6, 0.4, 0.0625, 0.00390625, 0.0000153, 0, -0.0000153, -0.00390625, -0.0625, -0.4, -0.6,
                -0.8, -1, np.nan], dtype=np.float32)


def decode_row(blob, bits, d):
    b = np.frombuffer(blob, dtype=np.uint8)
    if bits == 8:
        return ((b.astype(np.float32) / np.float32(254)) * np.float32(2)) - np.float32(1)
    if bits == 4:
        out = np.empty(d, np.float32)
        out[0::2] = _U4[b >> 4]
        out[1::2] = _U4[b & 15]
        return out
    if bits == 16:
        v = b.view(np.uint16)
        f = v.astype(np.float32)
        lo = (f.astype(np.float64) * 0.00002 - 0.65).astype(np.float32)
        diff = ((v.astype(np.int32) - 65000).astype(np.float32) / np.float32(100))
        hi = (0.65 + diff.astype(np.float64))
        hi = np.where(v % 2 == 1, -hi, hi).astype(np.float32)
        return np.where(v > 65000, hi, lo)
    raise ValueError("bits must be 16, 8 or 4")
