#!/usr/bin/env python3
"""The two places where workgroups wait for each other, on a stream that may use only a few CUs (hipExtStreamCreateWithCUMask):
(a) the second launch of the two-launch path with rows left to it (tie-heavy rows and NaN rows at config 2's width), (b) the
large-n route's fall-back kernel with overflowing rows (chunk sort -> meeting -> chunk search).  Both bound their waiters by
the CUs the stream reports (hipExtStreamGetCUMask); totals against the rank-sort oracle.  usage: cu_mask_check.py [cus ...]
(default 8 and 40; multiples of 8).  GPU box only; run it under `timeout`: a wait that never ends would be the bug this looks for."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import oracle
from statdepth_amd import engine

hip = ctypes.CDLL("libamdhip64.so")
hip.hipExtStreamCreateWithCUMask.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint32)]
hip.hipExtStreamCreateWithCUMask.restype = ctypes.c_int


def masked_stream(cus):
    words = (ctypes.c_uint32 * 8)()
    for c in range(cus):                          # the first `cus` CUs
        words[c // 32] |= 1 << (c % 32)
    h = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(h), 8, words)
    assert rc == 0 and h.value, f"hipExtStreamCreateWithCUMask: {rc}"
    return torch.cuda.ExternalStream(h.value)


rng = np.random.default_rng(5)
# (a) config 2's width: tie-heavy rows overflow the 32-bit kernel's lists, a NaN row is flagged outright
Xa = rng.normal(size=(600, 10000)).cumsum(axis=0)
Xa[::3] = np.round(Xa[::3], 1)
Xa[7, ::13] = np.nan
wa = oracle.mbd_counts_ranksort(Xa, 2)
# (b) large-n route: two rows with most of their keys on one value, flagged buckets beside them
Xb = rng.normal(size=(5, 60000)).cumsum(axis=0)
Xb[1, rng.random(60000) < 0.7] = 0.25
Xb[3, rng.random(60000) < 0.5] = -1.0
Xb[2] = np.round(Xb[2], 1)
wb = oracle.mbd_counts_ranksort(Xb, 2)
torch.cuda.init()
print("inputs ready", flush=True)
# The mask's bits interleave over the 8 XCDs of the device (bit i: XCD i % 8): a mask that leaves an XCD without a CU starves
# every kernel with more workgroups than enabled XCDs, whoever wrote it -- so only multiples of 8 here (8: one CU per XCD).
for cus in [int(a) for a in sys.argv[1:]] or [8, 40]:
    assert cus % 8 == 0, "whole CUs per XCD"
    st = masked_stream(cus)
    with torch.cuda.stream(st):
        z = torch.ones(1 << 22, device="cuda")              # a plain kernel with many workgroups first: does the stream run at all?
        z = (z * 2).sum().item()
        print(f"{cus} CUs: stream runs ({z:.0f})", flush=True)
        for name, X, want in (("two-launch path", Xa, wa), ("large-n fall-backs", Xb, wb)):
            Xd = engine.to_device_matrix(X)
            for _ in range(2):
                got = engine.mbd_counts(Xd, None, 2, algo="rank")
                assert (got == want).all(), (cus, name)
            print(f"{cus} CUs: {name} ok", flush=True)
print("CU MASK OK")
