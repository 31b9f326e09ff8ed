#!/usr/bin/env python3
"""Times the library's LSD radix sort (5 passes on bits 24..63) on random keys through the debug hook; prints the
kernel-class times the library measured.  Usage: radix_probe.py [items]"""
import ctypes
import sys
import numpy as np
sys.path.insert(0, __file__.rsplit("/", 2)[0])
import kiss_amd
from kiss_amd import _lib
count = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
ctx = kiss_amd.Context(max_n=int(count / 0.3), profiling=True)
rng = np.random.default_rng(1)
keys = rng.integers(0, 1 << 63, count, dtype=np.int64).view(np.uint64)
pos = np.arange(count, dtype=np.uint32)
lib = _lib.load()
for rep in range(2):
    k, p = keys.copy(), pos.copy()
    rc = lib.kiss_hip_debug_radix_sort(ctx._ctx, k.ctypes.data, p.ctypes.data, count, 24)
    assert rc == 0, rc
ok = bool(np.all(np.diff((k >> np.uint64(24)).astype(np.int64)) >= 0))
st = ctx.stats()["kernels"]
print("items %d sorted %s: radix_scatter %.2f ms / %d launches, radix_hist %.2f ms" % (
    count, ok, st["radix_scatter"]["ms"], st["radix_scatter"]["launches"], st["radix_hist"]["ms"]))
