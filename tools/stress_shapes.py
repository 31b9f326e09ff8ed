#!/usr/bin/env python3
"""Timing + parity of adversarial text shapes at moderate size (guards against pathological slowness)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kiss_amd
from tests import gen, oracle_binding
orc = oracle_binding.load()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20_000_000
ctx = kiss_amd.Context(max_n=n)
rng = np.random.default_rng(1)
shapes = {
    "iid": gen.iid(n, 1),
    "allA": np.zeros(n, np.uint8),
    "allT": np.full(n, 3, np.uint8),
    "AC": np.tile(np.array([0, 1], np.uint8), n // 2),
    "period7": gen.periodic(n, 7, 3, 50),
    "period400": gen.periodic(n, 400, 4, 50),
    "long_runs": gen.iid(n, 2),
    "genome": gen.genome_like(n, 5),
}
S = shapes["long_runs"]
for i in range(200):
    p = int(rng.integers(0, n - 200_000)); S[p:p + int(rng.integers(1000, 100_000))] = i % 4
for name, S in shapes.items():
    for k in (256, 0xFFFFFFFF):
        if k != 256 and name in ("period7", "period400", "AC"):
            continue  # exact order on a fully periodic text needs n/32 rounds (documented limitation)
        t = time.time()
        sa = ctx.suffix_sort(S, k)
        dt = time.time() - t
        st = ctx.stats()
        ref = orc.suffix_sort(S, k)
        print("%-10s k=%-10d %8.1f ms device %8.1f ms  rounds %4d passes %5d  parity %s" % (
            name, k, dt * 1e3, st["ms_total"], st["lms_rounds"], st["induce_passes"], bool(np.array_equal(sa, ref))), flush=True)
