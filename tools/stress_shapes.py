#!/usr/bin/env python3
"""Timing + parity of adversarial text shapes at moderate size (guards against pathological slowness)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kiss_amd
from tests import gen, oracle_binding
orc = oracle_binding.load()


def sa_is_exact(S, sa):
    """linear-time suffix array check: a permutation, and (S[a], rank[a+1]) strictly increasing along SA"""
    n = S.size
    if sa.size != n + 1 or sa[0] != n:
        return False
    isa = np.full(n + 2, -1, np.int64)
    isa[sa] = np.arange(n + 1)
    if (isa[:n + 1] < 0).any():
        return False
    a, b = sa[1:-1].astype(np.int64), sa[2:].astype(np.int64)
    ka = S[a].astype(np.int64) * (n + 2) + isa[a + 1]
    kb = S[b].astype(np.int64) * (n + 2) + isa[b + 1]
    return bool((ka < kb).all())


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
    for k, algo in ((256, 0), (0xFFFFFFFF, 0), (0xFFFFFFFF, 1)):
        if k != 256 and algo == 0 and name in ("period7", "period400", "AC"):
            continue  # the MSD path compares 32 bases per round: n/32 rounds on a fully periodic text
        t = time.time()
        sa = ctx.suffix_sort(S, k, algo=algo)
        dt = time.time() - t
        st = ctx.stats()
        if k != 256 and name in ("period7", "period400", "AC"):
            ok = sa_is_exact(S, sa)  # the CPU oracle's comparison sort is quadratic here
        else:
            ok = bool(np.array_equal(sa, orc.suffix_sort(S, k)))
        print("%-10s k=%-10d algo %d %8.1f ms device %8.1f ms (refine %6.1f)  rounds %3d+%2d passes %5d tied %9d  parity %s" % (
            name, k, algo, dt * 1e3, st["ms_total"], st["ms_refine"], st["lms_rounds"], st["doubling_rounds"],
            st["induce_passes"], st["refine_items"], ok), flush=True)
