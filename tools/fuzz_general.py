#!/usr/bin/env python3
"""Fuzzing of the general-alphabet sorter (kiss_hip_suffix_sort_u8): random byte texts (alphabet size, repeats, runs,
periods) checked with the linear-time suffix-array test, small ones also against plain Python sorting.
Usage: fuzz_general.py [seconds] [seed]"""
import os
import sys
import time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kiss_amd

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)


def is_suffix_array(S, sa):
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


t0 = time.time()
cases = fails = 0
while time.time() - t0 < budget:
    n = int(np.exp(rng.uniform(np.log(1), np.log(2_000_000))))
    sigma = int(rng.choice([1, 2, 3, 4, 5, 16, 64, 255, 256]))
    kind = int(rng.integers(0, 4))
    if kind == 0:
        S = rng.integers(0, sigma, n, dtype=np.uint8)
    elif kind == 1:  # periodic with a few mutations
        u = rng.integers(0, sigma, int(rng.integers(1, 600)), dtype=np.uint8)
        S = np.tile(u, n // u.size + 1)[:n].copy()
        idx = rng.integers(0, n, int(rng.integers(0, 8)))
        S[idx] = rng.integers(0, sigma, idx.size, dtype=np.uint8)
    elif kind == 2:  # runs
        lens = np.maximum(1, (rng.pareto(1.1, max(1, n // 10)) * 4).astype(np.int64))
        S = np.repeat(rng.integers(0, sigma, lens.size, dtype=np.uint8), lens)[:n]
        if S.size < n:
            S = np.concatenate([S, rng.integers(0, sigma, n - S.size, dtype=np.uint8)])
    else:  # copies of long blocks
        S = rng.integers(0, sigma, n, dtype=np.uint8)
        for _ in range(int(rng.integers(1, 6))):
            if n > 20:
                L = int(rng.integers(1, n // 2))
                a, c = int(rng.integers(0, n - L)), int(rng.integers(0, n - L))
                S[c:c + L] = S[a:a + L].copy()
    if rng.random() < 0.3 and n > 2:  # values at the ends of the byte range
        S = (S.astype(np.int64) + (256 - sigma)).astype(np.uint8) if sigma < 256 else S
    S = np.ascontiguousarray(S, dtype=np.uint8)
    sa = kiss_amd.suffix_array_bytes(S.tobytes())
    ok = is_suffix_array(S, np.asarray(sa))
    if ok and n <= 600:
        b = S.tobytes()
        ok = np.array_equal(sa, np.array([n] + sorted(range(n), key=lambda i: b[i:]), dtype=np.uint32))
    cases += 1
    if not ok:
        fails += 1
        print("MISMATCH n=%d sigma=%d kind=%d" % (n, sigma, kind), flush=True)
print("fuzz_general: %d texts, %d failures, %.0f s, seed %d" % (cases, fails, time.time() - t0, seed), flush=True)
sys.exit(1 if fails else 0)
