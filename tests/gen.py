"""Seeded synthetic DNA generators shared by tests and bench.py (values 0..3 = A C G T)."""
import numpy as np


def iid(n, seed):
    return np.random.default_rng(seed).integers(0, 4, n, dtype=np.uint8)


def periodic(n, period, seed, mutations=0):
    rng = np.random.default_rng(seed)
    unit = rng.integers(0, 4, period, dtype=np.uint8)
    S = np.tile(unit, n // period + 1)[:n].copy()
    if mutations:
        idx = rng.integers(0, n, mutations)
        S[idx] = rng.integers(0, 4, mutations, dtype=np.uint8)
    return S


def genome_like(n, seed):
    """i.i.d. base layer + telomere-like ends + tandem arrays (3 %) + segmental duplications (5 %, half exact)
    + interspersed repeat families (10 %, 10 % divergence).  Follows SURVEY.md section 8(d)."""
    rng = np.random.default_rng(seed)
    S = rng.integers(0, 4, n, dtype=np.uint8)
    if n < 4096:
        return S

    def mutate(seg, rate):
        if rate <= 0 or seg.size == 0:
            return seg
        mask = rng.random(seg.size) < rate
        seg = seg.copy()
        seg[mask] = rng.integers(0, 4, int(mask.sum()), dtype=np.uint8)
        return seg

    # interspersed repeats: 10 families of 300 bases
    fams = rng.integers(0, 4, (10, 300), dtype=np.uint8)
    n_ins = int(0.10 * n / 300)
    pos = rng.integers(0, max(1, n - 300), n_ins)
    fam = rng.integers(0, 10, n_ins)
    for p, f in zip(pos.tolist(), fam.tolist()):
        S[p:p + 300] = mutate(fams[f], 0.10)
    # segmental duplications
    budget = int(0.05 * n)
    while budget > 0:
        L = int(np.exp(rng.uniform(np.log(1e3), np.log(min(1e5, max(2e3, n / 8))))))
        L = min(L, n // 4)
        src = int(rng.integers(0, n - L))
        dst = int(rng.integers(0, n - L))
        seg = S[src:src + L].copy()
        S[dst:dst + L] = seg if rng.random() < 0.5 else mutate(seg, 0.01)
        budget -= L
    # tandem arrays
    budget = int(0.03 * n)
    units = [1, 2, 3, 4, 5, 6, 12, 171]
    while budget > 0:
        u = units[int(rng.integers(0, len(units)))]
        L = int(min(n // 8, max(200, 200 * (1.0 / max(1e-6, rng.random())) ** 0.7)))
        L = min(L, 1_000_000)
        p = int(rng.integers(0, n - L))
        unit = rng.integers(0, 4, u, dtype=np.uint8)
        S[p:p + L] = mutate(np.tile(unit, L // u + 1)[:L], 0.005)
        budget -= L
    # chromosome ends: (TTAGGG) x 500
    tel = np.tile(np.array([3, 3, 0, 2, 2, 2], dtype=np.uint8), 500)
    nchr = 24
    for c in range(nchr):
        e = (c + 1) * (n // nchr)
        if e - tel.size > 0:
            S[e - tel.size:e] = tel
    return S
