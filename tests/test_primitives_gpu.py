"""GPU unit tests of the sorting / scanning primitives behind the C ABI's test hooks."""
import ctypes

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import kiss_amd
    c = kiss_amd.Context(max_n=8_000_000, device=0)
    yield c
    c.close()


def radix(ctx, keys, lo):
    import kiss_amd
    lib = kiss_amd.load()
    k = np.ascontiguousarray(keys, dtype=np.uint64).copy()
    p = np.arange(k.size, dtype=np.uint32)
    rc = lib.kiss_hip_debug_radix_sort(ctx._ctx, k.ctypes.data, p.ctypes.data, k.size, lo)
    assert rc == 0
    return k, p


@pytest.mark.parametrize("n", [1, 2, 63, 64, 65, 4095, 4096, 4097, 100_000, 262_144, 262_145, 700_001, 2_000_000])
@pytest.mark.parametrize("kind", ["uniform", "few", "constant", "steps"])
def test_radix_sort_is_a_stable_sort(ctx, n, kind):
    rng = np.random.default_rng(n)
    if kind == "uniform":
        keys, lo = rng.integers(0, 1 << 63, n, dtype=np.uint64) << np.uint64(1), 24
    elif kind == "few":
        keys, lo = rng.integers(0, 5, n, dtype=np.uint64) << np.uint64(48), 48
    elif kind == "constant":
        keys, lo = np.full(n, 7 << 56, dtype=np.uint64), 46
    else:  # what the chain collapse sorts: small step indexes in bits 48.., heavily duplicated
        keys, lo = (1 + rng.geometric(0.7, n).astype(np.uint64) % 300) << np.uint64(48), 48
    k, p = radix(ctx, keys, lo)
    shift = np.uint64(lo & ~7)
    order = np.argsort(keys >> shift, kind="stable").astype(np.uint32)
    assert np.array_equal(p, order)
    assert np.array_equal(k, keys[order])


@pytest.mark.parametrize("n", [1, 5, 4096, 4097, 131_841, 1_000_003])
def test_scan(ctx, n):
    import kiss_amd
    lib = kiss_amd.load()
    a = np.random.default_rng(n).integers(0, 1000, n, dtype=np.uint32)
    d = a.copy()
    assert lib.kiss_hip_debug_scan_u32(ctx._ctx, d.ctypes.data, n) == 0
    want = np.concatenate([[0], np.cumsum(a, dtype=np.uint64)[:-1]]).astype(np.uint32)
    assert np.array_equal(d, want)


@pytest.mark.parametrize("items,maxrun", [(795, 2420), (113341, 16), (5000, 300), (40, 60000)])
def test_radix_sort_sawtooth_steps(ctx, items, maxrun):
    # exactly what the chain collapse sorts: for every item the step indexes 1..r, item-major
    rng = np.random.default_rng(items)
    r = rng.integers(0, maxrun + 1, items)
    keys = np.concatenate([np.arange(1, x + 1, dtype=np.uint64) for x in r]) << np.uint64(48)
    k, p = radix(ctx, keys, 48)
    order = np.argsort(keys >> np.uint64(48), kind="stable").astype(np.uint32)
    assert np.array_equal(p, order)
