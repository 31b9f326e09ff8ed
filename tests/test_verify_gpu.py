"""kiss_hip_ctx_verify_sa_dev: the reference's own test property (tests/kiss.cpp:26-28: SA[0] = n, permutation, k-order of
adjacent pairs) and, for k >= n, the linear-time exactness proof -- as device kernels that read only the text and the SA.
Includes BASELINE.json configs[1] and configs[3] at FULL size (n = 3 117 292 070) and the reference's own test shapes
(random DNA and 'A'..'D' bytes, 100-200 k and 10-20 M, k = 256) on the HIP path."""
import numpy as np
import pytest

from tests import gen

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_dev():
    import torch
    return torch, torch.device("cuda", 0)


@pytest.fixture(scope="module")
def ctx():
    import kiss_amd
    c = kiss_amd.Context(max_n=20_000_000, device=0)
    yield c
    c.close()


def dev_verify(ctx, torch_dev, S, SA, k):
    torch, dev = torch_dev
    d_S = torch.from_numpy(np.ascontiguousarray(S, dtype=np.uint8)).to(dev)
    d_SA = torch.from_numpy(np.ascontiguousarray(SA, dtype=np.uint32).view(np.int32)).to(dev)
    torch.cuda.synchronize()
    return ctx.verify_sa_dev(d_S.data_ptr() if S.size else 0, S.size, d_SA.data_ptr(), k)


@pytest.mark.parametrize("k", [32, 256, 0xFFFFFFFF])
@pytest.mark.parametrize("shape", ["iid", "genome", "period3", "period171", "allA", "tiny0", "tiny1", "tiny9"])
def test_verifier_accepts_oracle_sa_and_digest_matches_host(ctx, torch_dev, oracle, shape, k):
    from kiss_amd import sorter
    S = {"iid": lambda: gen.iid(200_003, 1), "genome": lambda: gen.genome_like(1_000_000, 2),
         "period3": lambda: gen.periodic(50_000, 3, 3, 5), "period171": lambda: gen.periodic(80_000, 171, 4, 50),
         "allA": lambda: np.zeros(10_000, np.uint8), "tiny0": lambda: np.zeros(0, np.uint8),
         "tiny1": lambda: gen.iid(1, 5), "tiny9": lambda: gen.iid(9, 6)}[shape]()
    SA = oracle.suffix_sort(S, k)
    rep = dev_verify(ctx, torch_dev, S, SA, k)
    assert rep["ok"] == 1 and rep["order_violations"] == 0 and rep["duplicates"] == 0 and rep["out_of_range"] == 0, rep
    assert rep["exact"] == (1 if k >= S.size else 0)
    assert rep["digest"] == sorter.sa_digest(SA)
    if k >= S.size:
        # the exact SA also passes every bounded-k property check
        assert dev_verify(ctx, torch_dev, S, SA, 256)["ok"] == 1 or S.size <= 256


def test_verifier_rejects_what_it_should(ctx, torch_dev, oracle):
    S = gen.genome_like(300_000, 7)
    n = S.size
    SA = oracle.suffix_sort(S, 256)
    # the k = 256 SA of a text with long exact repeats is NOT the exact SA: the proof must fail, the property must hold
    exact = oracle.suffix_sort(S, 0xFFFFFFFF)
    assert not np.array_equal(SA, exact)
    assert dev_verify(ctx, torch_dev, S, SA, 256)["ok"] == 1
    rep = dev_verify(ctx, torch_dev, S, SA, 0xFFFFFFFF)
    assert rep["ok"] == 0 and rep["order_violations"] > 0 and rep["duplicates"] == 0
    # a swapped adjacent pair that differs within k bases
    b = S.tobytes()
    i = next(i for i in range(1000, n) if b[SA[i - 1]:SA[i - 1] + 256] != b[SA[i]:SA[i] + 256])
    bad = SA.copy()
    bad[i - 1], bad[i] = SA[i], SA[i - 1]
    rep = dev_verify(ctx, torch_dev, S, bad, 256)
    assert rep["ok"] == 0 and 1 <= rep["order_violations"] <= 3 and rep["first_violation"] in (i - 1, i, i + 1)
    # a duplicated value (and hence a missing one)
    bad = SA.copy()
    bad[5000] = bad[6000]
    rep = dev_verify(ctx, torch_dev, S, bad, 256)
    assert rep["ok"] == 0 and rep["duplicates"] == 1
    # sentinel not first / out of range
    bad = SA.copy()
    bad[0], bad[1] = SA[1], SA[0]
    rep = dev_verify(ctx, torch_dev, S, bad, 256)
    assert rep["ok"] == 0 and rep["sa0_ok"] == 0
    bad = SA.copy()
    bad[77] = n + 5
    rep = dev_verify(ctx, torch_dev, S, bad, 256)
    assert rep["ok"] == 0 and rep["out_of_range"] == 1
    # a tie at depth k may stand in either order for the property: swapping two entries equal through 256 bases passes
    ties = [i for i in range(1, n) if b[SA[i - 1]:SA[i - 1] + 256] == b[SA[i]:SA[i] + 256]
            and SA[i - 1] + 256 <= n and SA[i] + 256 <= n]
    assert ties
    ok = SA.copy()
    j = ties[0]
    ok[j - 1], ok[j] = SA[j], SA[j - 1]
    assert dev_verify(ctx, torch_dev, S, ok, 256)["ok"] == 1


@pytest.mark.parametrize("n,seed", [(150_000, 101), (15_000_000, 102)])
def test_reference_test_shapes_dna_on_hip_path(ctx, torch_dev, n, seed):
    """tests/kiss.cpp "kISS-1 DNA" / "kISS-1 DNA large": random bases, k = 256, the REQUIRE evaluated for every i"""
    torch, dev = torch_dev
    d_S = torch.from_numpy(gen.iid(n, seed)).to(dev)
    d_SA = torch.empty(n + 1, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    ctx.suffix_sort_dev(d_S.data_ptr(), n, d_SA.data_ptr(), k=256)
    rep = ctx.verify_sa_dev(d_S.data_ptr(), n, d_SA.data_ptr(), 256)
    assert rep["ok"] == 1, rep


@pytest.mark.parametrize("n,seed", [(150_000, 103), (15_000_000, 104)])
def test_reference_test_shapes_general_alphabet_on_hip_path(torch_dev, n, seed):
    """tests/kiss.cpp "kISS-1 general" / "general large": bytes 'A'..'D' (65..68), k = 256 property; the HIP entry
    returns the exact SA, so the exactness proof must hold as well"""
    import ctypes
    import kiss_amd
    from kiss_amd import _lib
    torch, dev = torch_dev
    S = (np.random.default_rng(seed).integers(65, 69, n)).astype(np.uint8)
    d_S = torch.from_numpy(S).to(dev)
    d_SA = torch.empty(n + 1, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    with kiss_amd.Context(max_n=n, device=0) as c:
        rc = _lib.load().kiss_hip_ctx_suffix_sort_u8_dev(c._ctx, ctypes.c_void_p(d_S.data_ptr()), n,
                                                         ctypes.c_void_p(d_SA.data_ptr()), None)
        assert rc == 0
        assert c.verify_sa_dev(d_S.data_ptr(), n, d_SA.data_ptr(), 256)["ok"] == 1
        assert c.verify_sa_dev(d_S.data_ptr(), n, d_SA.data_ptr(), 0xFFFFFFFF)["ok"] == 1


# ---- full size: BASELINE.json configs[1] and configs[3] ---------------------------------------------------------
CHM13_N = 3_117_292_070


@pytest.fixture(scope="module")
def full_size(torch_dev):
    import bench
    import kiss_amd
    torch, dev = torch_dev
    S = bench.gen_text_device(CHM13_N, 2, dev)  # the bench text (seed 2)
    SA = torch.empty(CHM13_N + 1, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    c = kiss_amd.Context(max_n=CHM13_N, device=0)
    yield c, S, SA
    c.close()
    del S, SA
    torch.cuda.empty_cache()


def test_full_size_k256_verified_on_device(full_size):
    import json
    import os
    c, S, SA = full_size
    c.suffix_sort_dev(S.data_ptr(), CHM13_N, SA.data_ptr(), k=256)
    rep = c.verify_sa_dev(S.data_ptr(), CHM13_N, SA.data_ptr(), 256)
    assert rep["ok"] == 1, rep
    pins = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "full_size_pins.json")))
    assert "%016x" % rep["digest"] == pins["chm13size_seed2_k256"]["digest"], rep


def test_full_size_exact_prefix_doubling_verified_on_device(full_size):
    import kiss_amd
    c, S, SA = full_size
    c.suffix_sort_dev(S.data_ptr(), CHM13_N, SA.data_ptr(), k=0xFFFFFFFF, algo=kiss_amd.ALGO_PREFIX_DOUBLING)
    rep = c.verify_sa_dev(S.data_ptr(), CHM13_N, SA.data_ptr(), 0xFFFFFFFF)
    assert rep["ok"] == 1 and rep["exact"] == 1, rep  # a proof, not a comparison of two of our own paths
