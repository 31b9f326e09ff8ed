"""Pins the oracle (oracle/kiss_oracle.c, our restatement) against the REFERENCE'S OWN CODE compiled unmodified into
oracle/_ref/libkiss_ref.so (oracle/ref_driver.cpp): get_lms, the PackedDNAString loads the comparator is built from,
put_lms_suffix and the induction sweeps.  CPU only.  What stays restated on both sides (the control flow of
lms_suffix_direct_sort_dna, kiss1_core.hpp:41-144 -- that header needs spdlog) is cross-checked between two
independent restatements (C byte compares in the oracle, AVX2 block compares on the reference's packed loads in the
driver)."""
import numpy as np
import pytest

from tests import gen, ref_binding

pytestmark = pytest.mark.skipif(not ref_binding.available(), reason="oracle/_ref not built and /root/reference absent")

KS = [32, 256, 0xFFFFFFFF]


def shapes():
    yield "iid_100003", gen.iid(100_003, 1)
    yield "iid_20000", gen.iid(20_000, 2)
    yield "genome_300k", gen.genome_like(300_000, 3)
    yield "genome_1M", gen.genome_like(1_000_000, 4)
    for p in (1, 2, 3, 5, 7, 37, 400):
        yield "periodic_%d" % p, gen.periodic(30_000, p, 10 + p, mutations=4)
    yield "all_A", np.zeros(20_000, np.uint8)
    yield "all_T", np.full(12_345, 3, np.uint8)
    yield "AC_n", np.tile(np.array([0, 1], np.uint8), 10_000)
    for cut in (124, 125, 126, 255, 256, 257, 374, 375, 376, 500):
        # text ending inside a repeat: a copy of the head at the very end, cut after `cut` bases (end-of-text rule)
        base = gen.iid(20_000, 50 + cut)
        yield "endrepeat_%d" % cut, np.concatenate([base, base[:cut]])
    for n in (1, 2, 3, 5, 16, 50, 1000):
        yield "tiny_%d" % n, gen.iid(n, 70 + n)


SHAPES = list(shapes())


@pytest.fixture(scope="module")
def ref():
    return ref_binding.load()


@pytest.mark.parametrize("name,S", SHAPES, ids=[s[0] for s in SHAPES])
def test_get_lms_equals_reference(oracle, ref, name, S):
    lms_o, hist_o = oracle.get_lms(S)
    for T in (1, 3, 8):
        lms_r, hist_r = ref.get_lms(S, T)
        assert np.array_equal(lms_o, lms_r), (name, T)
        # the reference's per-thread buckets summed over the threads (kiss_common.hpp:431-436)
        assert np.array_equal(hist_o[:, :4], hist_r[:, :4]), (name, T)


def key_of(S, p, L):
    seg = np.zeros(L, np.uint8)
    part = S[p:p + L]
    seg[:part.size] = part
    return seg


def test_prefix10_load_is_first_10_bases_zero_padded(ref):
    S = gen.iid(5000, 5)
    idx = np.concatenate([np.arange(0, 64), np.arange(S.size - 40, S.size + 1)]).astype(np.uint32)
    got = ref.prefix10(S, idx)
    for p, g in zip(idx.tolist(), got.tolist()):
        v = 0
        for b in key_of(S, p, 10).tolist():
            v = (v << 2) | b
        assert g == v, p


def test_load125_orders_like_the_first_125_bases(ref):
    # the 256-bit word of load_prefix_length_125 compared most-significant-byte-first (what the comparator does with it,
    # kiss1_core.hpp:103-113) must order positions exactly like their first 125 bases
    S = gen.periodic(6000, 37, 3, mutations=40)
    idx = np.random.default_rng(1).integers(0, S.size - 125, 400).astype(np.uint32)
    words = ref.load125(S, idx)
    keys_ref = [bytes(w[::-1]) for w in words]  # byte 31 first
    keys_txt = [S[p:p + 125].tobytes() for p in idx.tolist()]
    for a in range(0, 400, 7):
        for b in range(0, 400, 11):
            assert (keys_ref[a] < keys_ref[b]) == (keys_txt[a] < keys_txt[b])
            assert (keys_ref[a] == keys_ref[b]) == (keys_txt[a] == keys_txt[b])


@pytest.mark.parametrize("k", KS)
@pytest.mark.parametrize("name,S", SHAPES, ids=[s[0] for s in SHAPES])
def test_induction_by_reference_code_only(oracle, ref, name, S, k):
    """LMS order from the oracle, then ONLY reference code: get_lms + put_lms_suffix + induced_sort (multi-threaded
    block scheduling included) -> must reproduce the oracle's SA bit for bit."""
    sa_o, lms_sorted_o = oracle.suffix_sort(S, k, stages=True)
    for T in (1, 4):
        sa_r = ref.suffix_sort(S, k, T=T, sorted_lms=lms_sorted_o)
        assert np.array_equal(sa_o, sa_r), (name, k, T)


@pytest.mark.parametrize("k", KS)
@pytest.mark.parametrize("name,S", SHAPES, ids=[s[0] for s in SHAPES])
def test_full_pipeline_two_restatements_of_the_comparator_agree(oracle, ref, name, S, k):
    sa_o, lms_o = oracle.suffix_sort(S, k, stages=True)
    sa_r, lms_r = ref.suffix_sort(S, k, stages=True)
    assert np.array_equal(lms_o, lms_r), (name, k)
    assert np.array_equal(sa_o, sa_r), (name, k)


def test_committed_oracle_goldens_equal_reference(ref):
    """tests/golden/*.npz were written by the oracle (make_golden.py); the reference's own get_lms / placement /
    induction reproduce every one of them"""
    import os
    gdir = os.path.join(os.path.dirname(__file__), "golden")
    files = sorted(f for f in os.listdir(gdir) if f.endswith(".npz"))
    assert files
    for f in files:
        z = np.load(os.path.join(gdir, f))
        sa = ref.suffix_sort(z["S"], int(z["k"]))
        assert np.array_equal(sa, z["SA"]), f
