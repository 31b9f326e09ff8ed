"""CPU tests of the oracle itself (no GPU): the reference's own test property (k-order,
tests/kiss.cpp:26-28), equality with a naive suffix sorter for k >= n, brute-force FM hit sets,
the .fmi size formula, and the committed golden fixtures."""
import os

import numpy as np
import pytest

from tests import gen

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def naive_sa(S):
    b = S.tobytes()
    n = len(b)
    return np.array([n] + sorted(range(n), key=lambda i: b[i:]), dtype=np.uint32)


def k_order_ok(S, SA, k):
    """the reference's REQUIRE: S.substr(sa[i-1], k) <= S.substr(sa[i], k) for i >= 1 (bytes compare like chars)"""
    b = S.tobytes()
    for i in range(1, len(SA)):
        if not b[SA[i - 1]:SA[i - 1] + k] <= b[SA[i]:SA[i] + k]:
            return False
    return True


@pytest.mark.parametrize("n", [0, 1, 2, 3, 5, 16, 50, 200, 1000, 3000])
def test_exact_order_matches_naive(oracle, n):
    rng = np.random.default_rng(n)
    cases = [rng.integers(0, 4, n, dtype=np.uint8), np.zeros(n, np.uint8),
             np.tile(np.array([0, 1], np.uint8), n)[:n].copy(), gen.periodic(n, 7, 3) if n else np.zeros(0, np.uint8)]
    for S in cases:
        assert np.array_equal(oracle.suffix_sort(S, 0xFFFFFFFF), naive_sa(S))


@pytest.mark.parametrize("period", [1, 2, 3, 5, 37, 400])
@pytest.mark.parametrize("k", [32, 256])
def test_k_order_property_with_ties(oracle, period, k):
    S = gen.periodic(4000, period, 11 + period, mutations=5)
    SA = oracle.suffix_sort(S, k)
    assert SA[0] == S.size
    assert np.array_equal(np.sort(SA), np.arange(S.size + 1, dtype=np.uint32))
    assert k_order_ok(S, SA, k)


def test_reference_test_shape(oracle):
    # tests/kiss.cpp "kISS-1 DNA": random 100k..200k bases, k = 256, k-order property
    S = gen.iid(150_000, 42)
    SA = oracle.suffix_sort(S, 256)
    assert np.array_equal(np.sort(SA), np.arange(S.size + 1, dtype=np.uint32))
    b = S.tobytes()
    idx = np.random.default_rng(0).integers(1, SA.size, 20000)
    for i in idx:
        assert b[SA[i - 1]:SA[i - 1] + 256] <= b[SA[i]:SA[i] + 256]


def test_get_lms_definition(oracle):
    S = gen.iid(5000, 9)
    lms, hist = oracle.get_lms(S)
    n = S.size
    typ = np.zeros(n + 1, dtype=bool)  # True = S-type
    typ[n] = True
    for i in range(n - 2, -1, -1):
        typ[i] = S[i] < S[i + 1] or (S[i] == S[i + 1] and typ[i + 1])
    typ[n - 1] = False
    want = [i for i in range(1, n) if typ[i] and not typ[i - 1]] + [n]
    assert lms.tolist() == want
    assert hist[4, :4].tolist() == np.bincount(S, minlength=4).tolist()


def brute_hits(S, pat):
    b, p = S.tobytes(), pat.tobytes()
    out, i = [], b.find(p)
    while i >= 0:
        out.append(i)
        i = b.find(p, i + 1)
    return out


def test_fm_index_against_brute_force(oracle):
    # i.i.d. text with short planted copies: every tie is resolved well inside the k=32 build's depth of
    # 125 bases, so the k-ordered SA is the true SA and the index must agree with brute force
    # (long periodic repeats make the reference's k-ordered index inexact by design)
    S = gen.iid(60_000, 5)
    rng0 = np.random.default_rng(2)
    for _ in range(200):
        a, b = rng0.integers(0, S.size - 80, 2)
        S[b:b + 60] = S[a:a + 60]
    SA = oracle.suffix_sort(S, 32)
    fmi = oracle.fm_build(S, SA)
    assert len(fmi.serialize()) == fmi_size(S.size)
    rng = np.random.default_rng(1)
    L = 32
    pos = rng.integers(4, S.size - L, 300)
    pats = np.stack([S[p:p + L] for p in pos])
    pats[::10, 5] = (pats[::10, 5] + 1) % 4
    res = fmi.query_batch(pats)
    for q in range(pats.shape[0]):
        want = brute_hits(S, pats[q])
        a, b = res["offsets_index"][q], res["offsets_index"][q + 1]
        assert sorted(res["offsets"][a:b].tolist()) == want
        assert res["end"][q] - res["beg"][q] == len(want)


def fmi_size(n):
    N = n + 1
    return (20 + 8 + (N + 3) // 4 + 8 + (N // 256 + 1) * 16 + 8 + (N // 16 + 1) * 4 + 8 + ((N + 3) // 4) * 4 + 8 + 8
            + 8 + ((N + 63) // 64) * 8 + 8 + (N // 64 + 1) * 4)


def test_fmi_layout_size_formula(oracle):
    # n = 1000 -> 1847 bytes
    assert fmi_size(1000) == 1847
    S = gen.iid(1000, 7)
    fmi = oracle.fm_build(S, oracle.suffix_sort(S, 32))
    assert len(fmi.serialize()) == 1847


def test_golden_fixtures(oracle):
    files = sorted(f for f in os.listdir(GOLDEN) if f.endswith(".npz"))
    assert files, "tests/golden/*.npz missing (run tests/golden/make_golden.py)"
    for f in files:
        z = np.load(os.path.join(GOLDEN, f))
        S, k = z["S"], int(z["k"])
        SA = oracle.suffix_sort(S, k)
        if "SA" in z:
            assert np.array_equal(SA, z["SA"]), f
        assert oracle.fnv(SA) == int(z["sa_fnv"]), f


def test_reader_known_answers(oracle):
    # hand-derived from the reference's control flow (utils/io.hpp:6-18, file_io/fasta.hpp:117-151):
    # header dropped, non-ACGT -> 4 % 4 = 0, a '>' line right after a header is sequence, the next one a header
    A, C, G, T = 0, 1, 2, 3
    got = oracle.read_sequence(b">h1 x\nACGT\nNNac\n>h2\n>zzA\n>h3\nTT")
    assert got.tolist() == [A, C, G, T, A, A, A, C, A, A, A, A, T, T]
    # text mode: every byte but '\n' is a base, '\r' included; '>' lines are not special
    assert oracle.read_sequence(b"ACGT\r\n>x\nGG").tolist() == [A, C, G, T, A, A, A, G, G]
    assert oracle.read_sequence(b"").size == 0
    assert oracle.read_sequence(b">only a header").size == 0
    assert oracle.read_sequence(b">h\n\n\nAC\n\nGT\n").tolist() == [A, C, G, T]
    assert oracle.read_sequence(b"\nAC").tolist() == [A, C]          # starts with '\n': text mode
    assert oracle.read_sequence(b">h\r\nAC\r\n").tolist() == [A, C, A]  # CRLF: the '\r' of a sequence line is a base
