"""General alphabet (bytes): the exact suffix array of kiss_hip_suffix_sort_u8 against plain Python / the DNA path."""
import numpy as np
import pytest

from tests import gen

pytestmark = pytest.mark.gpu


def naive_sa(b):
    n = len(b)
    return np.array([n] + sorted(range(n), key=lambda i: b[i:]), dtype=np.uint32)


def is_suffix_array(S, sa):
    """linear-time check: a permutation with SA[0] = n, and (S[a], rank[a + 1]) strictly increasing along SA"""
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


@pytest.mark.parametrize("text", [b"", b"a", b"aa", b"ab", b"ba", b"banana", b"mississippi", b"abracadabra" * 7,
                                  b"\x00\x00\x00", b"\x00\x01\x00\x01\x00", b"\xff" * 40, bytes(range(256)) * 3,
                                  b"aaaaaaab" * 50 + b"aaaaaaa", b"the quick brown fox jumps over the lazy dog " * 20])
def test_small_texts_against_python(text):
    import kiss_amd
    sa = kiss_amd.suffix_array_bytes(text)
    assert np.array_equal(sa, naive_sa(text))


def test_random_texts_against_python():
    import kiss_amd
    rng = np.random.default_rng(5)
    for case in range(60):
        n = int(rng.integers(0, 1500))
        sigma = int(rng.choice([1, 2, 3, 4, 16, 256]))
        b = rng.integers(0, sigma, n, dtype=np.uint8)
        if case % 3 == 0 and n > 50:  # plant repeats longer than the 7-character key
            a, c, ln = int(rng.integers(0, n // 2)), int(rng.integers(n // 2, n)), int(rng.integers(8, 40))
            b[c:c + ln] = b[a:a + ln][:b[c:c + ln].size]
        assert np.array_equal(kiss_amd.suffix_array_bytes(b.tobytes()), naive_sa(b.tobytes())), (case, n, sigma)


@pytest.mark.parametrize("kind", ["bytes", "english-like", "periodic", "dna"])
def test_large_texts_are_suffix_arrays(oracle, kind):
    import kiss_amd
    rng = np.random.default_rng(9)
    n = 2_000_000
    if kind == "bytes":
        S = rng.integers(0, 256, n, dtype=np.uint8)
    elif kind == "english-like":
        words = [bytes(rng.integers(97, 123, int(rng.integers(2, 9)), dtype=np.uint8)) + b" " for _ in range(300)]
        S = np.frombuffer(b"".join(words[int(i)] for i in rng.integers(0, 300, n // 5)), dtype=np.uint8)[:n].copy()
    elif kind == "periodic":
        S = np.tile(rng.integers(0, 256, 13, dtype=np.uint8), n // 13 + 1)[:n].copy()
    else:
        S = gen.genome_like(n, 3)
    sa = kiss_amd.suffix_array_bytes(S)
    assert is_suffix_array(S, sa)
    if kind == "dna":  # the byte path and the DNA paths agree (codes 0..3 are bytes too)
        assert np.array_equal(sa, oracle.suffix_sort(S, kiss_amd.K_UNBOUNDED))


@pytest.mark.parametrize("values", [b"ABCD", b"ACT", b"\x00\xff", b"z", b"\x05\x06\x07\x08"])
def test_texts_over_at_most_four_values_take_the_dna_path(oracle, monkeypatch, values):
    # general.hip maps such a text to codes 0..3 in value order and sorts it as DNA (exact order); the answer is the
    # one the 7-character path gives (KISS_HIP_NO_SMALL_ALPHABET=1) and the oracle's exact suffix array of the codes
    import kiss_amd
    codes = gen.genome_like(300_000, 3) % len(values) if len(values) > 1 else np.zeros(20_000, np.uint8)
    if codes.size > 60_000:
        codes[1000:1600] = codes[50_000:50_600]  # a copy longer than any key
    text = np.frombuffer(values, np.uint8)[codes]
    sa = kiss_amd.suffix_array_bytes(text.tobytes())
    assert is_suffix_array(text, sa)
    assert np.array_equal(sa, oracle.suffix_sort(codes.astype(np.uint8), 0xFFFFFFFF))
    monkeypatch.setenv("KISS_HIP_NO_SMALL_ALPHABET", "1")  # (a switch of the hooks build)
    assert np.array_equal(kiss_amd.suffix_array_bytes(text.tobytes(), hooks=True), sa)
