"""GPU parity tests of the FM-index path: build (.fmi bytes), backward search and locate, against the oracle."""
import numpy as np
import pytest

from tests import gen
from tests.fmi_layout import canonical

pytestmark = pytest.mark.gpu


def make_patterns(S, Q, L, seed):
    rng = np.random.default_rng(seed)
    pos = rng.integers(0, S.size - L, Q)
    pats = np.stack([S[p:p + L] for p in pos])
    mut = rng.random(Q) < 0.1
    col = rng.integers(0, L, Q)
    pats[mut, col[mut]] = (pats[mut, col[mut]] + 1 + rng.integers(0, 3, int(mut.sum()))) % 4
    return pats.astype(np.uint8)


@pytest.mark.parametrize("n,seed", [(1000, 7), (4096 - 1, 3), (100_000, 7), (1_000_000, 9)])
def test_build_matches_oracle_bytes(oracle, n, seed):
    import kiss_amd.fm_index as fm
    S = gen.genome_like(n, seed) if n >= 100_000 else gen.iid(n, seed)
    f = fm.FMIndex().build(S)
    ref = oracle.fm_build(S, oracle.suffix_sort(S, 32))
    assert canonical(f.to_bytes()) == canonical(ref.serialize())
    assert f.to_bytes() == canonical(f.to_bytes())  # our builder writes zeros into the undefined padding bits
    # and a round trip through the byte layout
    g = fm.FMIndex.from_bytes(f.to_bytes())
    assert canonical(g.to_bytes()) == canonical(ref.serialize())
    f.close()


@pytest.mark.parametrize("L", [1, 5, 20, 32])
def test_query_batch_matches_oracle(oracle, L):
    import kiss_amd.fm_index as fm
    S = gen.genome_like(400_000, 21)
    f = fm.FMIndex().build(S)
    ref = oracle.fm_build(S, oracle.suffix_sort(S, 32))
    pats = make_patterns(S, 20_000, L, 5)
    # patterns that hit the very start of the text exercise the primary-row corner of get_offsets
    pats[:8] = np.stack([S[i:i + L] for i in range(8)])
    a = f.query_batch(pats)
    b = ref.query_batch(pats)
    assert np.array_equal(a["beg"], b["beg"]) and np.array_equal(a["end"], b["end"])
    assert a["total_hits"] == b["total_hits"] and a["checksum"] == b["checksum"]
    assert np.array_equal(a["offsets_index"], b["offsets_index"])
    assert np.array_equal(a["offsets"], b["offsets"])
    f.close()


def test_config3_shape(oracle):
    # BASELINE.json configs[2] in miniature: index + batch of 32-base patterns, 90 % sampled / 10 % one substitution
    import kiss_amd.fm_index as fm
    S = gen.genome_like(3_000_000, 1)
    f = fm.FMIndex().build(S)
    ref = oracle.fm_build(S, oracle.suffix_sort(S, 32))
    pats = make_patterns(S, 200_000, 32, 3)
    a = f.query_batch(pats, want_offsets=False)
    b = ref.query_batch(pats, want_offsets=False)
    assert a["total_hits"] == b["total_hits"] and a["checksum"] == b["checksum"]
    assert np.array_equal(a["beg"], b["beg"]) and np.array_equal(a["end"], b["end"])
    f.close()


@pytest.mark.parametrize("L", [2, 7, 12, 40])
def test_heavy_patterns_keep_the_fifo_order(oracle, L):
    # patterns with thousands of occurrences (tandem repeats, homopolymers) are located by a workgroup
    # (fm.hip k_fm_locate_heavy): offsets must come out in the reference's breadth-first order, and the text
    # starts inside a repeat so that the primary row sits in a heavy range
    import kiss_amd.fm_index as fm
    rng = np.random.default_rng(77)
    S = gen.iid(300_000, 5)
    S[:40_000] = np.tile(np.array([0, 1, 2], np.uint8), 13_334)[:40_000]
    S[100_000:160_000] = 3
    S[200_000:230_000] = np.tile(rng.integers(0, 4, 11, dtype=np.uint8), 2728)[:30_000]
    f = fm.FMIndex().build(S)
    ref = oracle.fm_build(S, oracle.suffix_sort(S, 32))
    pos = np.concatenate([rng.integers(0, 40_000 - L, 300), rng.integers(100_000, 160_000 - L, 300),
                          rng.integers(200_000, 230_000 - L, 300), rng.integers(0, S.size - L, 300), np.arange(8)])
    pats = np.stack([S[p:p + L] for p in pos]).astype(np.uint8)
    a = f.query_batch(pats)
    b = ref.query_batch(pats)
    assert np.array_equal(a["beg"], b["beg"]) and np.array_equal(a["end"], b["end"])
    assert int((b["end"].astype(np.int64) - b["beg"]).max()) > 5000
    assert a["total_hits"] == b["total_hits"] and a["checksum"] == b["checksum"]
    assert np.array_equal(a["offsets_index"], b["offsets_index"])
    assert np.array_equal(a["offsets"], b["offsets"])
    f.close()


@pytest.mark.parametrize("tail", [3000, 700])
def test_locate_returns_more_than_the_range_on_tied_repeats(oracle, tail):
    """get_offsets (fm_index.hpp:472-482) checks `offsets.size() < end - beg` only before it takes a range from the queue:
    on the k = 32 index of a text with a long tandem array (ties past the comparison depth) the walk returns MORE
    positions than end - beg.  The library must return the same list (found by tools/fuzz_fm.py: a text that is mostly
    (TTAGGG)n; the old per-pattern capacity of end - beg + 4 cut one position off)."""
    import kiss_amd.fm_index as fm
    rng = np.random.default_rng(3)
    unit = np.array([3, 3, 0, 2, 2, 2], dtype=np.uint8)
    S = np.concatenate([gen.iid(500, 1), np.tile(unit, tail // 6), gen.iid(300, 2), np.tile(unit, tail // 6),
                        gen.iid(40, 4)]).astype(np.uint8)
    f = fm.FMIndex().build(S)
    ref = oracle.fm_build(S, oracle.suffix_sort(S, 32))
    pats = []
    for L in (6, 20, 38):
        for sh in range(6):
            pats.append(np.tile(unit, 10)[sh:sh + L])
        a = f.query_batch(np.stack(pats[-6:]), want_offsets=True)
        b = ref.query_batch(np.stack(pats[-6:]), want_offsets=True)
        assert a["total_hits"] == b["total_hits"] and a["checksum"] == b["checksum"]
        assert np.array_equal(a["beg"], b["beg"]) and np.array_equal(a["end"], b["end"])
        assert np.array_equal(a["offsets_index"], b["offsets_index"]) and np.array_equal(a["offsets"], b["offsets"])
    del rng
    f.close()


def test_index_file_with_undefined_padding_bits_loads_and_answers_identically(oracle):
    # a .fmi written by the reference carries junk in the padding bits of the last bwt byte / b_ word
    # (xbit_vector.hpp:1290-1309); the drop-in must read such a file and answer exactly as from its own
    import kiss_amd.fm_index as fm
    from tests.fmi_layout import with_garbage_padding
    for n in (1001, 99_999, 100_002):  # N % 4 and N % 64 non-zero
        S = gen.genome_like(n, 5) if n > 50_000 else gen.iid(n, 5)
        f = fm.FMIndex().build(S)
        junk = with_garbage_padding(f.to_bytes())
        assert junk != f.to_bytes() and canonical(junk) == f.to_bytes()
        g = fm.FMIndex.from_bytes(junk)
        pats = make_patterns(S, 3000, 12, 9)
        pats[:4] = np.stack([S[i:i + 12] for i in (0, 1, n - 12, n - 13)])
        a, b = f.query_batch(pats), g.query_batch(pats)
        for key in ("beg", "end", "offsets", "offsets_index"):
            assert np.array_equal(a[key], b[key]), key
        assert a["checksum"] == b["checksum"]
        f.close()
        g.close()
