"""GPU parity tests: libkiss_hip.so (through its C ABI) against the CPU oracle, bit-exact."""
import numpy as np
import pytest

from tests import gen

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import kiss_amd
    c = kiss_amd.Context(max_n=6_000_000, device=0)
    yield c
    c.close()


def check_parity(ctx, oracle, S, k, stages=True):
    S = np.ascontiguousarray(S, dtype=np.uint8)
    n = S.size
    sa_gpu = ctx.suffix_sort(S, k)
    sa_ref, lms_sorted_ref = oracle.suffix_sort(S, k, stages=True)
    if stages and n > 0:
        asc, srt, counts = ctx.stage_outputs()
        lms_ref, hist = oracle.get_lms(S)
        assert asc.size == lms_ref.size - 1, "LMS count"
        assert np.array_equal(asc, lms_ref[:-1]), "ascending LMS list (get_lms)"
        assert np.array_equal(counts[0:4], hist[4, :4].astype(np.uint64)), "count[c]"
        assert np.array_equal(counts[8:12], hist[2, :4].astype(np.uint64)), "lms_count[c]"
        # S-type count = pair types 01 (i S, i-1... see oracle) : S-type positions are those with bit1 of the pair type
        ref_sorted = lms_sorted_ref[1:]  # drop the sentinel
        if not np.array_equal(srt, ref_sorted):
            bad = np.nonzero(srt != ref_sorted)[0]
            raise AssertionError("k-ordered LMS list differs at %d of %d entries, first at %d: gpu %d ref %d"
                                 % (bad.size, srt.size, bad[0], srt[bad[0]], ref_sorted[bad[0]]))
    if not np.array_equal(sa_gpu, sa_ref):
        bad = np.nonzero(sa_gpu != sa_ref)[0]
        raise AssertionError("SA differs at %d of %d entries, first at %d: gpu %d ref %d"
                             % (bad.size, sa_gpu.size, bad[0], sa_gpu[bad[0]], sa_ref[bad[0]]))
    return sa_gpu


@pytest.mark.parametrize("n", [0, 1, 2, 3, 5, 16, 31, 32, 33, 50, 64, 65, 200, 1000, 4097, 100_003])
@pytest.mark.parametrize("k", [32, 256, 0xFFFFFFFF])
def test_iid(ctx, oracle, n, k):
    check_parity(ctx, oracle, gen.iid(n, 1000 + n), k)


@pytest.mark.parametrize("period", [1, 2, 3, 5, 7, 37, 400])
@pytest.mark.parametrize("k", [32, 256, 0xFFFFFFFF])
def test_repeats(ctx, oracle, period, k):
    n = 20_000
    check_parity(ctx, oracle, gen.periodic(n, period, 7 + period, mutations=6), k)


@pytest.mark.parametrize("k", [32, 256, 0xFFFFFFFF])
def test_all_same_and_alternating(ctx, oracle, k):
    for c in range(4):
        check_parity(ctx, oracle, np.full(3000, c, dtype=np.uint8), k)
    check_parity(ctx, oracle, np.tile(np.array([0, 1], np.uint8), 1500), k)
    check_parity(ctx, oracle, np.tile(np.array([3, 0], np.uint8), 1500), k)


@pytest.mark.parametrize("k", [32, 256])
def test_text_ends_inside_repeat(ctx, oracle, k):
    # exercises the near-end rule (reference kiss1_core.hpp:120-134): the tail of the text is a copy of
    # an earlier region, cut at different distances from the end
    rng = np.random.default_rng(5)
    base = rng.integers(0, 4, 30_000, dtype=np.uint8)
    for cut in [10, 100, 124, 125, 126, 255, 256, 257, 300, 374, 375, 376, 500]:
        S = np.concatenate([base, base[5000:5000 + cut]])
        check_parity(ctx, oracle, S, k)


@pytest.mark.parametrize("k", [32, 256, 0xFFFFFFFF])
def test_genome_like(ctx, oracle, k):
    check_parity(ctx, oracle, gen.genome_like(2_000_000, 11), k)


def test_long_homopolymer_runs(ctx, oracle):
    rng = np.random.default_rng(9)
    S = rng.integers(0, 4, 300_000, dtype=np.uint8)
    for i in range(40):
        p = int(rng.integers(0, S.size - 6000))
        S[p:p + 5000] = i % 4
    check_parity(ctx, oracle, S, 256)


def test_one_shot_entry(oracle):
    import kiss_amd
    S = gen.iid(50_000, 3)
    sa = kiss_amd.KISS1Sorter.get_suffix_array_dna(S, 256)
    assert np.array_equal(sa, oracle.suffix_sort(S, 256))
    sa2 = kiss_amd.KISS2Sorter.get_suffix_array_dna(S, kiss_amd.K_UNBOUNDED)
    assert np.array_equal(sa2, oracle.suffix_sort(S, 0xFFFFFFFF))


@pytest.mark.parametrize("k", [256, 0xFFFFFFFF])
def test_megabase_runs_of_one_base(ctx, oracle, k):
    # runs longer than the 16-bit and the 22-bit chain-collapse steps (induce.hip run_collapse: narrow, then
    # wide step fields; run lengths counted by a whole wave past the first 1024 bases)
    n = 5_500_000
    S = gen.iid(n, 77)
    rng = np.random.default_rng(78)
    S[300_000:300_000 + 4_400_000] = 3            # > 2^22
    for i, ln in enumerate([70_000, 131_073, 300_000, 65_535, 65_536, 66_000 + 1024, 2048 + 1024, 1024, 1023]):
        p = 4_750_000 + i * 40_000 if ln < 40_000 else None
        if p is None:
            p = int(rng.integers(300_000, 4_000_000))
        S[p:p + ln] = i % 4
    check_parity(ctx, oracle, S, k, stages=False)
    for c in (0, 3):
        check_parity(ctx, oracle, np.full(n, c, np.uint8), k, stages=False)
