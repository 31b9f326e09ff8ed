"""GPU parity tests: libkiss_hip.so (through its C ABI) against the CPU oracle, bit-exact."""
import numpy as np
import pytest

from tests import gen

pytestmark = pytest.mark.gpu
EXACT_H0 = 512  # order of the bounded phase in front of the rank doubling (KISS_EXACT_H0, kiss_internal.hpp)


@pytest.fixture(scope="module")
def ctx():
    import kiss_amd
    c = kiss_amd.Context(max_n=6_000_000, device=0)
    yield c
    c.close()


@pytest.fixture(scope="module")
def hctx():
    # a context of the hooks build (libkiss_hip_hooks.so): the only one whose behaviour KISS_HIP_* environment switches
    # change -- the tests that force the rare paths and compare the A-B forms use it; everything else runs on the shipped library
    import kiss_amd
    c = kiss_amd.Context(max_n=6_000_000, device=0, hooks=True)
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


def _exact_by_doubling(ctx, oracle, S, h0=None, monkeypatch=None):
    import kiss_amd
    S = np.ascontiguousarray(S, dtype=np.uint8)
    sa = ctx.suffix_sort(S, kiss_amd.K_UNBOUNDED, algo=1)
    ref = oracle.suffix_sort(S, kiss_amd.K_UNBOUNDED)
    if not np.array_equal(sa, ref):
        bad = np.nonzero(sa != ref)[0]
        raise AssertionError("doubling SA differs at %d of %d entries, first at %d: gpu %d ref %d (stats %s)"
                             % (bad.size, sa.size, bad[0], sa[bad[0]], ref[bad[0]],
                                {k: v for k, v in ctx.stats().items() if k != "kernels"}))
    return ctx.stats()


@pytest.mark.parametrize("shape", ["iid", "genome", "period1", "period2", "period7", "period400", "period5000",
                                   "runs", "nested", "tail_repeat"])
def test_prefix_doubling_exact(ctx, oracle, shape):
    # KISS2 / PREFIX_DOUBLING path (reference kiss2_core.hpp:835-886): bounded phase + rank doubling over the
    # full suffix array; the result is the unique exact suffix array, also on texts whose repeats are as long as
    # the text (where comparing 32 bases per round would need n/32 rounds)
    n = 300_000
    rng = np.random.default_rng(11)
    if shape == "iid":
        S = gen.iid(n, 3)
    elif shape == "genome":
        S = gen.genome_like(n, 4)
    elif shape.startswith("period"):
        p = int(shape[6:])
        S = np.tile(rng.integers(0, 4, p, dtype=np.uint8), n // p + 1)[:n]
        if p == 400:
            S = S.copy()
            S[rng.integers(0, n, 5)] = 1  # a few mutations
    elif shape == "runs":
        S = gen.iid(n, 5)
        for i in range(40):
            a = int(rng.integers(0, n - 9000))
            S[a:a + int(rng.integers(300, 9000))] = i % 4
    elif shape == "nested":
        u = rng.integers(0, 4, 700, dtype=np.uint8)
        S = np.concatenate([u, u, u[:350], gen.iid(5000, 6), u, u, u, u, gen.iid(n - 5000 - 700 * 6 - 350, 7)])
    else:  # the text ends inside a long copy of its own beginning
        base = gen.iid(n // 2, 8)
        S = np.concatenate([base, gen.iid(100, 9), base[:n // 2 - 100 - 37]])
    st = _exact_by_doubling(ctx, oracle, S)
    assert st["refine_depth"] == EXACT_H0
    # (period1, a text of one base, has no LMS suffix at all: the induction yields the exact order by itself, nothing is
    #  tainted and the doubling phase finds nothing to do)
    if shape in ("period2", "period7", "period400", "period5000", "nested", "tail_repeat"):
        assert st["doubling_rounds"] >= 1 and st["refine_items"] > 0
        assert st["doubling_rounds"] <= 14  # log2(n / 256) + slack, not n / 32


def test_prefix_doubling_small_and_depths(hctx, oracle, monkeypatch):
    import kiss_amd
    for n in [0, 1, 2, 31, 300, 1024, 2047, 2048, 2049, 5000]:
        S = gen.periodic(n, 3, 1, mutations=2) if n > 10 else gen.iid(n, n)
        _exact_by_doubling(hctx, oracle, S)
    S = gen.periodic(60_000, 37, 2, mutations=3)
    for h0 in ("32", "33", "64", "124", "125", "1000"):
        monkeypatch.setenv("KISS_HIP_DOUBLING_H0", h0)
        st = _exact_by_doubling(hctx, oracle, S)
        assert st["refine_depth"] == int(h0)


def test_prefix_doubling_binned_inverse(oracle, monkeypatch):
    # the inverse suffix array of the doubling phase is built by a two-level partition (isa.hip) once it no longer
    # fits the last-level cache; force that path at a size the oracle still sorts in seconds:
    # two level-1 bins, the second one short (a few full level-2 bins + a partial one)
    import kiss_amd
    monkeypatch.setenv("KISS_HIP_ISA_DIRECT_MAX", "1000")
    n = (1 << 24) + 5 * (1 << 16) + 12345
    S = gen.genome_like(n, 21)
    S[1000:9000] = 2                      # sorted stretches of SA (one-bin waves in the partition)
    S[2_000_000:2_300_000] = S[9_000_000:9_300_000]
    c = kiss_amd.Context(max_n=n, device=0, hooks=True)
    try:
        st = _exact_by_doubling(c, oracle, S)
        assert st["refine_items"] > 300_000
    finally:
        c.close()


def test_exact_msd_falls_back_to_doubling(ctx, oracle):
    # PARALLEL_SORTING with k >= n compares 32 bases per round; past 32768 tied bases the library restarts
    # through the doubling path (api.hip sort_dev) -- same, unique result
    import kiss_amd
    n = 200_000
    S = np.tile(np.array([0, 2, 1, 3, 3, 0, 1], np.uint8), n // 7 + 1)[:n]
    sa = ctx.suffix_sort(S, kiss_amd.K_UNBOUNDED, algo=0)
    st = ctx.stats()
    assert st["refine_depth"] == EXACT_H0 and st["doubling_rounds"] >= 1
    assert np.array_equal(sa, oracle.suffix_sort(S, kiss_amd.K_UNBOUNDED))
    # a repeat shorter than the switch-over depth stays on the 32-bases-per-round path
    S2 = gen.iid(n, 31)
    S2[100_000:120_000] = S2[10_000:30_000]
    sa2 = ctx.suffix_sort(S2, kiss_amd.K_UNBOUNDED, algo=0)
    assert ctx.stats()["refine_depth"] == 0
    assert np.array_equal(sa2, oracle.suffix_sort(S2, kiss_amd.K_UNBOUNDED))


def test_error_codes(ctx):
    # error behaviour of the ABI: negative status codes, never a crash, never a CPU fallback
    import ctypes
    import kiss_amd
    from kiss_amd import _lib
    lib = _lib.load()
    S = gen.iid(1000, 1)
    with pytest.raises(kiss_amd.KissHipError) as e:      # text longer than the workspace was created for
        ctx.suffix_sort(np.zeros(ctx.max_n + 1, np.uint8), 256)
    assert e.value.status == _lib.KISS_HIP_E_INVALID
    with pytest.raises(kiss_amd.KissHipError) as e:      # unknown algorithm
        ctx.suffix_sort(S, 256, algo=7)
    assert e.value.status == _lib.KISS_HIP_E_INVALID
    # KISS2 with a bounded k (the CLI's `-s PREFIX_DOUBLING` with the default k): accepted, the k-ordered SA of KISS1
    assert np.array_equal(ctx.suffix_sort(S, 256, algo=1), ctx.suffix_sort(S, 256, algo=0))
    SA = np.empty(4, np.uint32)
    assert lib.kiss_hip_suffix_sort_dna_u32(None, 3, 256, 0, SA.ctypes.data, 0) == _lib.KISS_HIP_E_INVALID
    assert lib.kiss_hip_suffix_sort_dna_u32(S.ctypes.data, 3, 256, 0, None, 0) == _lib.KISS_HIP_E_INVALID
    assert lib.kiss_hip_suffix_sort_dna_u32(S.ctypes.data, 3, 256, 0, SA.ctypes.data, 99) == -2   # no such device
    c = ctypes.c_void_p()
    assert lib.kiss_hip_ctx_create(ctypes.byref(c), 0, (1 << 32)) == _lib.KISS_HIP_E_INVALID      # n must fit 32 bits
    # and the context is still usable after the errors
    sa = ctx.suffix_sort(S, 256)
    assert sa[0] == S.size and sorted(sa.tolist()) == list(range(S.size + 1))


def test_context_recovers_after_running_out_of_memory(oracle, monkeypatch):
    # a text with more LMS suffixes than the DNA-typical reservation makes the work arrays grow; when that growth does not
    # fit (fault injection: arrays over 700 kB "do not fit") the call reports KISS_HIP_E_NOMEM, and the same context sorts the
    # next text as if nothing had happened (found by tools/stress_verify.py at n = 2.4e9: it did not)
    import kiss_amd
    from kiss_amd import _lib
    n = 200_000   # default reservation: 0.32 n LMS suffixes (545 kB of keys); ACAC.. has n/2 (821 kB)
    c = kiss_amd.Context(max_n=n, device=0, hooks=True)
    try:
        dna = gen.iid(n, 77)
        acac = np.tile(np.array([0, 1], np.uint8), n // 2)              # n/2 LMS suffixes, all of them tied
        want_dna = oracle.suffix_sort(dna, 256)
        assert np.array_equal(c.suffix_sort(dna, 256), want_dna)
        for k in (256, 0xFFFFFFFF):
            if k != 256:
                # exact order through the LMS-level doubling needs no array beyond what the k = 256 sort of this text has
                # grown already; the suffix-array form (all n suffixes tied) does: that is the growth that fails here
                monkeypatch.setenv("KISS_HIP_NO_LMS_EXACT", "1")
            assert _lib.load(True).kiss_hip_debug_fail_alloc_over(c._ctx, 700000) == 0
            with pytest.raises(kiss_amd.KissHipError) as e:
                c.suffix_sort(acac, k)
            assert e.value.status == _lib.KISS_HIP_E_NOMEM
            with pytest.raises(kiss_amd.KissHipError) as e:             # still failing while memory is short, still cleanly
                c.suffix_sort(acac, k)
            assert e.value.status == _lib.KISS_HIP_E_NOMEM
            assert _lib.load(True).kiss_hip_debug_fail_alloc_over(c._ctx, 0) == 0
            assert np.array_equal(c.suffix_sort(dna, 256), want_dna)
            assert np.array_equal(c.suffix_sort(acac, k), oracle.suffix_sort(acac, k))
            monkeypatch.delenv("KISS_HIP_NO_LMS_EXACT", raising=False)
            assert np.array_equal(c.suffix_sort(acac, k), oracle.suffix_sort(acac, k))
    finally:
        c.close()


@pytest.mark.parametrize("hook", ["KISS_HIP_NO_FC0_ONEPASS", "KISS_HIP_MERGE_LMS"])
def test_round2_shortcuts_equal_the_forms_they_replace(hctx, oracle, monkeypatch, hook):
    # KISS_HIP_NO_FC0_ONEPASS: round 0 flags + compacts with count / scan / compact instead of the one-pass look-back;
    # KISS_HIP_MERGE_LMS: the induction reads a merged copy of the LMS list instead of far list + near-end table.
    # Both forms must give the oracle's SA; with the tied-segment arrays too small for the first attempt as well
    # (the one-pass form learns the survivor count only afterwards and runs again).
    S = gen.genome_like(3_000_000, 31)
    want = oracle.suffix_sort(S, 256)
    assert np.array_equal(hctx.suffix_sort(S, 256), want)
    monkeypatch.setenv(hook, "1")
    assert np.array_equal(hctx.suffix_sort(S, 256), want)
    monkeypatch.delenv(hook)
    import kiss_amd
    rep = np.tile(gen.iid(3000, 5), 700)          # every LMS suffix tied after round 0: more than the default t_cap
    c = kiss_amd.Context(max_n=rep.size, device=0, hooks=True)
    try:
        assert np.array_equal(c.suffix_sort(rep, 256), oracle.suffix_sort(rep, 256))
    finally:
        c.close()


def test_one_pass_induction_equals_count_scan_scatter(hctx, oracle, monkeypatch):
    # round 4, measured and dropped: a source segment of the induction partitioned in ONE pass, the tiles finding
    # the class counts of their predecessors by look-back (induce.hip: k_induce_onepass, KISS_HIP_INDUCE_ONE_PASS in the hooks
    # build) instead of count kernel + scan + scatter.  Same suffix array, also on runs of one base (chains of self-feeding
    # passes), on a text whose passes have thousands of tiles, and in exact order.
    import kiss_amd
    texts = [gen.genome_like(5_000_000, 41), np.zeros(300_000, np.uint8), gen.periodic(600_000, 3, 9, mutations=25),
             np.repeat(gen.iid(40_000, 6), 17)]
    for S in texts:
        for k, algo in ((256, 0), (kiss_amd.K_UNBOUNDED, 1)):
            want = oracle.suffix_sort(S, k)
            monkeypatch.delenv("KISS_HIP_INDUCE_ONE_PASS", raising=False)
            assert np.array_equal(hctx.suffix_sort(S, k, algo=algo), want), (S.size, k, "count + scan + scatter")
            monkeypatch.setenv("KISS_HIP_INDUCE_ONE_PASS", "1")
            assert np.array_equal(hctx.suffix_sort(S, k, algo=algo), want), (S.size, k, "one pass")
    monkeypatch.delenv("KISS_HIP_INDUCE_ONE_PASS", raising=False)


def test_count_pass_on_class_bytes_equals_the_word_path(hctx, oracle, monkeypatch):
    # round 4: every kernel that writes a context word beside SA also writes one class byte, and the count pass of
    # the induction reads the bytes (2 KiB per tile, aligned 16-byte pieces; a byte that says "no bases left" sends its tile
    # to the word path, which refreshes word and byte).  KISS_HIP_NO_CLASS_BYTES (hooks build) is the form of rounds 1-3.
    # Texts: passes of thousands of tiles; runs of one base of 17 and of 40 (context words run empty: the refresh path, in
    # both sweeps); a text of one base; lengths that leave the byte stretches of the tiles at every alignment.
    import kiss_amd
    texts = [gen.genome_like(5_000_000, 43), np.repeat(gen.iid(300_000, 7), 17), np.repeat(gen.iid(90_000, 8), 40),
             np.zeros(300_001, np.uint8)] + [gen.genome_like(700_000 + d, 44 + d) for d in (1, 5, 11)]
    for S in texts:
        for k, algo in ((256, 0), (kiss_amd.K_UNBOUNDED, 1)):
            want = oracle.suffix_sort(S, k)
            monkeypatch.delenv("KISS_HIP_NO_CLASS_BYTES", raising=False)
            assert np.array_equal(hctx.suffix_sort(S, k, algo=algo), want), (S.size, k, "class bytes")
            monkeypatch.setenv("KISS_HIP_NO_CLASS_BYTES", "1")
            assert np.array_equal(hctx.suffix_sort(S, k, algo=algo), want), (S.size, k, "context words")
    monkeypatch.delenv("KISS_HIP_NO_CLASS_BYTES", raising=False)


def test_kernel_class_timing_can_be_limited_to_chosen_classes(oracle):
    # kiss_hip_ctx_set_profiling_mask: HIP events only around the launches of the named classes (what bench.py does for the
    # dominant kernel inside its timed region); the result does not depend on what is timed
    import kiss_amd
    S = gen.genome_like(1_500_000, 9)
    want = oracle.suffix_sort(S, 256)
    c = kiss_amd.Context(max_n=S.size, device=0)
    try:
        assert np.array_equal(c.suffix_sort(S, 256), want)
        assert all(v["launches"] == 0 for v in c.stats()["kernels"].values())          # off by default
        c.set_profiling(True, ["radix_scatter"])
        assert np.array_equal(c.suffix_sort(S, 256), want)
        k = c.stats()["kernels"]
        assert k["radix_scatter"]["launches"] > 0 and k["radix_scatter"]["ms"] > 0
        assert all(v["launches"] == 0 for name, v in k.items() if name != "radix_scatter")
        c.set_profiling(True)
        assert np.array_equal(c.suffix_sort(S, 256), want)
        k = c.stats()["kernels"]
        assert sum(1 for v in k.values() if v["launches"] > 0) >= 8
        c.set_profiling(False)
        c.suffix_sort(S, 256)
        assert all(v["launches"] == 0 for v in c.stats()["kernels"].values())
    finally:
        c.close()


def _random_text(rng, n):
    """small adversarial texts: mixtures of i.i.d. stretches, runs, tandem repeats and copies of earlier pieces"""
    out = []
    total = 0
    while total < n:
        kind = int(rng.integers(0, 5))
        ln = int(rng.integers(1, max(2, n // 3 + 1)))
        if kind == 0:
            piece = rng.integers(0, 4, ln, dtype=np.uint8)
        elif kind == 1:
            piece = np.full(ln, rng.integers(0, 4), dtype=np.uint8)
        elif kind == 2:
            unit = rng.integers(0, 4, int(rng.integers(1, 9)), dtype=np.uint8)
            piece = np.tile(unit, ln // unit.size + 1)[:ln]
        elif kind == 3 and total > 0:
            src = np.concatenate(out)
            a = int(rng.integers(0, src.size))
            piece = src[a:a + ln].copy()
            if piece.size == 0:
                continue
        else:
            piece = rng.integers(0, 2, ln, dtype=np.uint8) * int(rng.integers(1, 4))  # two-letter text
        out.append(piece)
        total += piece.size
    return np.concatenate(out)[:n].astype(np.uint8)


@pytest.mark.parametrize("block", range(8))
def test_randomised_small_texts(ctx, oracle, block):
    # 8 x 40 random (text, k, algorithm) cases per run: stage outputs and SA against the oracle, sizes 0 .. 5000,
    # k around the stride boundaries of the comparator (125, 250, 375) and unbounded, both exact-order paths
    import kiss_amd
    rng = np.random.default_rng(9000 + block)
    ks = [1, 2, 31, 32, 33, 124, 125, 126, 249, 250, 251, 256, 374, 375, 376, 1000, kiss_amd.K_UNBOUNDED]
    for case in range(40):
        n = int(rng.integers(0, 5001)) if case % 4 else int(rng.integers(0, 70))
        S = _random_text(rng, n) if n else np.zeros(0, np.uint8)
        k = ks[int(rng.integers(0, len(ks)))]
        try:
            check_parity(ctx, oracle, S, k)
            if k == kiss_amd.K_UNBOUNDED:
                sa1 = ctx.suffix_sort(S, k, algo=1)
                assert np.array_equal(sa1, oracle.suffix_sort(S, k)), "PREFIX_DOUBLING"
        except AssertionError as e:
            raise AssertionError("block %d case %d: n=%d k=%d text=%s...: %s" % (block, case, n, k, S[:60].tolist(), e))


def test_fallback_paths_in_a_fresh_process():
    # two switches of the hooks build: KISS_HIP_NO_ONESWEEP (radix passes with per-tile histograms instead of
    # look-back) and KISS_HIP_TCAP0 (tiny tied-segment arrays, so every growth path runs).
    # One child process (KISS_AMD_LIB=hooks) sorts a handful of texts with both set and compares with the oracle.
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r"""
import numpy as np, kiss_amd
from tests import gen, oracle_binding
orc = oracle_binding.load()
ctx = kiss_amd.Context(max_n=400_000)
texts = [gen.genome_like(300_000, 3), gen.periodic(200_000, 7, 1, 40), np.zeros(100_000, np.uint8),
         np.tile(np.array([0, 1], np.uint8), 150_000), gen.iid(250_000, 4)]
for S in texts:
    for k, algo in ((256, 0), (32, 0), (kiss_amd.K_UNBOUNDED, 1)):
        assert np.array_equal(ctx.suffix_sort(S, k, algo=algo), orc.suffix_sort(S, k)), (S[:8], k, algo)
print("variants ok")
"""
    env = dict(os.environ, KISS_AMD_LIB="hooks", KISS_HIP_NO_ONESWEEP="1", KISS_HIP_TCAP0="1500", PYTHONPATH=root)
    r = subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "variants ok" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_two_contexts_concurrently(oracle):
    # "thread-safe per distinct ctx": two host threads, each with its own workspace and stream, sort different
    # texts at the same time on the same device (ctypes releases the GIL during the calls)
    import threading
    import kiss_amd
    texts = [gen.genome_like(400_000, 11), gen.periodic(300_000, 5, 3, 30)]
    refs = [oracle.suffix_sort(S, 256) for S in texts]
    refs_exact = [oracle.suffix_sort(S, kiss_amd.K_UNBOUNDED) for S in texts]
    errors = []

    def work(i):
        try:
            c = kiss_amd.Context(max_n=texts[i].size)
            for _ in range(4):
                if not np.array_equal(c.suffix_sort(texts[i], 256), refs[i]):
                    errors.append("thread %d: k = 256 differs" % i)
                if not np.array_equal(c.suffix_sort(texts[i], kiss_amd.K_UNBOUNDED, algo=1), refs_exact[i]):
                    errors.append("thread %d: exact differs" % i)
            c.close()
        except Exception as e:  # noqa: BLE001
            errors.append("thread %d: %r" % (i, e))

    th = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errors, errors


def test_one_shot_calls_reuse_a_cached_context(oracle):
    # kiss_hip_suffix_sort_dna_u32 (what the C++ facade binds) keeps one context per device between calls: grown for a longer
    # text, reused for a shorter one, given back by kiss_hip_release_cached_contexts; same results as a context of one's own
    import torch
    import kiss_amd
    from kiss_amd import _lib
    lib = _lib.load()
    lib.kiss_hip_release_cached_contexts.restype = int
    assert lib.kiss_hip_release_cached_contexts() == 0
    free0 = torch.cuda.mem_get_info(0)[0]
    for n, k in ((200_000, 256), (1_500_000, 256), (300_000, kiss_amd.K_UNBOUNDED), (1_500_000, 32)):
        S = gen.genome_like(n, 7 + n % 13)
        sorter = kiss_amd.KISS2Sorter if k == kiss_amd.K_UNBOUNDED else kiss_amd.KISS1Sorter
        assert np.array_equal(sorter.get_suffix_array_dna(S, k), oracle.suffix_sort(S, k)), (n, k)
    held = free0 - torch.cuda.mem_get_info(0)[0]
    assert held > 20 * 1_500_000  # the context of the longest text is still there ...
    assert lib.kiss_hip_release_cached_contexts() == 0
    assert free0 - torch.cuda.mem_get_info(0)[0] < held // 4  # ... and gone now


def test_sort_beside_verification_and_queries_on_another_context(oracle):
    # what include/kiss_hip.h promises about contexts used from different host threads, one deterministic run per claim:
    # while thread A sorts (k = 256, then exact order) thread B, on its own context, verifies a finished suffix array on the
    # device and runs FM queries.  Everything against the oracle.
    import threading
    import torch
    import kiss_amd
    import kiss_amd.fm_index as fm
    dev = torch.device("cuda", 0)
    S = gen.genome_like(400_000, 11)
    T = gen.genome_like(300_000, 5)
    want256, want_exact = oracle.suffix_sort(S, 256), oracle.suffix_sort(S, kiss_amd.K_UNBOUNDED)
    sa_T = oracle.suffix_sort(T, 256)
    f = fm.FMIndex().build(T)
    ref = oracle.fm_build(T, oracle.suffix_sort(T, 32))
    rng = np.random.default_rng(3)
    pats = np.stack([T[p:p + 24] for p in rng.integers(0, T.size - 24, 5000)]).astype(np.uint8)
    want_q = ref.query_batch(pats)
    errors = []

    def sorter():
        try:
            with kiss_amd.Context(max_n=S.size) as c:
                for _ in range(6):
                    if not np.array_equal(c.suffix_sort(S, 256), want256):
                        errors.append("k = 256 differs")
                    if not np.array_equal(c.suffix_sort(S, kiss_amd.K_UNBOUNDED, algo=1), want_exact):
                        errors.append("exact order differs")
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))

    def checker():
        try:
            with kiss_amd.Context(max_n=T.size) as c:
                d_T = torch.from_numpy(T).to(dev)
                d_SA = torch.from_numpy(sa_T.view(np.int32)).to(dev)
                for _ in range(12):
                    rep = c.verify_sa_dev(d_T.data_ptr(), T.size, d_SA.data_ptr(), 256)
                    if not rep["ok"]:
                        errors.append("verification of a correct SA failed: %r" % (rep,))
                    got = f.query_batch(pats)
                    if not (np.array_equal(got["beg"], want_q["beg"]) and np.array_equal(got["end"], want_q["end"])
                            and got["checksum"] == want_q["checksum"] and np.array_equal(got["offsets"], want_q["offsets"])):
                        errors.append("FM queries differ")
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))

    th = [threading.Thread(target=sorter), threading.Thread(target=checker)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    f.close()
    assert not errors, errors


@pytest.mark.parametrize("k", [130_000, 200_000, 999_000])
def test_bounded_k_with_many_near_end_suffixes(oracle, k):
    # k in the hundred thousands (but < n): more than 65 536 LMS suffixes have fewer than D bases left and follow the
    # comparator's scalar tail (kiss1_core.hpp:120-134).  Round 1 refused these; the reference takes any k.
    import kiss_amd
    S = gen.genome_like(1_000_000, 17)
    with kiss_amd.Context(max_n=S.size, device=0) as c:
        sa = check_parity(c, oracle, S, k)
        assert c.stats()["near_end"] > 4096  # the merge-sort form ran
        assert sa[0] == S.size


def test_near_end_merge_form_equals_pairwise_form(hctx, oracle, monkeypatch):
    # the same inputs through both near-end forms (the switch is read per call)
    rng = np.random.default_rng(77)
    base = rng.integers(0, 4, 30_000, dtype=np.uint8)
    cases = [np.concatenate([base, base[5000:5000 + cut]]) for cut in (10, 124, 125, 126, 256, 300, 375, 376, 500)]
    cases += [gen.periodic(20_000, p, 3 + p, mutations=4) for p in (1, 2, 3, 7, 37, 400)]
    cases += [np.tile(np.array([0, 1], np.uint8), 2000), gen.iid(5, 1), gen.iid(300, 2)]
    for S in cases:
        for k in (32, 256, 1000):
            monkeypatch.setenv("KISS_HIP_NEAR_MERGE_MIN", "1")
            a = check_parity(hctx, oracle, S, k)
            monkeypatch.setenv("KISS_HIP_NEAR_MERGE_MIN", "1000000000")
            b = check_parity(hctx, oracle, S, k)
            assert np.array_equal(a, b)
    monkeypatch.delenv("KISS_HIP_NEAR_MERGE_MIN")


@pytest.mark.parametrize("k", [32, 256, 1000, 0xFFFFFFFF])
def test_pivot_rounds_equal_32_base_rounds(oracle, monkeypatch, k):
    # from the second refinement round on (bounded depth), a big segment is compared with its middle member to the full
    # depth and sorted once on (side, first difference, base) -- k_pivot_lcp; KISS_HIP_NO_PIVOT_ROUNDS keeps the 32-base
    # rounds.  Both against the oracle, on texts whose ties run deep: tandem arrays with sparse and dense mutations,
    # unit lengths around the word and step sizes of the walk, and a high-copy repeat family.
    import kiss_amd
    rng = np.random.default_rng(5)
    texts = [gen.genome_like(2_000_000, 41)]
    for unit, muts in ((3, 300), (31, 200), (32, 200), (33, 200), (127, 150), (128, 150), (171, 400), (500, 100)):
        texts.append(gen.periodic(400_000, unit, 100 + unit, mutations=muts))
    fam = rng.integers(0, 4, 600, dtype=np.uint8)  # 3000 copies of one 600-base element, 3 % divergence
    S = gen.iid(2_400_000, 9)
    for c in range(3000):
        cp = fam.copy()
        m = rng.random(600) < 0.03
        cp[m] = rng.integers(0, 4, int(m.sum()), dtype=np.uint8)
        S[c * 800:c * 800 + 600] = cp
    texts.append(S)
    texts.append(gen.periodic(600_000, 7, 3, mutations=60_000))
    with kiss_amd.Context(max_n=2_400_000, device=0, hooks=True) as c:
        for S in texts:
            want = oracle.suffix_sort(S, k)
            monkeypatch.delenv("KISS_HIP_NO_PIVOT_ROUNDS", raising=False)
            a = c.suffix_sort(S, k)
            monkeypatch.setenv("KISS_HIP_NO_PIVOT_ROUNDS", "1")
            b = c.suffix_sort(S, k)
            monkeypatch.delenv("KISS_HIP_NO_PIVOT_ROUNDS")
            monkeypatch.setenv("KISS_HIP_PIVOT_FROM_ROUND2", "1")
            monkeypatch.setenv("KISS_HIP_PIVOT_SLOTS", "1")
            d = c.suffix_sort(S, k)
            monkeypatch.delenv("KISS_HIP_PIVOT_FROM_ROUND2")
            monkeypatch.delenv("KISS_HIP_PIVOT_SLOTS")
            assert np.array_equal(a, want), "pivot rounds"
            assert np.array_equal(b, want), "32-base rounds"
            assert np.array_equal(d, want), "pivot rounds from the second round on, one deviation per key"


@pytest.mark.parametrize("k", [32, 256])
def test_host_entry_with_page_locked_buffers_downloads_early(oracle, monkeypatch, k):
    # kiss_hip_ctx_suffix_sort_dna_u32 with page-locked host buffers: the finished stretches of SA (L-type part of a bucket
    # after the L sweep has passed it, S-type part after the S sweep has) are downloaded while the sweeps still run;
    # KISS_HIP_NO_EARLY_OUT (read once per process) is covered by the pageable call, which never arms it
    import torch
    import kiss_amd
    S = gen.genome_like(6_000_000, 23)
    want = oracle.suffix_sort(S, k)
    S_pin = torch.from_numpy(S).pin_memory()
    SA_pin = torch.empty(S.size + 1, dtype=torch.int32).pin_memory()
    with kiss_amd.Context(max_n=S.size, device=0) as c:
        for _ in range(2):  # the second call reuses the ctx-owned device copies, the copy stream and its events
            SA_pin.fill_(-1)
            c.suffix_sort_host(S_pin.numpy(), SA_pin.numpy().view(np.uint32), k=k)
            assert np.array_equal(SA_pin.numpy().view(np.uint32), want)
        SA_page = np.full(S.size + 1, 0xFFFFFFFF, dtype=np.uint32)
        c.suffix_sort_host(S, SA_page, k=k)
        assert np.array_equal(SA_page, want)
        # exact order is never downloaded early (the doubling phase rewrites SA after the sweeps)
        SA_pin.fill_(-1)
        c.suffix_sort_host(S_pin.numpy(), SA_pin.numpy().view(np.uint32), k=0xFFFFFFFF, algo=kiss_amd.ALGO_PREFIX_DOUBLING)
        assert np.array_equal(SA_pin.numpy().view(np.uint32), oracle.suffix_sort(S, 0xFFFFFFFF))


def test_host_entry_into_a_destination_that_was_never_touched(monkeypatch):
    # a pageable destination whose pages do not exist yet (numpy.empty of >= 64 MB): helper threads populate them while
    # the device sorts (xfer.hip: kiss_prefault_start); same suffix array as into a touched buffer and as with the
    # helpers switched off
    import kiss_amd
    S = gen.iid(20_000_000, 77)
    with kiss_amd.Context(max_n=S.size, device=0, hooks=True) as c:
        warm = np.zeros(S.size + 1, dtype=np.uint32)
        c.suffix_sort_host(S, warm, k=256)
        assert warm[0] == S.size
        fresh = np.empty(S.size + 1, dtype=np.uint32)   # 80 MB straight from mmap: not one page present
        c.suffix_sort_host(S, fresh, k=256)
        assert np.array_equal(fresh, warm)
        monkeypatch.setenv("KISS_HIP_PREFAULT_THREADS", "1")
        fresh2 = np.empty(S.size + 1, dtype=np.uint32)
        c.suffix_sort_host(S, fresh2, k=256)
        assert np.array_equal(fresh2, warm)
        monkeypatch.setenv("KISS_HIP_NO_PREFAULT", "1")  # helpers switched off (the hook is read per call): the copy takes the faults
        fresh3 = np.empty(S.size + 1, dtype=np.uint32)
        c.suffix_sort_host(S, fresh3, k=256)
        assert np.array_equal(fresh3, warm)
        monkeypatch.delenv("KISS_HIP_NO_PREFAULT")
        monkeypatch.setenv("KISS_HIP_XFER_THREADS", "3")  # (read per call as well; the pool grows on demand)
        fresh4 = np.empty(S.size + 1, dtype=np.uint32)
        c.suffix_sort_host(S, fresh4, k=256)
        assert np.array_equal(fresh4, warm)
        # the ctx keeps its device-side copies of S / SA between calls; they can be handed back
        before = c.workspace_bytes()
        c.release_io_buffers()
        assert before - c.workspace_bytes() == 5 * c.max_n + 4
        c.suffix_sort_host(S, fresh4, k=256)
        assert np.array_equal(fresh4, warm) and c.workspace_bytes() == before


def test_more_than_a_million_near_end_suffixes(oracle):
    # a bounded k of several million bases: E = #LMS suffixes with fewer than D bases left passes 2^20, where the tie
    # marking of the near-end rule used to ask for a grid of 2^32 threads ("invalid configuration argument"); the
    # reference takes any k (kiss1_core.hpp:94-135).  i.i.d. text: the comparisons of the oracle end after a few bases.
    import kiss_amd
    S = gen.iid(4_400_000, 123)
    k = 4_000_000
    want = oracle.suffix_sort(S, k)
    with kiss_amd.Context(max_n=S.size, device=0) as c:
        sa = c.suffix_sort(S, k)
        assert c.stats()["near_end"] > (1 << 20)
        assert np.array_equal(sa, want)


def test_exact_order_lms_rank_array_two_level_one_bins(oracle, monkeypatch):
    # the rank array of the LMS-level doubling (isa.hip: kiss_rank_build_lms) with more than one level-1 bin: indexes
    # p >> 1 beyond 2^24 need a text of more than 2^25 bases; the second bin is short (a few sub-bins + a partial window)
    import kiss_amd
    monkeypatch.setenv("KISS_HIP_ISA_DIRECT_MAX", "1000")
    n = (1 << 25) + 5 * (1 << 17) + 4321
    S = gen.genome_like(n, 23)
    S[3_000_000:3_400_000] = S[30_000_000:30_400_000]   # ties that reach across the two bins
    S[n - 200_000:n - 100_000] = S[1_000_000:1_100_000]
    with kiss_amd.Context(max_n=n, device=0, hooks=True) as c:
        sa = c.suffix_sort(S, 0xFFFFFFFF, algo=1)
        st = c.stats()
        assert st["refine_form"] == 1 and st["refine_items"] > 100_000
        assert np.array_equal(sa, oracle.suffix_sort(S, 0xFFFFFFFF))


def _lms_exact_shapes(shape):
    n = 400_000
    rng = np.random.default_rng(5)
    if shape == "genome":
        return gen.genome_like(1_200_000, 9)
    if shape == "tandem":
        return gen.periodic(600_000, 171, 5, mutations=300)
    if shape == "run_copies":
        # copies of stretches that hold long runs of one base: groups whose common window ends without an LMS position
        S = gen.iid(n, 6)
        S[1000:1700] = 0
        S[1700] = 1
        S[5000:5900] = 3
        S[20_000:20_600] = 2
        for dst in (100_000, 200_000, 300_000):
            S[dst:dst + 30_000] = S[0:30_000]
        return S
    if shape == "homopolymer_arrays":
        # arrays of one base with a few other bases sprinkled in: many LMS suffixes that start with the same long run,
        # one "stuck" group of far more than 64 members
        S = gen.iid(n, 7)
        for a, base in ((10_000, 0), (150_000, 2), (280_000, 0)):
            S[a:a + 100_000] = base
            S[a + rng.integers(0, 100_000, 120)] = (base + 1 + rng.integers(0, 3, 120)) % 4
        return S
    if shape == "a1000c":  # LMS suffixes further apart than any window: the LMS form gives up, the suffix-array form takes over
        return np.tile(np.concatenate([np.zeros(1000, np.uint8), np.ones(1, np.uint8)]), n // 1001 + 1)[:n]
    if shape == "tail_repeat":  # the text ends inside a long copy of its own beginning: near-end suffixes in tie groups
        base = gen.iid(n // 2, 8)
        return np.concatenate([base, gen.iid(100, 9), base[:n // 2 - 100 - 37]])
    if shape == "telomere_end":  # ... and inside a tandem array
        return np.concatenate([gen.iid(n, 10), np.tile(np.array([3, 3, 0, 2, 2, 2], np.uint8), 700)])
    raise ValueError(shape)


@pytest.mark.parametrize("shape", ["genome", "tandem", "run_copies", "homopolymer_arrays", "a1000c", "tail_repeat",
                                   "telomere_end"])
def test_exact_order_lms_level_doubling(oracle, monkeypatch, shape):
    # exact order is reached by rank doubling over the LMS suffixes BEFORE the induction (kiss_lms_exact_refine; the
    # reference's KISS2 order of things, kiss2_core.hpp:835-886); KISS_HIP_NO_LMS_EXACT=1 is the older form (bounded phase,
    # induction, rank doubling over the whole suffix array).  Both give THE suffix array; stats say which one ran.
    import kiss_amd
    S = _lms_exact_shapes(shape)
    want = oracle.suffix_sort(S, 0xFFFFFFFF)
    with kiss_amd.Context(max_n=S.size, device=0, hooks=True) as c:
        for direct_max in (None, "1000"):   # rank array by plain scatter / by the two-level partition (isa.hip)
            if direct_max:
                monkeypatch.setenv("KISS_HIP_ISA_DIRECT_MAX", direct_max)
            sa = c.suffix_sort(S, 0xFFFFFFFF, algo=1)
            st = c.stats()
            assert np.array_equal(sa, want), (shape, direct_max, {k: v for k, v in st.items() if k != "kernels"})
            assert st["refine_form"] == (2 if shape == "a1000c" else 1)
            assert st["refine_depth"] == EXACT_H0
            if shape != "a1000c":
                assert 0 < st["refine_items"] <= st["m"]      # tied LMS suffixes, not tied suffixes
                assert 1 <= st["doubling_rounds"] <= 14
        monkeypatch.delenv("KISS_HIP_ISA_DIRECT_MAX", raising=False)
        monkeypatch.setenv("KISS_HIP_NO_LMS_EXACT", "1")
        assert np.array_equal(c.suffix_sort(S, 0xFFFFFFFF, algo=1), want)
        assert c.stats()["refine_form"] == 2
        monkeypatch.delenv("KISS_HIP_NO_LMS_EXACT")
        # k-ordered calls are untouched by any of this
        assert np.array_equal(c.suffix_sort(S, 256), oracle.suffix_sort(S, 256))
        assert c.stats()["refine_form"] == 0
    # the multi-device entry runs the same doubling on device 0, over the gathered list: tie flags by comparison, the bin
    # sizes of the rank array counted from the list (no device holds the whole ascending list)
    for direct_max in (None, "1000"):
        if direct_max:
            monkeypatch.setenv("KISS_HIP_ISA_DIRECT_MAX", direct_max)
        with kiss_amd.MultiContext([0, 0], max_n=S.size) as mc:
            assert np.array_equal(mc.suffix_sort(S, 0xFFFFFFFF, algo=1), want), (shape, direct_max)
    monkeypatch.delenv("KISS_HIP_ISA_DIRECT_MAX", raising=False)


@pytest.mark.parametrize("shape", ["tandem", "near_end_ties", "all_tied", "genome"])
def test_exact_order_taint_shortcut_equals_comparing_every_pair(oracle, monkeypatch, shape):
    # the exact-order finish looks for tie groups only among TAINTED neighbours (suffixes whose place after the bounded
    # phase may be the tie rule's); KISS_HIP_NO_TAINT compares every adjacent pair of the 256-ordered SA instead.  Same
    # suffix array both ways, through the single call, through stage_refine_exact of the sharded form and through the
    # multi-device entry.
    import torch
    import kiss_amd
    from kiss_amd import multi_gpu
    if shape == "tandem":
        S = gen.periodic(600_000, 171, 5, mutations=300)
    elif shape == "near_end_ties":  # the text ends inside a long exact repeat: near-end suffixes tie with far ones
        base = gen.iid(300_000, 8)
        S = np.concatenate([base, base[:120_000]])
    elif shape == "all_tied":
        S = np.tile(np.array([3, 3, 0, 2, 2, 2], np.uint8), 50_000)
    else:
        S = gen.genome_like(1_500_000, 21)
    want = oracle.suffix_sort(S, 0xFFFFFFFF)
    n = S.size
    for no_taint in (False, True):
        if no_taint:
            monkeypatch.setenv("KISS_HIP_NO_TAINT", "1")
        else:
            monkeypatch.delenv("KISS_HIP_NO_TAINT", raising=False)
        with kiss_amd.Context(max_n=n, device=0, hooks=True) as c:
            assert np.array_equal(c.suffix_sort(S, 0xFFFFFFFF, algo=1), want)
            # sharded form: k = 256 through the stages, then stage_refine_exact
            dev = torch.device("cuda", 0)
            d_S = torch.from_numpy(S).to(dev)
            be = multi_gpu.GpuBackend(c, d_S, 256)
            counts = be.classify(0, n)
            keys, pos, m_far = be.local_lms()
            srt, cw = be.sort(keys[:m_far], pos[:m_far])
            near = pos[m_far:].clone()
            # stage_induce_exact first (the doubling over the LMS suffixes in front of the induction; its tie flags come from
            # comparing tainted neighbours, the hook has no say there): it leaves the sorted list as it was, while
            # refine_exact may regrow the work arrays these views point into
            if not no_taint:
                be.exact_h0 = 256
                SA2 = be.induce(srt, near, counts[:12], far_ctx=cw)
                be.exact_h0 = 0
                if be.last_induce_exact:
                    assert c.stats()["refine_form"] == 1
                    assert np.array_equal(SA2.cpu().numpy().view(np.uint32), want)
                del SA2
            SA = be.induce(srt, near, counts[:12], far_ctx=cw)
            be.refine_exact(SA, 256)
            assert np.array_equal(SA.cpu().numpy().view(np.uint32), want)
        with kiss_amd.MultiContext([0, 0], max_n=n) as mc:
            assert np.array_equal(mc.suffix_sort(S, 0xFFFFFFFF, algo=1), want)
    monkeypatch.delenv("KISS_HIP_NO_TAINT", raising=False)


@pytest.mark.gpu
def test_staged_and_multi_sort_after_an_exact_sort_of_a_sparser_text(oracle):
    """An exact-order sort leaves tie flags behind (one byte per far LMS suffix, at the far end of the context-word array,
    sized for THAT text's LMS count).  A staged sort and a multi-device sort of a text of the same length with four times
    as many LMS suffixes, on the same contexts, must not mark through the stale pointer (stages.hip, multi.hip
    reset ctx->hfar / h_depth at every entry)."""
    import torch
    import kiss_amd
    from kiss_amd import multi_gpu
    n = 600_000
    rng = np.random.default_rng(5)
    sparse = np.repeat(rng.integers(0, 4, n // 8 + 1, dtype=np.uint8), 8)[:n].copy()  # runs of 8: one LMS per ~30 bases
    dense = gen.iid(n, 9)
    with kiss_amd.Context(max_n=n, device=0) as c:
        assert np.array_equal(c.suffix_sort(sparse, 0xFFFFFFFF, algo=1), oracle.suffix_sort(sparse, 0xFFFFFFFF))
        m_sparse = c.stats()["m"]
        d_S = torch.from_numpy(dense).to(torch.device("cuda", 0))
        be = multi_gpu.GpuBackend(c, d_S, 256)
        counts = be.classify(0, n)
        keys, pos, m_far = be.local_lms()
        assert m_far > 2 * m_sparse
        srt, cw = be.sort(keys[:m_far], pos[:m_far])
        SA = be.induce(srt, pos[m_far:].clone(), counts[:12], far_ctx=cw)
        assert np.array_equal(SA.cpu().numpy().view(np.uint32), oracle.suffix_sort(dense, 256))
    with kiss_amd.MultiContext([0, 0], max_n=n) as mc:
        assert np.array_equal(mc.suffix_sort(sparse, 0xFFFFFFFF, algo=1), oracle.suffix_sort(sparse, 0xFFFFFFFF))
        assert np.array_equal(mc.suffix_sort(dense, 256), oracle.suffix_sort(dense, 256))
