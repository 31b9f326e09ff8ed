"""GPU parity of the device-side FASTA / text reader (kiss_amd/csrc/fasta.hip) against the oracle's literal
restatement of the reference reader (oracle/kiss_oracle_io.c)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import kiss_amd
    c = kiss_amd.Context(max_n=12_000_000, device=0)
    yield c
    c.close()


def roundtrip(ctx, oracle, tmp_path, raw, name="x.fa"):
    p = tmp_path / name
    p.write_bytes(raw)
    got = ctx.read_sequence(p)
    ref = oracle.read_sequence(raw)
    assert got.size == ref.size, "n: gpu %d ref %d" % (got.size, ref.size)
    if not np.array_equal(got, ref):
        bad = np.nonzero(got != ref)[0]
        raise AssertionError("codes differ at %d of %d, first at %d" % (bad.size, got.size, bad[0]))
    return got


SMALL = [
    b"", b"\n", b">", b">\n", b">h", b">h\n", b"A", b"ACGT", b"ACGT\n", b"acgtnN\n", b">h\nACGT", b">h\nACGT\n",
    b">h1 x\nACGT\nNNac\n>h2\n>zzA\n>h3\nTT", b"ACGT\r\n>x\nGG", b">h\r\nAC\r\nGT\r\n", b">h\n\n\nAC\n\nGT\n",
    b"\nAC", b" >h\nAC", b">a\n>b\n>c\n>d\n>e\nAC\n", b">a\n>b\n>c\n>d\nAC\n", b">a\nAC>GT\n>b\nTT",
    b">a\n" + b"ACGT" * 3000 + b"\n>b\n" + b"T" * 5000, bytes(range(256)) * 3, b">" + bytes(range(256)) * 3,
]


@pytest.mark.parametrize("idx", range(len(SMALL)))
def test_small_files(ctx, oracle, tmp_path, idx):
    roundtrip(ctx, oracle, tmp_path, SMALL[idx])


def make_fasta(rng, records, width, crlf=False, lower=0.0, nfrac=0.0, blank=0.0, double_headers=0.0):
    out = []
    nl = b"\r\n" if crlf else b"\n"
    for r in range(records):
        out.append(b">chr%d some description %d" % (r, int(rng.integers(0, 10**9))) + nl)
        if rng.random() < double_headers:
            out.append(b">extra header line directly after a header ACGT" + nl)
        ln = int(rng.integers(0, 40_000))
        seq = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, ln)].copy()
        if lower:
            m = rng.random(ln) < lower
            seq[m] += 32
        if nfrac:
            seq[rng.random(ln) < nfrac] = ord("N")
        s = seq.tobytes()
        for a in range(0, ln, width):
            out.append(s[a:a + width] + nl)
            if rng.random() < blank:
                out.append(nl)
    return b"".join(out)


@pytest.mark.parametrize("case", ["plain60", "crlf", "lower_n", "blank_lines", "double_headers", "width1", "wide",
                                  "many_records"])
def test_generated_fasta(ctx, oracle, tmp_path, case):
    rng = np.random.default_rng(abs(hash(case)) % 1000)
    kw = dict(records=60, width=60)
    if case == "crlf":
        kw.update(crlf=True)
    elif case == "lower_n":
        kw.update(lower=0.3, nfrac=0.05)
    elif case == "blank_lines":
        kw.update(blank=0.05)
    elif case == "double_headers":
        kw.update(double_headers=0.5)
    elif case == "width1":
        kw.update(records=8, width=1)
    elif case == "wide":
        kw.update(width=100_000)
    elif case == "many_records":
        kw.update(records=4000, width=70)
    raw = make_fasta(rng, **kw)
    if case == "many_records":  # short records: a header every other line
        raw = b"".join(b">r%d\n%s\n" % (i, b"ACGTTGCA"[: 1 + i % 8]) for i in range(200_000))
    roundtrip(ctx, oracle, tmp_path, raw)


def test_text_mode_and_sort_from_file(ctx, oracle, tmp_path):
    # the whole CLI data path without host-side base handling: file -> device codes -> suffix array
    import ctypes
    from tests import gen
    S = gen.genome_like(1_000_000, 3)
    raw = b"".join(np.frombuffer(b"ACGT", dtype=np.uint8)[S[a:a + 80]].tobytes() + b"\n" for a in range(0, S.size, 80))
    got = roundtrip(ctx, oracle, tmp_path, raw, "plain.txt")
    assert np.array_equal(got, S)
    p = tmp_path / "g.fa"
    p.write_bytes(b">chr1\n" + raw)
    d_S, n = ctx.load_text_file(p)
    try:
        assert n == S.size
        import torch
        SA = torch.empty(n + 1, dtype=torch.int32, device="cuda:0")
        ctx.suffix_sort_dev(d_S, n, SA.data_ptr(), k=256)
        torch.cuda.synchronize()
        sa = SA.cpu().numpy().view(np.uint32)
    finally:
        ctx.free_dev(d_S)
    assert np.array_equal(sa, oracle.suffix_sort(S, 256))


def test_missing_file_is_an_error(ctx):
    import kiss_amd
    with pytest.raises(kiss_amd.KissHipError):
        ctx.load_text_file("/nonexistent/file.fa")


def test_random_byte_soup(ctx, oracle, tmp_path):
    # 300 random files over an alphabet rich in '>' and line ends: header / sequence alternation inside runs of
    # '>' lines, empty lines, CR, files without a final newline -- against the literal restatement of the reader
    rng = np.random.default_rng(4242)
    alphabet = np.frombuffer(b"ACGTNacgt>>>\n\n\n\r xy", dtype=np.uint8)
    for case in range(300):
        n = int(rng.integers(0, 3000)) if case % 3 else int(rng.integers(0, 12))
        raw = alphabet[rng.integers(0, alphabet.size, n)].tobytes()
        if case % 2:
            raw = b">" + raw
        try:
            roundtrip(ctx, oracle, tmp_path, raw, "soup.fa")
        except AssertionError as e:
            raise AssertionError("case %d (%r...): %s" % (case, raw[:80], e))
