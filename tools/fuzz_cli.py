#!/usr/bin/env python3
"""Random files through the `kiss` command line (suffix_sort --output-sa, fmindex_build, fmindex_query -b) against the
oracle: reader + sort + .fmi bytes + batch counts.  Usage: fuzz_cli.py [cases] [seed]"""
import os
import struct
import subprocess
import sys
import tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import gen, oracle_binding
from tests.fmi_layout import canonical

KISS = os.path.join(ROOT, "kiss_amd", "kiss")
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
orc = oracle_binding.load()
tmp = tempfile.mkdtemp()
fails = 0
for c in range(cases):
    n = int(np.exp(rng.uniform(np.log(50), np.log(400_000))))
    S = gen.genome_like(n, int(rng.integers(0, 1 << 30))) if rng.random() < 0.7 else gen.periodic(n, int(rng.integers(1, 90)), 3, 5)
    letters = np.frombuffer(b"ACGT" if rng.random() < 0.5 else b"acgt", dtype=np.uint8)[S]
    width = int(rng.choice([1, 7, 60, 70, 1000, 10 ** 7]))
    fasta = rng.random() < 0.7
    nl = b"\r\n" if rng.random() < 0.2 else b"\n"
    raw = []
    if fasta:
        cuts = [0] + sorted(rng.integers(0, n, int(rng.integers(0, 4))).tolist()) + [n]
        for a, b in zip(cuts, cuts[1:]):
            raw.append(b">rec %d" % a + nl)
            if rng.random() < 0.15:
                raw.append(b">second header line" + nl)  # swallowed as the "sequence" line of the header before it?
            seg = letters[a:b].tobytes()
            raw += [seg[i:i + width] + nl for i in range(0, len(seg), width)]
    else:
        seg = letters.tobytes()
        raw += [seg[i:i + width] + nl for i in range(0, len(seg), width)]
    raw = b"".join(raw)
    if rng.random() < 0.3 and raw.endswith(nl):
        raw = raw[:-len(nl)]
    path = os.path.join(tmp, "f%d.fa" % c)
    open(path, "wb").write(raw)
    T = orc.read_sequence(raw)  # what the reference reader makes of the file
    ok = True
    k = int(rng.choice([32, 256, -1]))
    algo = "PREFIX_DOUBLING" if (k == -1 and rng.random() < 0.5) else "PARALLEL_SORTING"
    out = os.path.join(tmp, "sa.bin")
    r = subprocess.run([KISS, "suffix_sort", path, "-k", str(k), "-s", algo, "--output-sa", out], capture_output=True, text=True)
    if r.returncode != 0 or not np.array_equal(np.fromfile(out, dtype="<u4"), orc.suffix_sort(T, k & 0xFFFFFFFF)):
        ok = False
        print("  suffix_sort differs (rc %d) %s" % (r.returncode, r.stderr[-300:]), flush=True)
    if T.size >= 40:
        r = subprocess.run([KISS, "fmindex_build", path], capture_output=True, text=True)
        ref = orc.fm_build(T, orc.suffix_sort(T, 32))
        if r.returncode != 0 or canonical(open(path + ".fmi", "rb").read()) != canonical(ref.serialize()):
            ok = False
            print("  fmindex_build differs (rc %d) %s" % (r.returncode, r.stderr[-300:]), flush=True)
        L, Q = int(rng.integers(1, 33)), int(rng.integers(1, 2000))
        pos = rng.integers(0, T.size - L, Q)
        pats = T[pos[:, None] + np.arange(L)[None, :]].copy()
        pats[::5, 0] = (pats[::5, 0] + 1) % 4
        pf = os.path.join(tmp, "p.bin")
        with open(pf, "wb") as f:
            f.write(struct.pack("<II", L, Q))
            f.write(np.frombuffer(b"ACGT", dtype=np.uint8)[pats.reshape(-1)].tobytes())
        r = subprocess.run([KISS, "fmindex_query", path, "-b", pf], capture_output=True, text=True)
        want = ref.query_batch(pats, want_offsets=False)
        if (r.returncode != 0 or "number of matched locations: %d" % want["total_hits"] not in r.stderr
                or "location checksum: %d" % want["checksum"] not in r.stderr):
            ok = False
            print("  fmindex_query differs (rc %d) %s" % (r.returncode, r.stderr[-300:]), flush=True)
    if not ok:
        fails += 1
        print("FAIL case %d: %d bytes, n=%d fasta=%s width=%d k=%d %s" % (c, len(raw), T.size, fasta, width, k, algo), flush=True)
print("fuzz_cli: %d cases, %d failures" % (cases, fails), flush=True)
sys.exit(1 if fails else 0)
