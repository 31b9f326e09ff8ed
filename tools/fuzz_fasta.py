#!/usr/bin/env python3
"""Differential fuzzing of the device-side FASTA / text reader against the oracle's restatement of the reference
reader: random byte soups (rich in '>' and line ends) of up to a few hundred KiB, so that runs of header lines, empty
lines and CRs fall on the reader's 4 KiB tile boundaries.  Usage: fuzz_fasta.py [seconds] [seed]"""
import os
import sys
import tempfile
import time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kiss_amd
from tests import oracle_binding

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
orc = oracle_binding.load()
ctx = kiss_amd.Context(max_n=4_000_000)
alphabets = [b"ACGTNacgt>>>\n\n\n\r xy", b"ACGT\n", b"ACGT>\n", b">\n", b">\nA", b"ACGTACGTACGTACGTACGTACGTACGT\n>", b"\r\n>AC",
             bytes(range(256))]
t0 = time.time()
cases = fails = 0
tmp = tempfile.mkdtemp()
path = os.path.join(tmp, "f.fa")
while time.time() - t0 < budget:
    ab = np.frombuffer(alphabets[int(rng.integers(0, len(alphabets)))], dtype=np.uint8)
    n = int(np.exp(rng.uniform(np.log(1), np.log(400_000))))
    raw = ab[rng.integers(0, ab.size, n)]
    # plant long lines / long header runs around multiples of 4096
    for _ in range(int(rng.integers(0, 6))):
        at = int(rng.integers(0, max(1, n // 4096 + 1))) * 4096 + int(rng.integers(-3, 4))
        what = [b">", b"\n", b">\n>\n>\n", b"\n\n", b"\r\n", b">x\n>y\n"][int(rng.integers(0, 6))]
        if 0 <= at and at + len(what) <= n:
            raw[at:at + len(what)] = np.frombuffer(what, dtype=np.uint8)
    raw = raw.tobytes()
    if rng.random() < 0.5:
        raw = b">" + raw
    with open(path, "wb") as fh:
        fh.write(raw)
    got = ctx.read_sequence(path)
    ref = orc.read_sequence(raw)
    cases += 1
    if got.size != ref.size or not np.array_equal(got, ref):
        fails += 1
        np.save(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", "fuzz_fasta_fail_%d.npy" % fails),
                np.frombuffer(raw, dtype=np.uint8))
        print("MISMATCH case %d: %d bytes, n gpu %d ref %d" % (cases, len(raw), got.size, ref.size), flush=True)
print("fuzz_fasta: %d files, %d failures, %.0f s, seed %d" % (cases, fails, time.time() - t0, seed), flush=True)
sys.exit(1 if fails else 0)
