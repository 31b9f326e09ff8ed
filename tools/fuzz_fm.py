#!/usr/bin/env python3
"""Differential fuzzing of the FM-index on the GPU box: random texts and pattern batches, library vs CPU oracle:
.fmi bytes, ranges, hit counts, checksums and the offsets in the reference's get_offsets order.
Usage: fuzz_fm.py [seconds] [seed]"""
import os
import sys
import time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kiss_amd.fm_index as fm
from tests import gen, oracle_binding
from tests.fmi_layout import canonical

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
orc = oracle_binding.load()
t0 = time.time()
cases = fails = 0
replay = os.environ.get("FUZZ_REPLAY")  # "kind,n,seed": that text only, many pattern batches, details on a mismatch
while time.time() - t0 < budget:
    kind = ["iid", "periodic", "genome", "two_letter"][int(rng.integers(0, 4))]
    n = int(np.exp(rng.uniform(np.log(40), np.log(1_500_000))))
    s = int(rng.integers(0, 1 << 30))
    if replay:
        kind, n, s = replay.split(",")[0], int(replay.split(",")[1]), int(replay.split(",")[2])
    if kind == "iid":
        S = gen.iid(n, s)
    elif kind == "periodic":
        S = gen.periodic(n, int(rng.integers(1, 300)), s, int(rng.integers(0, 30)))
    elif kind == "genome":
        S = gen.genome_like(n, s)
    else:
        S = (rng.integers(0, 2, n, dtype=np.uint8) * int(rng.integers(1, 4))).astype(np.uint8)
    S = np.ascontiguousarray(S, dtype=np.uint8)
    f = fm.FMIndex().build(S)
    ref = orc.fm_build(S, orc.suffix_sort(S, 32))
    ok = canonical(f.to_bytes()) == canonical(ref.serialize())
    if not ok:
        print("  .fmi bytes differ", flush=True)
    for _ in range(200 if replay else 3):
        L = int(rng.integers(1, int(os.environ.get("FUZZ_LMAX", "40")) + 1))
        if L >= n:
            continue
        Q = int(rng.integers(1, 3000))
        pos = rng.integers(0, n - L, Q)
        pats = S[pos[:, None] + np.arange(L)[None, :]].copy()
        mut = rng.random(Q) < 0.3
        col = rng.integers(0, L, Q)
        pats[mut, col[mut]] = rng.integers(0, 4, int(mut.sum()), dtype=np.uint8)
        rnd = rng.random(Q) < 0.1
        pats[rnd] = rng.integers(0, 4, (int(rnd.sum()), L), dtype=np.uint8)
        a = f.query_batch(pats, want_offsets=True)
        b = ref.query_batch(pats, want_offsets=True)
        parts = {"beg": np.array_equal(a["beg"], b["beg"]), "end": np.array_equal(a["end"], b["end"]),
                 "hits": a["total_hits"] == b["total_hits"], "checksum": a["checksum"] == b["checksum"],
                 "offsets_index": np.array_equal(a["offsets_index"], b["offsets_index"]),
                 "offsets": np.array_equal(a["offsets"], b["offsets"])}
        if not all(parts.values()):
            ok = False
            bad = [k for k, v in parts.items() if not v]
            print("  batch L=%d Q=%d differs in %s" % (L, Q, bad), flush=True)
            print("    hits gpu %d ref %d, sum(end-beg) %d" % (a["total_hits"], b["total_hits"],
                                                               int((b["end"].astype(np.int64) - b["beg"]).sum())), flush=True)
            if not parts["offsets_index"]:
                ca, cb = np.diff(a["offsets_index"].astype(np.int64)), np.diff(b["offsets_index"].astype(np.int64))
                for q in np.nonzero(ca != cb)[0][:3].tolist():
                    print("    pattern %d %s: range [%d,%d) gpu count %d ref count %d; gpu %s ref %s" % (
                        q, "".join("ACGT"[c] for c in pats[q]), b["beg"][q], b["end"][q], ca[q], cb[q],
                        a["offsets"][int(a["offsets_index"][q]):int(a["offsets_index"][q + 1])][:6].tolist(),
                        b["offsets"][int(b["offsets_index"][q]):int(b["offsets_index"][q + 1])][:6].tolist()), flush=True)
            if parts["offsets_index"] and not parts["offsets"]:
                ia = a["offsets_index"].astype(np.int64)
                for q in range(Q):
                    x, y = a["offsets"][ia[q]:ia[q + 1]], b["offsets"][ia[q]:ia[q + 1]]
                    if not np.array_equal(x, y):
                        print("    pattern %d: %d hits, range [%d,%d); same set: %s; first diff at %d: gpu %s ref %s" % (
                            q, x.size, b["beg"][q], b["end"][q], np.array_equal(np.sort(x), np.sort(y)),
                            int(np.argmax(x != y)), x[:8].tolist(), y[:8].tolist()), flush=True)
                        break
            if replay:
                break
    f.close()
    cases += 1
    if replay:
        print('replay ok' if ok else 'replay FAILED', flush=True)
        sys.exit(0 if ok else 1)
    if not ok:
        fails += 1
        print("MISMATCH kind=%s n=%d seed=%d" % (kind, n, s), flush=True)
print("fuzz_fm: %d indexes (3 pattern batches each), %d failures, %.0f s, seed %d" % (cases, fails, time.time() - t0, seed),
      flush=True)
sys.exit(1 if fails else 0)
