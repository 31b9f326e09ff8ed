#!/usr/bin/env python3
"""Exact order through the LMS-level doubling (kiss_lms_exact_refine) against the oracle and against the suffix-array form
(KISS_HIP_NO_LMS_EXACT=1), shape by shape, with the library's debug trace.  Runs on the GPU box."""
import os, sys
os.environ.setdefault("KISS_AMD_LIB", "hooks")  # KISS_HIP_* switches exist in the hooks build only
import time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kiss_amd
from tests import gen
from tests import oracle_binding

def shapes(n):
    rng = np.random.default_rng(11)
    yield "iid", gen.iid(n, 3)
    yield "genome", gen.genome_like(n, 4)
    for p in (2, 7, 400, 5000):
        S = np.tile(rng.integers(0, 4, p, dtype=np.uint8), n // p + 1)[:n].copy()
        if p == 400:
            S[rng.integers(0, n, 5)] = 1
        yield "period%d" % p, S
    S = gen.iid(n, 5)
    for i in range(40):
        a = int(rng.integers(0, n - 9000))
        S[a:a + int(rng.integers(300, 9000))] = i % 4
    yield "runs", S
    # copies of stretches that hold long runs of one base: stuck groups
    S = gen.iid(n, 6)
    S[1000:1700] = 0
    S[1700] = 1
    S[5000:5900] = 3
    S[100_000:110_000] = S[0:10_000]
    S[200_000:210_000] = S[0:10_000]
    yield "run_copies", S
    base = gen.iid(n // 2, 8)
    yield "tail_repeat", np.concatenate([base, gen.iid(100, 9), base[:n // 2 - 100 - 37]])
    yield "a1000c", np.tile(np.concatenate([np.zeros(1000, np.uint8), np.ones(1, np.uint8)]), n // 1001 + 1)[:n]
    yield "tandem171", gen.periodic(2 * n, 171, 5, mutations=300)

def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 300_000
    oracle = oracle_binding.load()
    bad = 0
    with kiss_amd.Context(max_n=2 * n + 16, device=0) as c:
        for name, S in shapes(n):
            want = oracle.suffix_sort(S, 0xFFFFFFFF)
            for hook in (None, "1"):
                if hook: os.environ["KISS_HIP_NO_LMS_EXACT"] = hook
                else: os.environ.pop("KISS_HIP_NO_LMS_EXACT", None)
                sys.stderr.write("---- %s %s\n" % (name, "(suffix-array form)" if hook else "(LMS form)"))
                sys.stderr.flush()
                sa = c.suffix_sort(S, 0xFFFFFFFF, algo=1)
                st = c.stats()
                ok = np.array_equal(sa, want)
                bad += 0 if ok else 1
                print("%-12s %-6s %s rounds %2d tied %8d refine %.2f ms total %.2f ms" % (
                    name, "sa" if hook else "lms", "ok  " if ok else "DIFF", st["doubling_rounds"], st["refine_items"],
                    st["ms_refine"], st["ms_total"]), flush=True)
                if not ok:
                    d = np.nonzero(sa != want)[0]
                    print("   differs at %d entries, first %d: got %d want %d" % (d.size, d[0], sa[d[0]], want[d[0]]))
    os.environ.pop("KISS_HIP_NO_LMS_EXACT", None)
    print("lx_probe: %d failures" % bad)
    return 1 if bad else 0

if __name__ == "__main__":
    sys.exit(main())
