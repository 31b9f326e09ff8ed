#!/usr/bin/env python3
"""bench_fm.py -- FM-index queries/sec on MI355X (the second metric of BASELINE.json; configs[2]).

Workload: index of a dm-sized synthetic text (n = 48 800 648, built on the GPU with the k = 32 sort like the
reference's fmindex_build) + Q x 32-base patterns (90 % sampled from the text, 10 % with one substitution,
SURVEY.md section 8(d) C3).  A step = one batched get_range + get_offsets pass over all Q patterns with the
index and the patterns resident in HBM.  Prints ONE JSON line.  (bench.py stays the headline suffix_sort bench.)
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
DM_N = 48_800_648  # reference README.md:88


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=DM_N)
    ap.add_argument("--queries", type=int, default=1_000_000)
    ap.add_argument("--len", type=int, default=32)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--cpu-queries", type=int, default=200_000, help="patterns of the single-thread CPU baseline (0 = skip)")
    args = ap.parse_args()
    result_out = os.fdopen(os.dup(1), "w")  # stdout = the JSON line only (libraries print to fd 1 too)
    sys.stdout.flush()
    os.dup2(2, 1)
    import torch
    import kiss_amd.fm_index as fm
    from bench import gen_text_device
    dev = torch.device("cuda", 0)
    S = gen_text_device(args.n, 1, dev)
    S_host = S.cpu().numpy()
    t0 = time.perf_counter()
    f = fm.FMIndex().build(S_host)
    build_s = time.perf_counter() - t0
    rng = np.random.default_rng(3)
    Q, L = args.queries, args.len
    pos = rng.integers(0, args.n - L, Q)
    idx = pos[:, None] + np.arange(L)[None, :]
    pats = S_host[idx]
    mut = rng.random(Q) < 0.1
    col = rng.integers(0, L, Q)
    pats[mut, col[mut]] = (pats[mut, col[mut]] + 1 + rng.integers(0, 3, int(mut.sum()))) % 4
    pats = np.ascontiguousarray(pats, dtype=np.uint8)
    d_p = torch.from_numpy(pats).to(dev)
    f._context(max(f.N, 4 * Q)).set_profiling(True)
    for _ in range(args.warmup):
        r = f.query_batch(None, want_offsets=False, d_patterns=d_p)
    torch.cuda.synchronize()
    kms0 = f._ctx.stats()["kernels"]["fm_query"]["ms"]  # the library accumulates kernel time across calls
    t0 = time.perf_counter()
    for _ in range(args.steps):
        r = f.query_batch(None, want_offsets=False, d_patterns=d_p)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    kms = f._ctx.stats()["kernels"]["fm_query"]["ms"] - kms0
    qps = Q * args.steps / el
    hits = r["total_hits"]
    # algorithmic bytes of THIS formulation: backward search = 2 LF steps of 24 B per pattern character; locate walks
    # ranges, not hits (<= 85 ranges per pattern), so a hit costs its sampled-SA word in and its offset out (8 B).
    # SURVEY.md 8(d) prices the reference's walk at 48 L per pattern + 64 B per hit (one LF walk per hit).
    bytes_per_step = 48.0 * L * Q + 8.0 * hits
    survey_bytes_per_step = 48.0 * L * Q + 64.0 * hits
    kernel_s = 1e-3 * kms / args.steps
    out = {
        "metric": "FM-index queries/sec (batched get_range + get_offsets, 32-base patterns)",
        "value": qps, "unit": "queries/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * el / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "u32", "data": "synthetic",
        "config": {"workload": "fmindex_query --batch: %d x %d-base patterns on a dm-sized synthetic index (n=%d), "
                               "index + patterns resident in HBM" % (Q, L, args.n),
                   "hits": hits, "checksum": r["checksum"], "index_build_s_incl_sort_and_upload": build_s},
        "roofline": {"bound": "hbm", "kernel": "k_fm_range+k_fm_locate", "achieved": bytes_per_step / kernel_s / 1e9,
                     "peak": 8000.0, "unit": "GB/s", "frac": bytes_per_step / kernel_s / 1e9 / 8000.0, "traffic": None,
                     "kernel_ms_per_step": 1e3 * kernel_s,
                     "algorithmic_bytes_per_step": bytes_per_step,
                     "survey_8d_bytes_per_step": survey_bytes_per_step,
                     "note": "bytes = 48 L per pattern + 8 per hit (range-wise locate); a dm-sized index (~85 MB) is "
                             "Infinity-Cache resident, so this is a latency-bound gather workload and the HBM fraction "
                             "says little"},
    }
    if args.cpu_queries > 0:
        from tests import oracle_binding
        orc = oracle_binding.load()
        t0 = time.perf_counter()
        ref = orc.fm_build(S_host, orc.suffix_sort(S_host, 32))
        cq = min(Q, args.cpu_queries)
        t1 = time.perf_counter()
        rr = ref.query_batch(pats[:cq], want_offsets=False)
        dt = time.perf_counter() - t1
        out["cpu_baseline"] = {"value": cq / dt, "unit": "queries/s", "cores": 1, "kind": "port",
                               "sample": "first %d patterns of the same batch, single thread like the reference loop "
                                         "(fmindex_query.hpp:79-95), %.1f s" % (cq, dt)}
        sub = f.query_batch(pats[:cq], want_offsets=False)
        out["config"]["parity_vs_oracle_on_sample"] = bool(sub["total_hits"] == rr["total_hits"] and
                                                           sub["checksum"] == rr["checksum"])
    result_out.write(json.dumps(out) + "\n")
    result_out.flush()


if __name__ == "__main__":
    main()
