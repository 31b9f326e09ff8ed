#!/usr/bin/env python3
"""Full-size parity check (not part of pytest: minutes of CPU time, tens of GB of host memory).

Generates the bench text on the GPU, sorts it with libkiss_hip.so, verifies the SA on the device, downloads it and
compares it bit for bit with
  --checker oracle : oracle/kiss_oracle.c (our restatement), and/or
  --checker ref    : oracle/_ref/libkiss_ref.so -- the reference's OWN get_lms / put_lms_suffix / induced_sort (compiled
                     unmodified) around the restated LMS sort of oracle/ref_driver.cpp, on the host cores.
Prints one JSON line; with --write-pin also the entry for tests/golden/full_size_pins.json (digest + FNV-1a-64 of the SA
all checkers agreed on)."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=1_000_000_000)
    ap.add_argument("--k", type=int, default=256)
    ap.add_argument("--seed", type=int, default=2)
    ap.add_argument("--checker", choices=["oracle", "ref", "both"], default="both")
    ap.add_argument("--threads", type=int, default=32, help="threads of the reference's OpenMP regions (--checker ref)")
    ap.add_argument("--write-pin", default=None)
    args = ap.parse_args()
    import torch
    import kiss_amd
    from kiss_amd import sorter
    from bench import gen_text_device
    from tests import oracle_binding, ref_binding
    dev = torch.device("cuda", 0)
    n, k = args.n, args.k
    S = gen_text_device(n, args.seed, dev)
    SA = torch.empty(n + 1, dtype=torch.int32, device=dev)
    ctx = kiss_amd.Context(max_n=n)
    t0 = time.time()
    ctx.suffix_sort_dev(S.data_ptr(), n, SA.data_ptr(), k=k)
    gpu_s = time.time() - t0
    st = ctx.stats()
    rep = ctx.verify_sa_dev(S.data_ptr(), n, SA.data_ptr(), k)
    S_h = S.cpu().numpy()
    SA_h = SA.cpu().numpy().view(np.uint32)
    del S, SA
    ctx.close()
    print("gpu sort done: %.3f s wall, %.1f ms device, device verify ok=%d (%.1f ms); running the checker(s) on %d bases ..."
          % (gpu_s, st["ms_total"], rep["ok"], rep["ms"], n), flush=True)
    out = {"n": n, "k": k, "seed": args.seed, "lms": st["m"], "device_verify_ok": bool(rep["ok"]),
           "tied_pairs": rep["tied_pairs"], "sa_digest": "%016x" % rep["digest"],
           "sa_fnv1a64": "%016x" % sorter.fnv1a64(SA_h), "gpu_device_ms": st["ms_total"]}
    assert sorter.sa_digest(SA_h) == rep["digest"], "device digest != host digest of the downloaded SA"
    ok = bool(rep["ok"])
    if args.checker in ("oracle", "both"):
        orc = oracle_binding.load()
        t0 = time.time()
        want = orc.suffix_sort(S_h, k)
        out["oracle_seconds"], out["oracle_threads"] = time.time() - t0, orc.num_threads()
        out["sa_equal_to_oracle"] = bool(np.array_equal(SA_h, want))
        ok = ok and out["sa_equal_to_oracle"]
        if not out["sa_equal_to_oracle"]:
            bad = np.nonzero(SA_h != want)[0]
            out["oracle_mismatches"], out["oracle_first_bad_index"] = int(bad.size), int(bad[0])
        del want
    if args.checker in ("ref", "both"):
        ref = ref_binding.load()
        t0 = time.time()
        want = ref.suffix_sort(S_h, k, T=args.threads)
        out["ref_seconds"], out["ref_threads"] = time.time() - t0, args.threads
        out["sa_equal_to_reference_code"] = bool(np.array_equal(SA_h, want))
        out["reference_code"] = ("reference get_lms + put_lms_suffix + induced_sort compiled unmodified; LMS sort of "
                                 "kiss1_core.hpp:41-144 restated in oracle/ref_driver.cpp")
        ok = ok and out["sa_equal_to_reference_code"]
        if not out["sa_equal_to_reference_code"]:
            bad = np.nonzero(SA_h != want)[0]
            out["ref_mismatches"], out["ref_first_bad_index"] = int(bad.size), int(bad[0])
        del want
    print(json.dumps(out), flush=True)
    if ok and args.write_pin:
        key = "chm13size_seed%d_k%d" % (args.seed, k) if n == 3_117_292_070 else "n%d_seed%d_k%d" % (n, args.seed, k)
        pin = {key: {"digest": out["sa_digest"], "sa_fnv1a64": out["sa_fnv1a64"], "n": n, "k": k, "seed": args.seed,
                     "lms": st["m"],
                     "source": "tools/full_parity.py --checker %s: HIP SA bit-equal to %s" % (
                         args.checker, " and ".join(x for x, c in (("oracle/kiss_oracle.c", "oracle"),
                                                                   ("oracle/_ref (reference get_lms/placement/induction)", "ref"))
                                                    if args.checker in (c, "both")))}}
        with open(args.write_pin, "w") as f:
            json.dump(pin, f, indent=1)
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
