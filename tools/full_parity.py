#!/usr/bin/env python3
"""Full-size parity check (not part of pytest: minutes of CPU time, tens of GB of host memory).

Generates the bench text on the GPU, sorts it with libkiss_hip.so, downloads the SA and compares it
bit for bit with the CPU oracle run on the host cores.  Prints one JSON line."""
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
    args = ap.parse_args()
    import torch
    import kiss_amd
    from bench import gen_text_device
    from tests import oracle_binding
    dev = torch.device("cuda", 0)
    n, k = args.n, args.k
    S = gen_text_device(n, args.seed, dev)
    SA = torch.empty(n + 1, dtype=torch.int32, device=dev)
    ctx = kiss_amd.Context(max_n=n)
    t0 = time.time()
    ctx.suffix_sort_dev(S.data_ptr(), n, SA.data_ptr(), k=k)
    gpu_s = time.time() - t0
    st = ctx.stats()
    S_h = S.cpu().numpy()
    SA_h = SA.cpu().numpy().view(np.uint32)
    del S, SA
    ctx.close()
    print("gpu sort done: %.3f s wall, %.1f ms device; running the oracle on %d bases ..." % (gpu_s, st["ms_total"], n),
          flush=True)
    orc = oracle_binding.load()
    t0 = time.time()
    ref = orc.suffix_sort(S_h, k)
    cpu_s = time.time() - t0
    equal = bool(np.array_equal(SA_h, ref))
    out = {"n": n, "k": k, "seed": args.seed, "lms": st["m"], "sa_equal_to_oracle": equal,
           "sa_fnv1a64": "%016x" % orc.fnv(SA_h), "gpu_device_ms": st["ms_total"], "oracle_seconds": cpu_s,
           "oracle_threads": orc.num_threads()}
    if not equal:
        bad = np.nonzero(SA_h != ref)[0]
        out["mismatches"] = int(bad.size)
        out["first_bad_index"] = int(bad[0])
    print(json.dumps(out), flush=True)
    return 0 if equal else 1


if __name__ == "__main__":
    sys.exit(main())
