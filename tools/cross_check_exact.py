#!/usr/bin/env python3
"""Full-size cross-check of the two exact-order paths (no oracle involved): the suffix array of the bench text
from PARALLEL_SORTING with k >= n (32 bases per round) and from PREFIX_DOUBLING (bounded phase + rank doubling
over the full SA) must be bit-equal.  Usage: cross_check_exact.py [n] [seed]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import kiss_amd  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else bench.CHM13_N
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 2
dev = torch.device("cuda:0")
S = bench.gen_text_device(n, seed, dev)
ctx = kiss_amd.Context(max_n=n, device=0)
out = []
for algo in (0, 1):
    SA = torch.empty(n + 1, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    t = time.time()
    ctx.suffix_sort_dev(S.data_ptr(), n, SA.data_ptr(), k=kiss_amd.K_UNBOUNDED, algo=algo)
    torch.cuda.synchronize()
    st = ctx.stats()
    print("algo %d: %.1f ms wall, device %.1f ms, lms rounds %d, doubling rounds %d, tied at depth %d: %d" % (
        algo, 1e3 * (time.time() - t), st["ms_total"], st["lms_rounds"], st["doubling_rounds"], st["refine_depth"],
        st["refine_items"]), flush=True)
    out.append(SA)
same = bool(torch.equal(out[0], out[1]))
print("n = %d seed %d: exact SA of the two paths bit-equal: %s" % (n, seed, same))
sys.exit(0 if same else 1)
