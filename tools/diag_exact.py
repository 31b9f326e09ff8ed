#!/usr/bin/env python3
"""Where does the wall time of the FIRST exact-order sort of a heavily repetitive text go?  (round 3: 6.2 s for a sort
whose later calls take 0.6 s)  Prints host wall-clock around the calls; run with KISS_HIP_DEBUG=1 for the library's own
lines (reserve timings, rounds)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import kiss_amd
from tests import gen
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_400_000_000
dev = torch.device("cuda", 0)
t = time.time(); ctx = kiss_amd.Context(max_n=n); print("ctx create %.3f s, workspace %.1f GB" % (time.time() - t, ctx.workspace_bytes() / 1e9), flush=True)
S = torch.from_numpy(gen.periodic(n, 171, 4, n // 100)).to(dev)
SA = torch.empty(n + 1, dtype=torch.int32, device=dev)
torch.cuda.synchronize()
for rep in range(3):
    for k, algo in ((256, 0), (0xFFFFFFFF, 1)):
        t = time.time()
        ctx.suffix_sort_dev(S.data_ptr(), n, SA.data_ptr(), k=k, algo=algo)
        st = ctx.stats()
        print("rep %d k=%d algo %d: wall %.3f s device %.1f ms (refine %.1f ms) workspace %.1f GB" % (rep, k, algo, time.time() - t, st["ms_total"], st["ms_refine"], ctx.workspace_bytes() / 1e9), flush=True)
