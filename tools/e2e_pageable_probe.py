import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import kiss_amd, bench
n = 3117292070
dev = torch.device("cuda", 0)
S = bench.gen_text_device(n, 2, dev)
ctx = kiss_amd.Context(max_n=n)
S_h = torch.empty(n, dtype=torch.uint8); SA_h = torch.empty(n + 1, dtype=torch.int32)
S_h.copy_(S); torch.cuda.synchronize(); del S
S_np, SA_np = S_h.numpy(), SA_h.numpy().view(np.uint32)
for r in range(5):
    if r in (0, 3):  # a destination that has never been touched (what numpy.empty / new[] hand over)
        SA_h = torch.empty(n + 1, dtype=torch.int32); SA_np = SA_h.numpy().view(np.uint32)
    t0 = time.perf_counter(); ctx.suffix_sort_host(S_np, SA_np, k=256); dt = time.perf_counter() - t0
    st = ctx.stats()
    print("threads %s prefault %s rep %d%s: %.1f ms (h2d %.1f device %.1f d2h %.1f)" % (os.environ.get("KISS_HIP_XFER_THREADS", "8"), "off" if os.environ.get("KISS_HIP_NO_PREFAULT") else "on", r, " (fresh destination)" if r in (0, 3) else "", 1e3 * dt, st["ms_h2d"], st["ms_total"], st["ms_d2h"]), flush=True)
