#!/usr/bin/env python3
"""End-to-end wall time of the `kiss` command line on a synthetic FASTA of n bases (default: chm13 size):
file read -> pinned upload -> device-side parse -> suffix sort.  Usage: cli_e2e.py [n] [extra kiss args...]"""
import os
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else bench.CHM13_N
extra = sys.argv[2:]
path = "/tmp/kiss_e2e_%d.fa" % n
t = time.time()
S = bench.gen_text_device(n, 2, torch.device("cuda:0"))
lut = torch.tensor([65, 67, 71, 84], dtype=torch.uint8, device=S.device)
W = 80
with open(path, "wb") as f:
    chroms = 24
    per = (n + chroms - 1) // chroms
    for c in range(chroms):
        a, b = c * per, min(n, (c + 1) * per)
        if a >= b:
            break
        f.write(b">chr%d synthetic\n" % (c + 1))
        seg = lut[S[a:b].long()]
        full = (b - a) // W * W
        lines = torch.empty((full // W, W + 1), dtype=torch.uint8, device=S.device)
        lines[:, :W] = seg[:full].view(-1, W)
        lines[:, W] = 10
        f.write(lines.cpu().numpy().tobytes())
        if full < b - a:
            f.write(seg[full:].cpu().numpy().tobytes() + b"\n")
        del lines, seg
del S
torch.cuda.empty_cache()
print("wrote %s (%.2f GB) in %.1f s" % (path, os.path.getsize(path) / 1e9, time.time() - t), flush=True)
for rep in range(2):  # second run: page cache warm
    t = time.time()
    r = subprocess.run([os.path.join(ROOT, "kiss_amd", "kiss"), "suffix_sort", path, "--verbose"] + extra,
                       capture_output=True, text=True)
    print("run %d: wall %.3f s, rc %d\n%s" % (rep, time.time() - t, r.returncode, r.stderr.strip()), flush=True)
os.remove(path)
