#!/usr/bin/env python3
"""torch-free driver for rocprofv3 --pmc runs: one suffix sort of a host-generated text through the C ABI."""
import os
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kiss_amd
from tests import gen
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000_000
kind = sys.argv[2] if len(sys.argv) > 2 else "genome"
S = gen.genome_like(n, 2) if kind == "genome" else gen.iid(n, 2)
ctx = kiss_amd.Context(max_n=n)
sa = ctx.suffix_sort(S, 256)
st = ctx.stats()
print("n", n, "m", st["m"], "device ms", st["ms_total"], "SA[0]", sa[0])
