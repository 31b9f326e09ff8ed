#!/usr/bin/env python3
"""Generates tests/golden/*.npz from the CPU oracle (oracle/kiss_oracle.c).

The reference can not be built in this image (its hot-path headers need the un-vendored spdlog), and its
own tests hold no golden vectors, so these fixtures pin the *oracle* against regressions; they are not
reference outputs.  Inputs are seeded (tests/gen.py); small cases store the full SA, all store its FNV-1a-64."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests import gen, oracle_binding  # noqa: E402

orc = oracle_binding.load()
out = os.path.dirname(os.path.abspath(__file__))
cases = {
    "iid_n50": gen.iid(50, 1), "iid_n1000": gen.iid(1000, 2), "iid_n20003": gen.iid(20003, 3),
    "allA_n300": np.zeros(300, np.uint8), "ac_n300": np.tile(np.array([0, 1], np.uint8), 150),
    "period1_n4000": gen.periodic(4000, 1, 4, 5), "period3_n4000": gen.periodic(4000, 3, 5, 5),
    "period37_n4000": gen.periodic(4000, 37, 6, 5), "period400_n6000": gen.periodic(6000, 400, 7, 5),
    "genome_n30000": gen.genome_like(30000, 8),
}
base = gen.iid(8000, 9)
cases["tail_in_repeat_300"] = np.concatenate([base, base[1000:1300]])
cases["tail_in_repeat_100"] = np.concatenate([base, base[1000:1100]])
for name, S in cases.items():
    for k in (32, 256, 0xFFFFFFFF):
        SA = orc.suffix_sort(S, k)
        np.savez_compressed(os.path.join(out, "%s_k%d.npz" % (name, k if k < 1 << 31 else -1)), S=S, k=np.int64(k), SA=SA,
                            sa_fnv=np.uint64(orc.fnv(SA)))
print("wrote", len(cases) * 3, "fixtures")
