#!/usr/bin/env python3
"""Random (world size, n, k) cases of the sharded pipeline with the real stage kernels: `world` processes share the one
GPU of the box, gloo transport staged through host memory (tests/test_multi_gpu.py), SA compared with the oracle.
Usage: fuzz_sharded.py [cases] [seed]"""
import os
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import test_multi_gpu as T

def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    fails = 0
    for c in range(cases):
        world = int(rng.integers(2, 5))
        n = int(np.exp(rng.uniform(np.log(2_000), np.log(3_000_000))))
        k = int(rng.choice([32, 125, 256, 400, 0xFFFFFFFF]))
        try:
            T.test_sharded_pipeline_real_kernels(world, n, k)
            print("ok   world=%d n=%d k=%d" % (world, n, k), flush=True)
        except BaseException as e:  # noqa: BLE001
            fails += 1
            print("FAIL world=%d n=%d k=%d: %r" % (world, n, k, e), flush=True)
    print("fuzz_sharded: %d cases, %d failures" % (cases, fails), flush=True)
    sys.exit(1 if fails else 0)


if __name__ == "__main__":
    main()
