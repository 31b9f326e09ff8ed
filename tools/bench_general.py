#!/usr/bin/env python3
"""Throughput of the general-alphabet entry (kiss_hip_ctx_suffix_sort_u8_dev: exact suffix array of a byte text, SURVEY
section 8 row f3) on device-resident texts, each result checked on the device (kiss_hip_ctx_verify_sa_dev, exactness
proof).  usage: bench_general.py [n]"""
import ctypes
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import kiss_amd  # noqa: E402
from kiss_amd import _lib  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000_000
dev = torch.device("cuda", 0)
ctx = kiss_amd.Context(max_n=n)
lib = _lib.load()
g = torch.Generator(device=dev)
g.manual_seed(5)


def english_like():
    # skewed byte distribution (Zipf over 64 symbols) with planted copies: what a natural-language or source text does
    w = 1.0 / torch.arange(1, 65, dtype=torch.float64, device=dev)
    S = torch.multinomial(w / w.sum(), n, replacement=True, generator=g).to(torch.uint8) + 32
    for i in range(200):
        a, b, ln = (int(x) for x in torch.randint(0, n - 300_000, (3,), generator=g, device=dev).tolist())
        ln = 1000 + ln % 200_000
        S[b:b + ln] = S[a:a + ln]
    return S


shapes = [("bytes uniform over 256 symbols", lambda: torch.randint(0, 256, (n,), dtype=torch.uint8, device=dev, generator=g)),
          ("text-like: Zipf over 64 symbols + 200 copies of up to 200 kB", english_like),
          ("'A'..'D' (the reference's test alphabet, tests/kiss.cpp)",
           lambda: torch.randint(0, 4, (n,), dtype=torch.uint8, device=dev, generator=g) + 65)]
for name, make in shapes:
    S = make()
    SA = torch.empty(n + 1, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(3):
        t = time.perf_counter()
        rc = lib.kiss_hip_ctx_suffix_sort_u8_dev(ctx._ctx, ctypes.c_void_p(S.data_ptr()), n, ctypes.c_void_p(SA.data_ptr()), None)
        assert rc == 0, rc
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t)
    rep = ctx.verify_sa_dev(S.data_ptr(), n, SA.data_ptr(), 0xFFFFFFFF)
    print("%-62s n=%d  %.1f ms = %.2f Gbytes/s  verify %s (exact=%d)" % (
        name, n, 1e3 * best, n / best / 1e9, "ok" if rep["ok"] == 1 else "FAILED %r" % rep, rep["exact"]), flush=True)
    del S, SA
