"""kiss_hip_multi_* (include/kiss_hip.h): ONE process driving several devices -- what `kiss suffix_sort --gpus N` runs.

CPU: the key-range rule (splitters, piece sizes) of the C++ driver against the Python orchestration's, whose exchange
order and stability the world-2 gloo tests of tests/test_multi_gpu.py check with oracle pieces.
GPU: the whole path through the C ABI -- one device (bit-equal to the direct entry), and two / three shares on the one
GPU of the box (a device listed more than once: real partition, real peer-copy plumbing with same-device copies, real
gather), SA against the oracle bit for bit."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

from kiss_amd import _lib
from kiss_amd.multi_gpu import choose_splitters, group_counts
from tests import gen

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def c_splitters(h, G):
    lib = _lib.load()
    h = np.ascontiguousarray(h, dtype=np.uint64)
    sp = np.zeros(max(1, G - 1), dtype=np.uint32)
    gc = np.zeros(G, dtype=np.uint64)
    assert lib.kiss_hip_debug_splitters(h.ctypes.data, h.size, G, sp.ctypes.data, gc.ctypes.data) == 0
    return [int(x) for x in sp[:G - 1]], [int(x) for x in gc]


@pytest.mark.parametrize("seed", range(6))
def test_key_ranges_of_the_c_driver_equal_the_python_orchestration(seed):
    rng = np.random.default_rng(seed)
    bins = 1 << 14
    shapes = [rng.integers(0, 50, bins), np.zeros(bins, np.int64), rng.integers(0, 3, bins) * rng.integers(0, 2, bins)]
    hot = rng.integers(0, 50, bins)
    hot[int(rng.integers(0, bins))] = 5_000_000  # one bin (poly-A, a satellite) holds most of the suffixes
    shapes.append(hot)
    one = np.zeros(bins, np.int64)
    one[bins - 1] = 12345  # everything in the last bin: (AC)^n-like
    shapes.append(one)
    for h in shapes:
        for G in (1, 2, 3, 4, 8, 64):
            sp, gc = c_splitters(h, G)
            assert sp == choose_splitters(h, G)
            assert gc == group_counts(h, sp, G)
            assert sum(gc) == int(h.sum()) and all(a <= b for a, b in zip(sp, sp[1:]))


def test_debug_splitters_rejects_bad_arguments():
    lib = _lib.load()
    h = np.ones(16, dtype=np.uint64)
    sp = np.zeros(4, dtype=np.uint32)
    gc = np.zeros(4, dtype=np.uint64)
    assert lib.kiss_hip_debug_splitters(None, 16, 2, sp.ctypes.data, gc.ctypes.data) == _lib.KISS_HIP_E_INVALID
    assert lib.kiss_hip_debug_splitters(h.ctypes.data, 16, 0, sp.ctypes.data, gc.ctypes.data) == _lib.KISS_HIP_E_INVALID
    assert lib.kiss_hip_debug_splitters(h.ctypes.data, 16, 65, sp.ctypes.data, gc.ctypes.data) == _lib.KISS_HIP_E_INVALID


def test_multi_entry_fails_loudly_without_devices():
    import torch
    if torch.cuda.is_available():
        return
    lib = _lib.load()
    mc = ctypes.c_void_p()
    dev = (ctypes.c_int * 2)(0, 1)
    assert lib.kiss_hip_multi_create(ctypes.byref(mc), dev, 2, 1000) == _lib.KISS_HIP_E_NO_DEVICE
    S = np.zeros(10, np.uint8)
    SA = np.zeros(11, np.uint32)
    assert lib.kiss_hip_suffix_sort_dna_u32_multi(S.ctypes.data, 10, 256, 0, SA.ctypes.data, dev, 2) == _lib.KISS_HIP_E_NO_DEVICE
    # n = 0 needs no device at all, like the single-device entry (kiss1_core.hpp:237-238)
    assert lib.kiss_hip_suffix_sort_dna_u32_multi(None, 0, 256, 0, SA.ctypes.data, dev, 2) == 0 and SA[0] == 0


# ------------------------------------------------------------------ GPU ---------------------------------------------
SHAPES = [
    ("genome", 300_000, 256), ("genome", 1_000_000, 32), ("genome", 200_000, 0xFFFFFFFF),
    ("ttaggg", 65_541, 0xFFFFFFFF),   # ties deeper than the bounded-round exact path: k = 256 + rank doubling on device 0
    ("ac", 200_001, 256),             # (AC)^n: every LMS suffix in ONE key range -> the receiving share regrows its arrays
    ("iid", 60_000, 20_000),          # a third of the LMS suffixes are near-end: the hand-over to device 0
    ("iid", 5, 256), ("iid", 1, 256), ("iid", 2000, 256),
    ("alla", 50_000, 256),            # no LMS suffix at all
]


def make(kind, n):
    if kind == "genome":
        return gen.genome_like(n, 7)
    if kind == "ttaggg":
        return np.tile(np.array([3, 3, 0, 2, 2, 2], np.uint8), n // 6 + 1)[:n].copy()
    if kind == "ac":
        return np.tile(np.array([0, 1], np.uint8), n // 2 + 1)[:n].copy()
    if kind == "alla":
        return np.zeros(n, np.uint8)
    return gen.iid(n, 3)


@pytest.mark.gpu
@pytest.mark.parametrize("devices", [[0], [0, 0], [0, 0, 0]], ids=["1dev", "2shares", "3shares"])
@pytest.mark.parametrize("kind,n,k", SHAPES, ids=["%s-%d-k%d" % (a, b, c if c < 1 << 31 else -1) for a, b, c in SHAPES])
def test_multi_device_entry_equals_the_oracle(oracle, devices, kind, n, k):
    import kiss_amd
    S = make(kind, n)
    want = oracle.suffix_sort(S, k)
    with kiss_amd.MultiContext(devices, max_n=n) as mc:
        for algo in ((0, 1) if k >= n else (0,)):
            sa = mc.suffix_sort(S, k, algo=algo)
            assert np.array_equal(sa, want)
        st = mc.stats()
        assert st["ndev"] == len(devices) and st["n"] == n and st["ms_total"] > 0
    if len(devices) == 1:  # one device: the staged pipeline gives what the direct entry gives
        with kiss_amd.Context(max_n=max(n, 1), device=0) as ctx:
            assert np.array_equal(ctx.suffix_sort(S, k), want)


@pytest.mark.gpu
def test_one_shot_multi_entry_and_facade(oracle):
    import kiss_amd
    S = gen.genome_like(400_000, 9)
    want = oracle.suffix_sort(S, 256)
    assert np.array_equal(kiss_amd.KISS1Sorter.get_suffix_array_dna(S, 256, devices=[0, 0]), want)
    assert np.array_equal(kiss_amd.KISS1Sorter.get_suffix_array_dna(S, 256, devices=[0]), want)
    assert np.array_equal(kiss_amd.KISS2Sorter.get_suffix_array_dna(S, kiss_amd.K_UNBOUNDED, devices=[0, 0]),
                          oracle.suffix_sort(S, 0xFFFFFFFF))
    assert kiss_amd.KISS1Sorter.get_suffix_array_dna(np.zeros(0, np.uint8), 256, devices=[0, 0]).tolist() == [0]


@pytest.mark.gpu
def test_multi_context_is_reusable_and_device_resident_form(oracle):
    import torch
    import kiss_amd
    dev = torch.device("cuda", 0)
    with kiss_amd.MultiContext([0, 0], max_n=2_000_000) as mc:
        for n, k, seed in ((2_000_000, 256, 1), (700_000, 32, 2), (2_000_000, 256, 1), (1, 256, 4)):
            S = gen.genome_like(n, seed)
            d_S = torch.from_numpy(S).to(dev)
            d_SA = torch.empty(n + 1, dtype=torch.int32, device=dev)
            torch.cuda.synchronize()
            mc.suffix_sort_dev(d_S.data_ptr(), n, d_SA.data_ptr(), k=k)
            assert np.array_equal(d_SA.cpu().numpy().view(np.uint32), oracle.suffix_sort(S, k))
        st = mc.stats()
        assert sum(st["piece"]) <= st["m"]
        r0 = mc.rank_context(0).stats()
        assert r0["induce_passes"] > 0


@pytest.mark.gpu
def test_cli_gpus_flag(tmp_path, oracle):
    S = gen.genome_like(500_000, 12)
    fa = tmp_path / "t.fa"
    with open(fa, "w") as f:
        f.write(">x\n")
        txt = "".join("ACGT"[c] for c in S)
        for a in range(0, len(txt), 70):
            f.write(txt[a:a + 70] + "\n")
    exe = os.path.join(ROOT, "kiss_amd", "kiss")
    want = oracle.suffix_sort(S, 256)
    for extra in (["--devices", "0,0"], ["--devices", "0,0,0"], ["--gpus", "1"]):
        out = tmp_path / ("sa_%s.bin" % "_".join(extra).replace(",", "").replace("-", ""))
        p = subprocess.run([exe, "suffix_sort", str(fa), "-k", "256", "--verbose", "--output-sa", str(out)] + extra,
                           capture_output=True, text=True, timeout=300)
        assert p.returncode == 0, p.stderr
        assert "suffix sorting elapsed" in p.stderr and ("n = %d, k = 256" % S.size) in p.stderr
        if "--devices" in extra:
            assert "%d devices" % len(extra[1].split(",")) in p.stderr and "exchange" in p.stderr
        assert np.array_equal(np.fromfile(out, dtype=np.uint32), want)
    # a device that does not exist: an error, not a silent single-GPU run
    p = subprocess.run([exe, "suffix_sort", str(fa), "--gpus", "64"], capture_output=True, text=True, timeout=300)
    assert p.returncode != 0 and "no usable HIP device" in p.stderr
