#!/usr/bin/env python3
"""bench.py -- bases/sec of k-ordered suffix sorting on MI355X (BASELINE.json metric).

A "step" is one pass of the hot path (pack -> get_lms -> k-ordered LMS sort -> induction) over one
synthetic chm13-sized text (n = 3 117 292 070, k = 256) that is already resident in HBM when the
timed region starts; the SA stays in HBM.  One process per GPU.

Launch forms:
  python bench.py --gpus 1 ...                       one process, one GPU
  python bench.py --gpus N ...   (WORLD_SIZE unset)  this process starts N fresh rank processes itself (one per GPU,
                                                     RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set), relays rank 0's JSON
                                                     line and exits with the worst child status; it never touches the GPU
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...   (WORLD_SIZE = N in the environment)
                                                     this process IS one rank; --gpus must equal WORLD_SIZE

For N > 1 the default (--mode sharded) sorts ONE text: the LMS sort is sharded by key range over the ranks (RCCL
all-to-all of the LMS list, gather of the sorted pieces, induction on rank 0) -> "scaling": "strong"; --mode replicas
sorts one independent text per rank (no data-path collective) -> "scaling": "weak".  A rank that fails exits non-zero;
nothing is retried inside a process group whose collective has failed.  Only the self-launching parent may start a
FRESH set of rank processes in --mode replicas after a sharded failure, and the line then says "scaling": "weak" and
carries config.sharded_error.

Prints ONE JSON line on rank 0 (see the contract in the task description).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CHM13_N = 3_117_292_070  # reference README.md:101
HBM_PEAK_GBS = 8000.0    # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)

# algorithmic bytes per item of each kernel class (DESIGN.md section "Kernels"): bytes that must move
# through HBM for one item even with perfect caching
ALGO_BYTES = {
    "radix_scatter": 24.0,   # read key64+pos32, write key64+pos32 (a seg32 adds 8 in refinement rounds; not counted)
    "radix_hist": 8.0,       # read key64
    "induce_scatter": 16.0,  # read pos32+ctx32, write pos32+ctx32 of the induced item
    "induce_count": 4.0,     # read ctx32
    "classify": 0.25,        # 2-bit text, per base per launch
    "pack": 1.25,            # read 1 byte, write 2 bits
    "scan": 8.0,             # read + write u32
    "place": 12.0,           # read pos32, write pos32 + ctx32 (+ one text gather)
    "keygather": 12.0,       # read pos32, write key64 (+ one text gather)
    "flag_compact": 28.0,
    "induce_small": 16.0,
}


DOMINANT_CLASS = "radix_scatter"  # timed inside the timed region; bench checks it against the full breakdown

# HBM traffic measured with rocprofv3 PMC passes (tools/pmc.sh; profiles/r02_pmc_fetch_n5e8.csv and
# profiles/r02_pmc_write_tcc_n5e8.csv, round 1: r01_pmc_*; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for wide
# streaming reads) per algorithmic byte of the same launches (n = 5e8, 41 dispatches of k_radix_scatter<0,false,true>: the
# five round-0 passes of 145.4e6 items plus the small chain-collapse sorts): (2 x 4.609e6 KB + 9.271e6 KB) /
# (5 x 145.4e6 x 24 B + <= 0.3e9 B) = 1.04 .. 1.06 in all three rounds (the kernel is unchanged; round 3: profiles/r03_pmc_*.csv)
MEASURED_TRAFFIC_PER_ALGO_BYTE = {"radix_scatter": 1.05}


def gen_text_device(n, seed, device, harsh=False):
    """Genome-like synthetic text on the GPU (torch ops only; deterministic for a given seed).
    Same recipe as tests/gen.py::genome_like.
    harsh=True (bench.py --harsh; sensitivity runs, not the headline): satellite content closer to chm13's -- tandem
    arrays covering 6 % of the text with lengths up to 30 Mb, a third of them alpha-satellite-like higher-order repeats
    (12 monomers of 171 bases, 25 % apart from each other, the 2052-base unit repeated with 1 % divergence between
    copies) -- instead of 3 % in arrays of at most 1 Mb."""
    import torch
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    S = torch.randint(0, 4, (n,), dtype=torch.uint8, device=device, generator=g)
    if n < 1_000_000:
        return S
    cpu = np.random.default_rng(seed)

    def mutate(seg, rate):
        if rate <= 0:
            return seg
        mask = torch.rand(seg.shape, device=device, generator=g) < rate
        rnd = torch.randint(0, 4, seg.shape, dtype=torch.uint8, device=device, generator=g)
        return torch.where(mask, rnd, seg)

    # interspersed repeats: 10 families x 300 bases, 10 % of the text, 10 % divergence
    fams = torch.randint(0, 4, (10, 300), dtype=torch.uint8, device=device, generator=g)
    n_ins = int(0.10 * n / 300)
    stratum = n // n_ins  # one insertion per stratum of ~3000 bases: no overlapping writes -> deterministic text
    chunk = 200_000
    ar = torch.arange(300, device=device, dtype=torch.int64)
    for b in range(0, n_ins, chunk):
        c = min(chunk, n_ins - b)
        base = (torch.arange(b, b + c, device=device, dtype=torch.int64)) * stratum
        pos = base + torch.randint(0, stratum - 300, (c,), device=device, generator=g, dtype=torch.int64)
        fam = torch.randint(0, 10, (c,), device=device, generator=g, dtype=torch.int64)
        vals = mutate(fams[fam], 0.10)
        idx = (pos[:, None] + ar[None, :]).reshape(-1)
        S[idx] = vals.reshape(-1)
        del idx, vals
    # segmental duplications: 5 %, log-uniform 1e3..1e5, half exact
    budget = int(0.05 * n)
    while budget > 0:
        L = int(np.exp(cpu.uniform(np.log(1e3), np.log(1e5))))
        src = int(cpu.integers(0, n - L))
        dst = int(cpu.integers(0, n - L))
        seg = S[src:src + L].clone()
        S[dst:dst + L] = seg if cpu.random() < 0.5 else mutate(seg, 0.01)
        budget -= L
    # tandem arrays: 3 % (harsh: 6 %, arrays up to 30 Mb, higher-order repeats)
    budget = int((0.06 if harsh else 0.03) * n)
    units = [1, 2, 3, 4, 5, 6, 12, 171]
    lmax = min(30_000_000, n // 8) if harsh else 1_000_000
    while budget > 0:
        u = units[int(cpu.integers(0, len(units)))]
        L = int(min(lmax, max(200, 200 * (1.0 / max(1e-6, cpu.random())) ** (1.0 if harsh else 0.7))))
        p = int(cpu.integers(0, n - L))
        if harsh and cpu.random() < 1.0 / 3.0:  # higher-order repeat: 12 diverged monomers form the repeated unit
            mono = torch.randint(0, 4, (171,), dtype=torch.uint8, device=device, generator=g)
            unit = torch.cat([mutate(mono, 0.25) for _ in range(12)])
            L = max(L, 20 * unit.numel())
            L = min(L, n // 8)
            p = int(cpu.integers(0, n - L))
            S[p:p + L] = mutate(unit.repeat(L // unit.numel() + 1)[:L], 0.01)
        else:
            unit = torch.randint(0, 4, (u,), dtype=torch.uint8, device=device, generator=g)
            S[p:p + L] = mutate(unit.repeat(L // u + 1)[:L], 0.005)
        budget -= L
    tel = torch.tensor([3, 3, 0, 2, 2, 2], dtype=torch.uint8, device=device).repeat(500)
    for c in range(24):
        e = (c + 1) * (n // 24)
        S[e - tel.numel():e] = tel
    return S


FULL_SIZE_PINS = os.path.join(ROOT, "tests", "golden", "full_size_pins.json")


def verify_leg(ctx, S, SA, n, k, seed, iid, algo, with_fnv, harsh=False):
    """After the timed region: the LAST suffix array is checked on the device (kiss_hip_ctx_verify_sa_dev: permutation,
    SA[0] = n, the reference's own k-order test for every adjacent pair -- or, for k >= n, the linear-time proof of
    exactness) and hashed.  `sa_fnv1a64` (FNV-1a-64 over the u32-LE bytes, the hash the full-size oracle comparisons of
    tools/full_parity.py record) is compared with the committed value when the run is the pinned configuration."""
    import torch
    from kiss_amd import sorter
    rep = ctx.verify_sa_dev(S.data_ptr(), n, SA.data_ptr(), k)
    out = {"verified": bool(rep["ok"]), "verify": {kk: rep[kk] for kk in ("exact", "sa0_ok", "out_of_range", "duplicates",
                                                                         "order_violations", "tied_pairs")},
           "sa_digest": "%016x" % rep["digest"]}
    out["verify"]["ms"] = rep["ms"]
    out["verify"]["what"] = ("k >= n: proof that SA is THE suffix array" if rep["exact"] else
                             "SA[0] = n, permutation, substr(SA[i-1], k) <= substr(SA[i], k) for all i (tests/kiss.cpp:26-28)")
    key = None
    if not iid and not harsh and n == CHM13_N:
        key = "chm13size_seed%d_%s" % (seed, "exact" if k >= n else "k%d" % k)
    pins = {}
    if os.path.exists(FULL_SIZE_PINS):
        with open(FULL_SIZE_PINS) as f:
            pins = json.load(f)
    pin = pins.get(key) if key else None
    if with_fnv:
        t0 = time.perf_counter()
        chunk = 64 << 20  # u32 entries per piece
        buf = torch.empty(chunk, dtype=torch.int32, pin_memory=True)
        h = sorter.FNV1A64_SEED
        for a in range(0, n + 1, chunk):
            b = min(n + 1, a + chunk)
            buf[:b - a].copy_(SA[a:b])
            h = sorter.fnv1a64(buf[:b - a].numpy(), h)
        out["sa_fnv1a64"] = "%016x" % h
        out["verify"]["fnv_seconds"] = time.perf_counter() - t0
    if pin:
        out["sa_pinned"] = {"key": key, "digest": pin.get("digest"), "sa_fnv1a64": pin.get("sa_fnv1a64"),
                            "source": pin.get("source")}
        ok = pin.get("digest") in (None, out["sa_digest"])
        if with_fnv and pin.get("sa_fnv1a64"):
            ok = ok and pin["sa_fnv1a64"] == out["sa_fnv1a64"]
        out["sa_matches_pinned_hash"] = bool(ok)
        out["verified"] = out["verified"] and bool(ok)
    return out


def end_to_end_leg(ctx, S, n, k, algo, reps=2):
    """The reference's own timed region (command/suffix_sort.hpp:57-61): S in HOST memory -> SA in HOST memory, through
    kiss_hip_ctx_suffix_sort_dna_u32.  Two host-memory kinds: page-locked (what a host that cooperates allocates; the
    PCIe-limited number) and pageable (what kiss::vector / numpy hand over: 8 copy threads through bounce buffers)."""
    import torch
    res = {"region": "host S (1 byte per base) -> host SA (u32), command/suffix_sort.hpp:57-61",
           "bytes": n + 4 * (n + 1)}
    # PCIe rates on this box, one big page-locked copy each way
    probe = min(n, 1 << 30)
    hp = torch.empty(probe, dtype=torch.uint8, pin_memory=True)
    dp = torch.empty(probe, dtype=torch.uint8, device=S.device)
    for direction in ("h2d", "d2h"):
        best = 0.0
        for _ in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            if direction == "h2d":
                dp.copy_(hp, non_blocking=True)
            else:
                hp.copy_(dp, non_blocking=True)
            torch.cuda.synchronize()
            best = max(best, probe / (time.perf_counter() - t0) / 1e9)
        res["pcie_%s_GBps" % direction] = best
    del hp, dp
    for kind in ("pinned", "pageable"):
        if kind == "pinned":
            S_h = torch.empty(n, dtype=torch.uint8, pin_memory=True)
            SA_h = torch.empty(n + 1, dtype=torch.int32, pin_memory=True)
        else:
            S_h = torch.empty(n, dtype=torch.uint8)
            SA_h = torch.empty(n + 1, dtype=torch.int32)
        S_h.copy_(S)
        torch.cuda.synchronize()
        S_np, SA_np = S_h.numpy(), SA_h.numpy().view(np.uint32)
        best = None
        for r in range(reps + 1):  # the first call allocates the ctx-owned device copies (and faults SA_h in): not timed
            t0 = time.perf_counter()
            ctx.suffix_sort_host(S_np, SA_np, k=k, algo=algo)
            dt = time.perf_counter() - t0
            st = ctx.stats()
            cur = {"ms": 1e3 * dt, "bases_per_s": n / dt, "h2d_ms": st["ms_h2d"], "device_ms": st["ms_total"],
                   "d2h_ms": st["ms_d2h"]}
            if r and (best is None or cur["ms"] < best["ms"]):
                best = cur
        res[kind] = best
        del S_np, SA_np, S_h, SA_h
    floor = 1e3 * (n / (res["pcie_h2d_GBps"] * 1e9) + 4.0 * (n + 1) / (res["pcie_d2h_GBps"] * 1e9)) + res["pinned"]["device_ms"]
    res["floor_ms"] = floor
    res["floor_note"] = "bytes / measured PCIe rate per direction + device time, no overlap"
    res["pinned_over_floor"] = res["pinned"]["ms"] / floor
    return res


def fm_query_leg(device, Q=1_000_000, L=32, steps=5, n=48_800_648):
    """BASELINE.json configs[2] (second metric): FM-index queries/s, batched get_range + get_offsets of 1 M x 32-base
    patterns on a dm-sized index, index and patterns resident in HBM; parity of ranges / hit totals / checksum against
    the oracle on ALL patterns (single thread, like the reference's loop, fmindex_query.hpp:79-95)."""
    import torch
    import kiss_amd.fm_index as fm
    from tests import oracle_binding
    S = gen_text_device(n, 1, device)
    S_host = S.cpu().numpy()
    del S
    f = fm.FMIndex(device=device.index or 0).build(S_host)  # (first build: workspace + index allocations)
    # BASELINE.json configs[2] names fmindex_build as well (reference fmindex_build.hpp:27-34: the index of the whole text,
    # sorted with k = 32): a second build of the same index, host text -> index resident in HBM, timed
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    f.build(S_host)
    torch.cuda.synchronize()
    build_s = time.perf_counter() - t0
    build_sort_ms = f._ctx.stats()["ms_total"]
    rng = np.random.default_rng(3)
    pos = rng.integers(0, n - L, Q)
    pats = S_host[pos[:, None] + np.arange(L)[None, :]]
    mut = rng.random(Q) < 0.1
    col = rng.integers(0, L, Q)
    pats[mut, col[mut]] = (pats[mut, col[mut]] + 1 + rng.integers(0, 3, int(mut.sum()))) % 4
    pats = np.ascontiguousarray(pats, dtype=np.uint8)
    d_p = torch.from_numpy(pats).to(device)
    f._context(max(f.N, 4 * Q)).set_profiling(True)
    r = f.query_batch(None, want_offsets=False, d_patterns=d_p)
    torch.cuda.synchronize()
    st0 = f._ctx.stats()
    kms0 = st0["kernels"]["fm_query"]["ms"]
    t0 = time.perf_counter()
    for _ in range(steps):
        f.query_batch(None, want_offsets=False, d_patterns=d_p, keep_on_device=True)  # ranges left in HBM, like the SA
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    st1 = f._ctx.stats()
    kernel_s = 1e-3 * (st1["kernels"]["fm_query"]["ms"] - kms0) / steps
    range_ms = (st1["ms_fm_range"] - st0["ms_fm_range"]) / steps
    locate_ms = (st1["ms_fm_locate"] - st0["ms_fm_locate"]) / steps
    hits = r["total_hits"]
    index_bytes = sum(int(getattr(f, a).numel() * getattr(f, a).element_size()) for a in ("bwt", "occ1", "occ2", "sa", "b", "b_occ"))
    own = 48.0 * L * Q + 8.0 * hits          # this formulation: 2 LF steps of 24 B per character; range-wise locate
    survey = 48.0 * L * Q + 64.0 * hits      # 1 536 B + 64 B per hit for a 32-base pattern
    IC_GATHER_GBS = 8600.0  # MI355X_MICROARCH.md "Indexed rows": 38 MB table, uniformly random rows (Infinity Cache)
    range_bytes = 2.0 * L * 32.0 * Q  # backward search: 2 L occ per pattern, ONE 32-byte interleaved block each
    out = {"metric": "FM-index queries/sec (batched get_range + get_offsets, %d-base patterns)" % L,
           "value": Q * steps / el, "unit": "queries/s", "queries": Q, "steps": steps, "ms_per_step": 1e3 * el / steps,
           "kernel_ms_per_step": 1e3 * kernel_s, "range_kernel_ms": range_ms, "locate_kernel_ms": locate_ms, "hits": hits, "checksum": r["checksum"], "index_n": n,
           "index_bytes": index_bytes,
           "roofline": {"bound": "infinity_cache", "peak": IC_GATHER_GBS, "unit": "GB/s",
                        "peak_source": "MI355X_MICROARCH.md, gather of uniformly random rows from a 38 MB table",
                        "note": "the %.0f MB index is resident in the 256 MiB Infinity Cache: nothing here is an HBM "
                                "rate; every LF step is a dependent random 64-byte-sector read" % (index_bytes / 1e6),
                        "achieved_own_model": own / kernel_s / 1e9, "frac_own_model": own / kernel_s / 1e9 / IC_GATHER_GBS,
                        "own_model": "48 L per pattern + 8 B per hit",
                        "range_kernel": {"bytes": range_bytes, "model": "2 L occ per pattern x one 32-byte block (k_fm_range)",
                                         "achieved": range_bytes / max(1e-9, 1e-3 * range_ms) / 1e9,
                                         "frac": range_bytes / max(1e-9, 1e-3 * range_ms) / 1e9 / IC_GATHER_GBS},
                        "locate_kernel": {"bytes": 8.0 * hits, "model": "4 B sampled-SA read + 4 B offset write per hit",
                                          "achieved": 8.0 * hits / max(1e-9, 1e-3 * locate_ms) / 1e9,
                                          "frac": 8.0 * hits / max(1e-9, 1e-3 * locate_ms) / 1e9 / IC_GATHER_GBS},
                        "achieved_survey_8d_model": survey / kernel_s / 1e9,
                        "frac_survey_8d_model": survey / kernel_s / 1e9 / IC_GATHER_GBS,
                        "survey_8d_model": "48 L per pattern + 64 B per hit"}}
    orc = oracle_binding.load()
    t0 = time.perf_counter()
    ref = orc.fm_build(S_host, orc.suffix_sort(S_host, 32))
    t1 = time.perf_counter()
    rr = ref.query_batch(pats, want_offsets=False)
    dt = time.perf_counter() - t1
    out["cpu_baseline"] = {"value": Q / dt, "unit": "queries/s", "cores": 1, "kind": "port",
                           "sample": "all %d patterns, single thread like the reference loop (fmindex_query.hpp:79-95), "
                                     "%.1f s (+ %.1f s oracle index build)" % (Q, dt, t1 - t0)}
    from tests.fmi_layout import canonical
    fmi = f.to_bytes()
    out["fm_build"] = {"config": "BASELINE.json configs[2]: fmindex_build of the dm-size text (k = 32 order, SA_INTV = 4), 1 GPU",
                       "ms": 1e3 * build_s, "value": n / build_s, "unit": "bases/s", "suffix_sort_device_ms": build_sort_ms,
                       "region": "host text (1 byte per base) -> index arrays resident in HBM: upload, k = 32 suffix sort, "
                                 "bwt / occ / sampled-SA kernels",
                       "fmi_bytes": len(fmi),
                       "fmi_equal_to_oracle": bool(canonical(fmi) == canonical(ref.serialize())),
                       "fmi_note": "byte for byte the oracle's serialisation of the same index (fm_index.hpp:591-615), modulo the "
                                   "reference's uninitialised pad bits (tests/fmi_layout.py)",
                       "cpu_baseline": {"value": n / (t1 - t0), "unit": "bases/s", "cores": orc.num_threads(), "kind": "port",
                                        "sample": "the whole dm-size text: oracle suffix sort (k = 32) + index build, %.1f s" % (t1 - t0)}}
    del fmi
    out["parity_vs_oracle_all_patterns"] = bool(np.array_equal(r["beg"], rr["beg"]) and np.array_equal(r["end"], rr["end"])
                                                and hits == rr["total_hits"] and r["checksum"] == rr["checksum"])
    f.close()
    return out


def cpu_baseline(S_host_sample, k, threads=24, whole=False):
    """Times the CPU path on a bounded sample, on the GPU box's host cores.  Reported baseline, not the optimisation
    target.  Preferred: oracle/_ref/libkiss_ref.so ("reference": the reference's OWN get_lms, PackedDNAString loads,
    put_lms_suffix and induced_sort compiled unmodified from its sources, with its OpenMP block scheduling, at the
    README's 24 threads; the one stage that lives in kiss1_core.hpp -- bucket scatter + per-bucket std::sort, which needs
    spdlog to compile -- is the restated kref_lms_sort of oracle/ref_driver.cpp, OpenMP over buckets like the
    reference).  Fallback where the prebuilt library did not travel: the plain C oracle ("port")."""
    from tests import oracle_binding, ref_binding
    n = S_host_sample.size
    if ref_binding.available() and os.path.exists(ref_binding.LIB):
        ref = ref_binding.load()
        T = max(1, min(threads, ref.max_threads()))
        t0 = time.time()
        ref.suffix_sort(S_host_sample, k, T=T)
        dt = time.time() - t0
        return {"value": n / dt, "unit": "bases/s", "cores": T, "kind": "reference",
                "sample": "%s %d bases of the same synthetic text, k=%d, %.1f s; oracle/_ref: reference get_lms + "
                          "put_lms_suffix + induced_sort compiled unmodified, LMS sort (kiss1_core.hpp:41-144, needs "
                          "spdlog) restated in oracle/ref_driver.cpp" % ("the WHOLE text, all" if whole else "first", n, k, dt)}
    orc = oracle_binding.load()
    t0 = time.time()
    orc.suffix_sort(S_host_sample, k)
    dt = time.time() - t0
    return {"value": n / dt, "unit": "bases/s", "cores": orc.num_threads(), "kind": "port",
            "sample": "%s %d bases of the same synthetic text, k=%d, %.1f s (oracle/_ref not present)" % (
                "the WHOLE text, all" if whole else "first", n, k, dt)}


EXIT_TOO_FEW_DEVICES = 3  # a rank found fewer visible GPUs than --gpus: never retried, never a silent 1-GPU number
EXIT_VERIFY_FAILED = 4    # the last suffix array failed the device-side check or differs from the pinned hash
DM_N = 48_800_648         # drosophia_chr1_2 (reference README.md:88): BASELINE.json configs[0] / configs[2]


def free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def launch_ranks(n_ranks, argv, child_cmd=None, timeout=None):
    """Starts n_ranks fresh processes of this script (or `child_cmd`, a test stub), one per LOCAL_RANK, waits for all,
    and returns (worst exit status, rank 0's stdout).  The parent never imports torch and never touches HIP.  As soon
    as one rank exits non-zero the others are terminated (they would otherwise wait in a collective); ranks that are all
    alive but stuck (a collective that never completes) are terminated after `timeout` seconds -> status 124."""
    import subprocess
    port = free_port()
    cmd = list(child_cmd) if child_cmd else [sys.executable, os.path.abspath(__file__)]
    procs = []
    for r in range(n_ranks):
        env = dict(os.environ)
        env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(n_ranks), "LOCAL_WORLD_SIZE": str(n_ranks),
                    "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)})
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen(cmd + list(argv), env=env, stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL,
                                      text=True))
    t0 = time.time()
    worst = 0
    live = set(range(n_ranks))
    out0 = ""
    while live:
        for r in sorted(live):
            rc = procs[r].poll()
            if rc is None:
                continue
            live.discard(r)
            if rc != 0:
                print("[bench] rank %d exited with status %d" % (r, rc), file=sys.stderr, flush=True)
                worst = rc if worst == 0 or rc == EXIT_TOO_FEW_DEVICES else worst
        if (worst != 0 or (timeout and time.time() - t0 > timeout)) and live:
            if worst == 0:
                print("[bench] ranks %s still running after %.0f s: terminating" % (sorted(live), timeout),
                      file=sys.stderr, flush=True)
                worst = 124
            time.sleep(2.0)  # let the other ranks report their own error first
            for r in live:
                procs[r].terminate()  # exact PIDs we started
            for r in live:
                try:
                    procs[r].wait(20)
                except subprocess.TimeoutExpired:
                    procs[r].kill()
            live.clear()
            break
        if live:
            time.sleep(0.05)  # rank 0 prints one line at the very end: read after exit (the pipe buffer is ample)
    if procs[0].stdout is not None:
        try:
            out0 = procs[0].stdout.read()
        except Exception:  # noqa: BLE001
            out0 = ""
    return worst, out0


def parent_main(args, argv, child_cmd=None):
    """--gpus N > 1 without a launcher: be the launcher."""
    timeout = getattr(args, "rank_timeout", None) or None
    rc, out0 = launch_ranks(args.gpus, argv, child_cmd=child_cmd, timeout=timeout)
    if rc in (0, EXIT_VERIFY_FAILED):  # a line whose SA failed its check is relayed as it is (value null) with the status
        sys.stdout.write(out0)
        sys.stdout.flush()
        return rc
    if rc not in (EXIT_TOO_FEW_DEVICES, 124) and args.mode == "sharded" and not args.no_fallback:
        # (never after a timeout: ranks stuck in a collective say nothing a second set of ranks could use)
        reason = "sharded run failed (a rank exited with status %d, see stderr)" % rc
        print("[bench] %s; starting a fresh set of ranks with --mode replicas" % reason, file=sys.stderr, flush=True)
        rc2, out0 = launch_ranks(args.gpus, list(argv) + ["--mode", "replicas", "--sharded-error", reason],
                                 child_cmd=child_cmd, timeout=timeout)
        if rc2 in (0, EXIT_VERIFY_FAILED):
            sys.stdout.write(out0)
            sys.stdout.flush()
        return rc2
    return rc


def roofline_of(agg, prof_agg, steps, prof_steps, profile_all):
    """roofline object of the dominant kernel class: ALGORITHMIC bytes per launch / average launch duration from the HIP
    events of the library.  agg: {class: {ms, launches, items}} of the timed region (only the classes timed inside it),
    prof_agg: the full breakdown (the same dict when every class is timed inside the region)."""
    if not prof_agg or not any(v["launches"] for v in prof_agg.values()):
        return None
    name = max(prof_agg.items(), key=lambda kv: kv[1]["ms"])[0]
    in_timed_region = bool(agg.get(name, {}).get("launches"))
    a = agg[name] if in_timed_region else prof_agg[name]
    a_steps = steps if in_timed_region else prof_steps
    if not a["launches"] or a["ms"] <= 0:
        return None
    bytes_per_launch = ALGO_BYTES.get(name, 0.0) * a["items"] / a["launches"]
    avg_s = 1e-3 * a["ms"] / a["launches"]
    achieved = bytes_per_launch / avg_s / 1e9
    return {"bound": "hbm", "kernel": name, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS,
            "traffic": (bytes_per_launch * MEASURED_TRAFFIC_PER_ALGO_BYTE[name]
                        if name in MEASURED_TRAFFIC_PER_ALGO_BYTE else None),
            "traffic_note": "bytes per launch = algorithmic bytes x the PMC-measured traffic ratio of this "
                            "kernel (separate rocprofv3 --pmc runs, profiles/r03_pmc_*.csv)",
            "avg_launch_us": 1e6 * avg_s, "launches_per_step": a["launches"] / a_steps,
            "measured_in": ("timed region (HIP events around this class only)" if in_timed_region and not profile_all else
                            "timed region (HIP events around every class)" if in_timed_region else
                            "profiled steps after the timed region (not the class timed inside it)"),
            "algorithmic_bytes_per_item": ALGO_BYTES.get(name, 0.0),
            "kernel_ms_per_step": {kn: kv["ms"] / prof_steps for kn, kv in prof_agg.items() if kv["launches"]},
            "kernel_ms_per_step_from": ("timed region" if prof_agg is agg else
                                        "%d extra steps with every class timed" % prof_steps)}


def assemble_line(args, world, sharded, n, k, algo, elapsed, agg, prof_agg, prof_steps, stage, last_stats, workspace_bytes,
                  phase_ms=None, sharded_error=None, data="synthetic", text_desc=None):
    """The result line of rank 0 from what the timed region measured -- the same fields at every N (tests/
    test_bench_launcher.py checks the N = 2 form on CPU).  `value` is the whole-job aggregate."""
    total_bases = float(n) * args.steps * (1 if sharded else world)
    if text_desc is None:
        text_desc = "i.i.d." if args.iid else ("genome-like synthetic, HARSH satellite profile" if args.harsh
                                               else "genome-like synthetic")
    # one text sharded over the GPUs is the default form of this bench: the per-GPU work shrinks as N grows ("strong");
    # the N = 1 line of that series says so too.  --mode replicas (one independent text per GPU) is the weak form.
    strong = sharded or (world == 1 and args.mode == "sharded")
    out = {
        "metric": "bases/sec suffix_sort (chm13v2.0-size %s, k=%d)" % ("text" if data == "file" else "synthetic", k),
        "value": total_bases / elapsed,
        "unit": "bases/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True,
        "scaling": "strong" if strong else "weak",
        "vs_baseline": None,
        "dtype": "u8",
        "data": data,
        "config": {
            "workload": "suffix_sort %s n=%d k=%d (%sBASELINE.json configs[%d]); %s; text resident in HBM, SA left in HBM"
                        % (text_desc, n, k, "" if data == "file" else "stand-in for chm13v2.0.fa, ", 4 if world > 1 and sharded else 1,
                           "PREFIX_DOUBLING (bounded phase + rank doubling)" if algo else "PARALLEL_SORTING"),
            "n": n, "k": k, "seed": args.seed,
            "parallelism": ("single GPU" if world == 1 and not sharded else
                            ("one text, LMS sort sharded by key range over %d GPU%s (RCCL all-to-all of the LMS "
                             "list, gather of the sorted pieces, induction on rank 0)" % (world, "" if world == 1 else "s"))
                            if sharded else "1 text per GPU (independent replicas)"),
            "lms": last_stats["m"], "lms_rounds": last_stats["lms_rounds"],
            "tied_after_round0_item_rounds": max(0, last_stats["sort_item_rounds"] - last_stats["m"]),  # (this rank's share)
            "big_segment_item_rounds": last_stats["big_item_rounds"],
            "workspace_bytes": workspace_bytes,
            "induce_passes": last_stats["induce_passes"],
            "stage_ms_per_step": {s: v / args.steps for s, v in stage.items()},
        },
    }
    if sharded_error:
        out["config"]["sharded_error"] = sharded_error
    if phase_ms:
        out["config"]["sharded_phase_ms_rank0"] = {kk: v / args.steps for kk, v in phase_ms.items()}
    # roofline of the dominant kernel class (live HIP-event timing inside the library); in sharded runs the launches are
    # rank 0's (its key range of the sort + the induction)
    out["roofline"] = None if args.no_profile else roofline_of(agg, prof_agg, args.steps, prof_steps, args.profile_all)
    # whole-path algorithmic bytes: 0.25 n + 20 m + 16 (n+1) + 2 n
    m = last_stats["m"]
    path_bytes = 0.25 * n + 20.0 * m + 16.0 * (n + 1) + 2.0 * n
    dev_s = 1e-3 * stage["total"] / args.steps
    if dev_s <= 0:  # sharded runs: per-stage device times are not collected, use the step wall time
        dev_s = elapsed / args.steps
    out["path_roofline"] = {"algorithmic_bytes": path_bytes, "device_ms": 1e3 * dev_s,
                            "achieved_GBps": path_bytes / dev_s / 1e9, "frac": path_bytes / dev_s / 1e9 / HBM_PEAK_GBS}
    return out


def agree_on_rehearsal(dist, torch, device, local_ok, local_msg):
    """every rank contributes whether ITS part of the rehearsal went well; all ranks return the same verdict: None when
    all did, else a reason (the rank's own when it has one)"""
    if dist.get_backend() == "gloo":  # (the CPU / one-GPU tests)
        device = torch.device("cpu")
    flag = torch.tensor([1 if local_ok else 0], dtype=torch.int32, device=device)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    if int(flag.item()) == 1:
        return None
    return local_msg or "the rehearsal failed on another rank"


def rehearse_sharded(torch, dist, kiss_amd, S, n, k, rank, local_rank, device):
    """The sharded pipeline has never met a second GPU.  Before anything is timed, the ranks sort the first
    8 M bases of the text through it -- every collective and every stage entry point, small messages -- and rank 0
    checks the result on the device.  The ranks agree on the outcome; a failure that all ranks survive (a wrong suffix
    array, an error code, an exception raised on every rank) turns the run into --mode replicas, said so in the line
    (config.sharded_error).  A collective that never completes is not survivable: that is what --rank-timeout is for."""
    from kiss_amd import multi_gpu
    ns = int(min(n, 8_000_000))
    ok, msg = True, None
    try:
        Ss = S[:ns]
        c = kiss_amd.Context(max_n=ns, device=local_rank)
        try:
            sa = torch.empty(ns + 1, dtype=torch.int32, device=device) if rank == 0 else None
            multi_gpu.sharded_suffix_sort(multi_gpu.GpuBackend(c, Ss, k), ns, SA=sa)
            torch.cuda.synchronize()
            if rank == 0:
                rep = c.verify_sa_dev(Ss.data_ptr(), ns, sa.data_ptr(), k)
                if not rep["ok"]:
                    ok, msg = False, "the rehearsal's suffix array fails the device-side check: %s" % json.dumps(rep)
        finally:
            c.close()
    except Exception as e:  # noqa: BLE001 -- whatever it is, the ranks have to agree on what happens next
        ok, msg = False, "rank %d: %s: %s" % (rank, type(e).__name__, e)
    return agree_on_rehearsal(dist, torch, device, ok, msg)


# 1-GPU phase times of the headline configuration (ms; DESIGN.md 4, round 4 records) and the xGMI rate DESIGN.md 7 prices
# the two data-path transfers at: the inputs of the expected strong-scaling curve
MODEL_1GPU_MS = {"pack": 0.65, "classify": 4.0, "sort": 52.5, "induce": 17.9, "histogram": 3.4, "partition": 5.6}
MODEL_LINK_GBPS = 153.0
MODEL_LMS_FRACTION = 0.2906  # LMS suffixes per base of the synthetic chm13-size text (905 939 973 / 3 117 292 070)


def scaling_model(G, n, m=None):
    """DESIGN.md 7's expected curve for G GPUs of one node, as numbers a measured line can be checked against: the sharded
    phases divide by G, the all-to-all moves 12 B per LMS suffix ((G-1)/G of every rank's share over its G-1 links at
    once), the gather moves 8 B per LMS suffix into rank 0 over its G-1 links, pack and induction stay on one GPU.  The
    phase times are those of the chm13-size text scaled by n; NO multi-GPU run has confirmed any of it yet."""
    m = float(m) if m else MODEL_LMS_FRACTION * n
    scale = n / float(CHM13_N)
    t = {kk: v * scale for kk, v in MODEL_1GPU_MS.items()}
    link = MODEL_LINK_GBPS * 1e9
    if G > 1:
        exchange_bytes_per_rank = 12.0 * m / G * (G - 1) / G
        exchange_ms = 1e3 * exchange_bytes_per_rank / ((G - 1) * link)
        gather_bytes = 8.0 * m * (G - 1) / G
        gather_ms = 1e3 * gather_bytes / ((G - 1) * link)
        sharded_ms = (t["classify"] + t["histogram"] + t["partition"] + t["sort"]) / G + 0.3
    else:
        exchange_bytes_per_rank = gather_bytes = exchange_ms = gather_ms = 0.0
        sharded_ms = t["classify"] + t["sort"] + 0.3
    serial_ms = t["pack"] + t["induce"]
    return {"gpus": G, "expected_ms": sharded_ms + exchange_ms + gather_ms + serial_ms,
            "sharded_phases_ms": sharded_ms, "exchange_ms": exchange_ms, "gather_ms": gather_ms, "serial_on_rank0_ms": serial_ms,
            "exchange_bytes_per_rank": exchange_bytes_per_rank, "gather_bytes_into_rank0": gather_bytes,
            "link_GBps_assumed": MODEL_LINK_GBPS, "one_gpu_phase_ms": t,
            "ceiling_speedup": (t["pack"] + t["classify"] + t["sort"] + t["induce"]) / serial_ms,
            "status": "model only: no run on more than one GPU has been available to check it (DESIGN.md 7)"}


def exact_order_leg(ctx, S, SA, n, stream, steps=3):
    """BASELINE.json configs[3]: the same text, k = -1 (unbounded) through PREFIX_DOUBLING -- the bounded phase + rank
    doubling over the tied suffixes; the result is THE suffix array, proven on the device after the timed steps."""
    import torch
    K = 0xFFFFFFFF
    ctx.set_profiling(False)
    ctx.suffix_sort_dev(S.data_ptr(), n, SA.data_ptr(), k=K, algo=1, stream=stream)  # first-use allocations (inverse-SA pairs)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        ctx.suffix_sort_dev(S.data_ptr(), n, SA.data_ptr(), k=K, algo=1, stream=stream)
    torch.cuda.synchronize()
    el = (time.perf_counter() - t0) / steps
    st = ctx.stats()
    rep = ctx.verify_sa_dev(S.data_ptr(), n, SA.data_ptr(), K)
    return {"config": "BASELINE.json configs[3]: suffix_sort k=-1 (unbounded), PREFIX_DOUBLING, same text, 1 GPU",
            "ms_per_step": 1e3 * el, "value": n / el, "unit": "bases/s", "steps": steps,
            "device_ms": st["ms_total"], "bounded_phase_order": st["refine_depth"], "doubling_ms": st["ms_refine"],
            "tied_after_bounded_phase": st["refine_items"], "doubling_rounds": st["doubling_rounds"],
            # 1: rank doubling over the LMS suffixes before the induction (tied = tied LMS suffixes), 2: over the whole
            # suffix array after it (the fall-back; tied = tied suffixes)
            "doubling_over": {0: "nothing", 1: "lms_suffixes", 2: "suffix_array"}.get(st["refine_form"], "?"),
            "verified": bool(rep["ok"]), "verify": {"exact": rep["exact"], "order_violations": rep["order_violations"],
                                                    "duplicates": rep["duplicates"], "ms": rep["ms"]},
            "sa_digest": "%016x" % rep["digest"], "workspace_bytes": ctx.workspace_bytes()}


def sensitivity_leg(ctx, SA, n, k, seed, device, stream, steps=2):
    """The same configuration on the `--harsh` text (6 % of the bases in tandem arrays of up to 30 Mb, a third of them
    alpha-satellite-like higher-order repeats: the closest stand-in this generator has for chm13's centromeres), beside the
    headline text's 3 %: what the satellite content costs.  Verified on the device like the headline run."""
    import torch
    S = gen_text_device(n, seed, device, harsh=True)
    ctx.set_profiling(False)
    ctx.suffix_sort_dev(S.data_ptr(), n, SA.data_ptr(), k=k, stream=stream)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        ctx.suffix_sort_dev(S.data_ptr(), n, SA.data_ptr(), k=k, stream=stream)
    torch.cuda.synchronize()
    el = (time.perf_counter() - t0) / steps
    st = ctx.stats()
    rep = ctx.verify_sa_dev(S.data_ptr(), n, SA.data_ptr(), k)
    out = {"text": "--harsh: tandem arrays 6 % of the text, up to 30 Mb each, a third as 12 x 171-base higher-order repeats",
           "ms_per_step": 1e3 * el, "value": n / el, "unit": "bases/s", "steps": steps, "lms_rounds": st["lms_rounds"],
           "stage_ms": {"lms_sort": st["ms_lms_sort"], "induce": st["ms_induce"], "classify": st["ms_classify"]},
           "verified": bool(rep["ok"]), "verify": {"order_violations": rep["order_violations"], "duplicates": rep["duplicates"],
                                                   "tied_pairs": rep["tied_pairs"]},
           "sa_digest": "%016x" % rep["digest"], "workspace_bytes": ctx.workspace_bytes()}
    del S
    return out


def dm_leg(ctx_factory, device, k=256, threads=24):
    """BASELINE.json configs[0] (`suffix_sort example/drosophia_chr1_2.fa -k 256 -t 24`, reference README.md:85-88:
    0.4809 s = 101.5 Mbases/s on its authors' host): the dm-size stand-in C1 (n = 48 800 648, seed 1)
    through the reference-code pipeline of oracle/_ref at the README's 24 threads on THIS box's host cores (whole
    text, best of 3), beside the HIP path on the same text -- the two suffix arrays compared bit for bit."""
    import torch
    from tests import ref_binding
    S = gen_text_device(DM_N, 1, device)
    S_host = S.cpu().numpy()
    out = {"config": "BASELINE.json configs[0]: suffix_sort dm-size (n=%d) k=%d -t %d on the CPU reference" % (DM_N, k, threads),
           "published": {"seconds": 0.4809, "bases_per_s": DM_N / 0.4809, "threads": 24, "hardware": "unstated",
                         "source": "reference README.md:87-88"}}
    sa_ref = None
    if ref_binding.available() and os.path.exists(ref_binding.LIB):
        ref = ref_binding.load()
        T = max(1, min(threads, ref.max_threads()))
        best = None
        for _ in range(3):
            t0 = time.time()
            sa_ref = ref.suffix_sort(S_host, k, T=T)
            dt = time.time() - t0
            best = dt if best is None or dt < best else best
        out["cpu_baseline_dm"] = {"value": DM_N / best, "unit": "bases/s", "cores": T, "kind": "reference", "seconds": best,
                                  "sample": "the whole dm-size text (n=%d, seed 1), k=%d, best of 3; oracle/_ref: reference "
                                            "get_lms + put_lms_suffix + induced_sort compiled unmodified, LMS sort restated "
                                            "(oracle/ref_driver.cpp)" % (DM_N, k)}
    else:
        out["cpu_baseline_dm"] = None
    ctx = ctx_factory(DM_N)
    SA = torch.empty(DM_N + 1, dtype=torch.int32, device=device)
    stream = torch.cuda.current_stream().cuda_stream
    ctx.suffix_sort_dev(S.data_ptr(), DM_N, SA.data_ptr(), k=k, stream=stream)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 5
    for _ in range(reps):
        ctx.suffix_sort_dev(S.data_ptr(), DM_N, SA.data_ptr(), k=k, stream=stream)
    torch.cuda.synchronize()
    el = (time.perf_counter() - t0) / reps
    out["hip"] = {"ms_per_step": 1e3 * el, "value": DM_N / el, "unit": "bases/s"}
    if sa_ref is not None:
        out["hip"]["sa_equal_to_reference_pipeline"] = bool(np.array_equal(SA.cpu().numpy().view(np.uint32), sa_ref))
    ctx.close()
    return out


class _DevArray:
    def __init__(self, ptr, count):
        self.__cuda_array_interface__ = {"shape": (int(count),), "typestr": "|u1", "data": (int(ptr), False), "version": 2}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--text-len", dest="n", type=int, default=CHM13_N, help="text length (default: chm13v2.0 size)")
    ap.add_argument("--fasta", default=None,
                    help="sort this FASTA / plain-text file (e.g. a real chm13v2.0.fa) instead of the synthetic text: read, "
                         "uploaded and parsed on the device by kiss_hip_ctx_load_text_file; the line says \"data\": \"file\"")
    ap.add_argument("--k", type=int, default=256)
    ap.add_argument("--algo", choices=["parallel_sorting", "prefix_doubling"], default="parallel_sorting",
                    help="prefix_doubling: exact order (k is ignored), bounded phase + rank doubling")
    ap.add_argument("--seed", type=int, default=2)
    ap.add_argument("--iid", action="store_true", help="i.i.d. text instead of the genome-like generator")
    ap.add_argument("--harsh", action="store_true",
                    help="sensitivity run: 6 %% of the text in tandem arrays up to 30 Mb incl. higher-order repeats")
    ap.add_argument("--cpu-sample", type=int, default=CHM13_N,
                    help="bases of the CPU baseline sample (0 = skip); 1.5e9 bases = ~15 s of oracle/_ref time at 24 threads")
    ap.add_argument("--no-profile", action="store_true", help="do not time kernel classes with HIP events")
    ap.add_argument("--sharded-timings", action="store_true",
                    help="sharded mode: wall-clock ms per phase of rank 0 (adds a synchronisation per phase: a diagnostic, "
                         "the headline of such a run is slower than a plain one)")
    ap.add_argument("--profile-all", action="store_true",
                    help="time EVERY kernel class inside the timed region (two HIP events per launch, ~560 launches per "
                         "sort: +3 ms per step); default: only the dominant class there, the others in --profile-steps "
                         "extra steps after it")
    ap.add_argument("--profile-steps", type=int, default=2,
                    help="extra, untimed steps with every kernel class timed (the per-class breakdown of the line)")
    ap.add_argument("--mode", choices=["sharded", "replicas"], default="sharded",
                    help="N > 1: 'sharded' = ONE text, LMS sort sharded by key range over the ranks with an RCCL "
                         "all-to-all (strong scaling); 'replicas' = one independent text per rank (weak scaling)")
    ap.add_argument("--force-sharded", action="store_true",
                    help="run the sharded pipeline (RCCL collectives included) even with a single rank (test aid)")
    ap.add_argument("--multi-abi", default=None, metavar="DEVICES",
                    help="single process: time kiss_hip_multi_suffix_sort_dna_u32_dev over this comma-separated device list "
                         "(one process driving several devices, peer copies; '0,0' = two shares on one GPU) instead of the "
                         "single-device entry")
    ap.add_argument("--no-verify", action="store_true", help="skip the device-side check of the last SA")
    ap.add_argument("--no-fnv", action="store_true", help="skip the host-side FNV-1a-64 of the last SA (~15-20 s at chm13 size)")
    ap.add_argument("--no-e2e", action="store_true", help="skip the host S -> host SA leg (single GPU only)")
    ap.add_argument("--no-fm", action="store_true", help="skip the FM-index queries/s leg (single GPU only)")
    ap.add_argument("--no-exact", action="store_true", help="skip the exact-order leg (BASELINE configs[3]; single GPU only)")
    ap.add_argument("--no-dm", action="store_true", help="skip the dm-size leg (BASELINE configs[0]; single GPU only)")
    ap.add_argument("--no-sensitivity", action="store_true", help="skip the --harsh text beside the headline one (single GPU only)")
    ap.add_argument("--exact-steps", type=int, default=3)
    ap.add_argument("--fm-text-len", type=int, default=DM_N, help="text length of the FM-index leg (default: dm size)")
    ap.add_argument("--fm-queries", type=int, default=1_000_000)
    ap.add_argument("--no-rehearsal", action="store_true",
                    help="sharded runs on more than one GPU: skip the small rehearsal sort (and the agreed fall-back to "
                         "--mode replicas when it fails) in front of the measurement")
    ap.add_argument("--no-fallback", action="store_true",
                    help="self-launching parent only: do not start fresh --mode replicas ranks after a sharded failure")
    ap.add_argument("--rank-timeout", type=float, default=1500.0,
                    help="self-launching parent only: seconds after which ranks that are all alive but not finished (stuck in "
                         "a collective) are terminated -> exit status 124, no retry (0 = wait for ever)")
    ap.add_argument("--sharded-error", default=None, help=argparse.SUPPRESS)  # set by the parent on its fallback run
    args = ap.parse_args()
    if args.gpus < 1:
        ap.error("--gpus must be >= 1")
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and args.gpus > 1:
        # no launcher above us: start the ranks ourselves -- before torch is imported or HIP is touched
        sys.exit(parent_main(args, sys.argv[1:]))
    if env_world is not None and int(env_world) != args.gpus:
        print("[bench] --gpus %d but WORLD_SIZE=%s: refusing to report a number for a different job size"
              % (args.gpus, env_world), file=sys.stderr, flush=True)
        sys.exit(2)

    # stdout carries exactly ONE line, the JSON result: RCCL (version banner, NCCL_DEBUG output) and other libraries
    # print to file descriptor 1 as well, so everything but the result line is sent to stderr
    result_out = os.fdopen(os.dup(1), "w")
    sys.stdout.flush()
    os.dup2(2, 1)

    import torch
    import kiss_amd

    # RANK / LOCAL_RANK only mean something under a multi-process launcher (WORLD_SIZE > 1); a stray RANK in the
    # environment of a single-process run must not turn this process into a silent non-zero rank
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        rank = int(os.environ.get("RANK", "0"))
        local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    else:
        world, rank, local_rank = 1, 0, 0
    visible = torch.cuda.device_count()  # does not initialise the GPU
    # KISS_BENCH_SHARE_GPU=1 (tests): every rank uses GPU 0 and the ranks talk gloo (RCCL refuses two ranks on one device)
    # -- the whole N > 1 code path of this file on a one-GPU box, minus the transport
    share = world > 1 and os.environ.get("KISS_BENCH_SHARE_GPU") == "1"
    if share:
        local_rank = 0
    if not share and (visible < world or local_rank >= visible):
        print("[bench] rank %d: %d GPUs asked for, %d visible: not reporting a number for a smaller job"
              % (rank, world, visible), file=sys.stderr, flush=True)
        sys.exit(EXIT_TOO_FEW_DEVICES)
    if world > 1 or args.force_sharded:
        import torch.distributed as dist
        if world == 1:
            os.environ["RANK"], os.environ["WORLD_SIZE"], os.environ["LOCAL_RANK"] = "0", "1", "0"
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        if share:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    else:
        dist = None
        torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)

    n, k = args.n, args.k
    algo = 1 if args.algo == "prefix_doubling" else 0
    if algo:
        k = 0xFFFFFFFF
    sharded = (world > 1 and args.mode == "sharded") or args.force_sharded
    seed = args.seed if sharded else args.seed + 1000 * rank  # sharded: every rank holds the same text
    ctx = None
    data, text_desc, file_ptr = "synthetic", None, None
    if args.fasta:
        # a real genome when one is present: the file's bases are what every rank sorts; read, uploaded
        # and parsed on the device (fasta.hip) outside the timed region, like the reference's stopwatch (suffix_sort.hpp:57)
        nbytes = os.path.getsize(args.fasta)
        loader = kiss_amd.Context(max_n=max(nbytes, 1), device=local_rank, lms_capacity=max(1 << 20, nbytes // 2048))
        t0 = time.perf_counter()
        file_ptr, n = loader.load_text_file(args.fasta)
        load_s = time.perf_counter() - t0
        loader.close()
        S = torch.as_tensor(_DevArray(file_ptr, n), device=device)
        data, text_desc = "file", "file %s (%d bytes, read + upload + device-side parse %.2f s)" % (
            os.path.basename(args.fasta), nbytes, load_s)
    elif args.iid:
        g = torch.Generator(device=device)
        g.manual_seed(seed)
        S = torch.randint(0, 4, (n,), dtype=torch.uint8, device=device, generator=g)
    else:
        S = gen_text_device(n, seed, device, harsh=args.harsh)
    SA = torch.empty(n + 1, dtype=torch.int32, device=device)  # u32 payload; torch has no uint32 arithmetic needs
    torch.cuda.synchronize()

    sharded_error = args.sharded_error  # set by the self-launching parent on its fresh replicas run, or just below
    if sharded and world > 1 and not args.no_rehearsal:
        err = rehearse_sharded(torch, dist, kiss_amd, S, n, k, rank, local_rank, device)
        if err:
            print("[bench] rank %d: sharded rehearsal failed (%s): every rank measures --mode replicas instead" % (rank, err),
                  file=sys.stderr, flush=True)
            sharded, sharded_error = False, "sharded rehearsal failed, all ranks fell back to replicas: " + err
            if rank > 0 and not args.fasta:  # replicas sort independent texts (the file form: every rank its copy)
                del S
                seed = args.seed + 1000 * rank
                if args.iid:
                    g = torch.Generator(device=device)
                    g.manual_seed(seed)
                    S = torch.randint(0, 4, (n,), dtype=torch.uint8, device=device, generator=g)
                else:
                    S = gen_text_device(n, seed, device, harsh=args.harsh)
                torch.cuda.synchronize()

    multi = None
    if args.multi_abi:
        if world > 1 or sharded:
            ap.error("--multi-abi is the single-process form: not with --gpus > 1 / --force-sharded")
        devs = [int(x) for x in args.multi_abi.split(",")]
        multi = kiss_amd.MultiContext(devs, max_n=n)
        ctx = multi.rank_context(0)  # statistics / profiling / verification go through share 0's context
    else:
        # ranks > 0 of a sharded sort hold about 1/G of the LMS suffixes (their arrays regrow on demand)
        cap = int(0.32 * n / world * 1.25) + 65536 if (sharded and rank > 0) else 0
        ctx = kiss_amd.Context(max_n=n, device=local_rank, lms_capacity=cap)
    # The contract wants the dominant kernel's launch duration from HIP events inside the timed region.  Events around
    # every launch of every class cost ~5 us of stream time each; so the timed region times the dominant class only
    # (DOMINANT_CLASS, checked against the full breakdown below) and the other classes are timed in extra steps.
    if not args.no_profile:
        ctx.set_profiling(True, None if args.profile_all else [DOMINANT_CLASS])
    stream = torch.cuda.current_stream().cuda_stream

    phase_ms = None
    if sharded:
        from kiss_amd import multi_gpu
        backend = multi_gpu.GpuBackend(ctx, S, k)
        phase_ms = {} if args.sharded_timings else None

        def step():
            multi_gpu.sharded_suffix_sort(backend, n, SA=SA if rank == 0 else None, timings=phase_ms)
    elif multi is not None:
        def step():
            multi.suffix_sort_dev(S.data_ptr(), n, SA.data_ptr(), k=k, algo=algo)
    else:
        def step():
            ctx.suffix_sort_dev(S.data_ptr(), n, SA.data_ptr(), k=k, algo=algo, stream=stream)

    # A rank that fails here exits non-zero (the traceback goes to stderr): a process group whose collective has
    # failed is never used again, and there is no in-process retry in another mode.
    for _ in range(args.warmup):
        step()
    if phase_ms is not None:
        phase_ms.clear()  # the warm-up call holds the first-use allocations and the communicator set-up

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    agg = {}
    stage = {"pack": 0.0, "classify": 0.0, "lms_sort": 0.0, "place": 0.0, "induce": 0.0, "total": 0.0}
    last_stats = None
    multi_phase = {}
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
        st = ctx.stats()
        last_stats = st
        for name, v in st["kernels"].items():
            a = agg.setdefault(name, {"ms": 0.0, "launches": 0, "items": 0})
            a["ms"] += v["ms"]
            a["launches"] += v["launches"]
            a["items"] += v["items"]
        for s in stage:
            stage[s] += st["ms_" + s]
        if multi is not None:
            for kk, v in multi.stats().items():
                if kk.startswith("ms_"):
                    multi_phase[kk] = multi_phase.get(kk, 0.0) + v
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=torch.device("cpu") if dist.get_backend() == "gloo" else device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # sharded runs: ONE extra step after the timed region with a device synchronisation at every phase boundary (what
    # --sharded-timings does to every step): rank 0's phase times, the inputs of DESIGN.md 7's expected-curve table, in every
    # N > 1 line -- so that a measured curve can be checked against the model phase by phase
    phase_extra = None
    if sharded and phase_ms is None:
        phase_extra = {}
        multi_gpu.sharded_suffix_sort(backend, n, SA=SA if rank == 0 else None, timings=phase_extra)
        barrier()

    # per-class breakdown: every class timed, outside the timed region
    prof_agg, prof_steps = agg, args.steps
    if not args.no_profile and not args.profile_all and args.profile_steps > 0 and not sharded:
        ctx.set_profiling(True)
        prof_agg, prof_steps = {}, args.profile_steps
        for _ in range(args.profile_steps):
            step()
            for name, v in ctx.stats()["kernels"].items():
                a = prof_agg.setdefault(name, {"ms": 0.0, "launches": 0, "items": 0})
                a["ms"] += v["ms"]
                a["launches"] += v["launches"]
                a["items"] += v["items"]
        ctx.set_profiling(True, [DOMINANT_CLASS])
        barrier()

    status = 0
    if rank == 0:
        if multi is not None:
            last_stats["m"] = multi.stats()["m"] or last_stats["m"]
        out = assemble_line(args, world, sharded, n, k, algo, elapsed, agg, prof_agg, prof_steps, stage, last_stats,
                            ctx.workspace_bytes(), phase_ms=phase_ms, sharded_error=sharded_error, data=data,
                            text_desc=text_desc)
        if share:
            out["config"]["ranks_share_one_gpu"] = ("KISS_BENCH_SHARE_GPU=1: every rank on GPU 0, gloo transport -- a test of this "
                                                    "file's N > 1 code path, not a measurement")
        if phase_extra:
            out["config"]["sharded_phase_ms_rank0"] = dict(phase_extra)
            out["config"]["sharded_phase_ms_from"] = "one extra step after the timed region, phases closed by device synchronisations"
        if sharded:
            out["config"]["scaling_model"] = scaling_model(world, n, last_stats["m"] if world == 1 else None)
        if multi is not None:
            out["config"]["parallelism"] = ("ONE process driving devices %s through kiss_hip_multi_* (LMS sort sharded by key "
                                            "range, peer copies, induction on the first device)" % args.multi_abi)
            out["config"]["multi_phase_ms"] = {kk: v / args.steps for kk, v in multi_phase.items()}
            # the model's expectation for as many DISTINCT devices as shares were asked for (shares of one GPU, the only
            # form a one-GPU box can run, compete for it: they price the mechanics, not the curve)
            out["config"]["scaling_model"] = scaling_model(len(args.multi_abi.split(",")), n)
        if not args.no_verify:
            out.update(verify_leg(ctx, S, SA, n, k, args.seed, args.iid or data == "file", algo,
                                  with_fnv=(world == 1 and not args.no_fnv), harsh=args.harsh))
            if not out["verified"]:
                # a throughput for a wrong suffix array is not a result: the headline is withdrawn, the report stays
                print("[bench] the last suffix array FAILED its check: %s" % json.dumps(
                    {kk: out.get(kk) for kk in ("verify", "sa_digest", "sa_fnv1a64", "sa_pinned", "sa_matches_pinned_hash")}),
                    file=sys.stderr, flush=True)
                out["invalid"] = "the last suffix array failed the device-side check or differs from the pinned hash"
                out["unverified_value"], out["value"] = out["value"], None
                status = EXIT_VERIFY_FAILED
        single = world == 1 and not sharded and multi is None
        if single and not args.no_e2e:
            out["end_to_end"] = end_to_end_leg(ctx, S, n, k, algo)
            ctx.release_io_buffers()
        if single and not args.no_exact and not algo and n >= 4 * 256 + 1024:
            out["exact_order"] = exact_order_leg(ctx, S, SA, n, stream, steps=args.exact_steps)
            if not out["exact_order"]["verified"]:
                status = EXIT_VERIFY_FAILED
        if single and not args.no_sensitivity and not args.harsh and not args.iid and data != "file" and not algo:
            # (after the CPU copy of the headline text is taken below would need S twice: the harsh text replaces nothing)
            out["sensitivity"] = {"harsh": sensitivity_leg(ctx, SA, n, k, args.seed, device, stream)}
            if not out["sensitivity"]["harsh"]["verified"]:
                status = EXIT_VERIFY_FAILED
        # the CPU baseline: rank 0's host cores, at every N (the other ranks wait at the closing barrier)
        if args.cpu_sample > 0:
            ns = min(n, args.cpu_sample)
            sample = S[:ns].cpu().numpy()
            out["cpu_baseline"] = cpu_baseline(sample, k, whole=(ns == n))
            del sample
        else:
            out["cpu_baseline"] = None
        if single and (not args.no_fm or not args.no_dm):
            del S, SA
            ctx.close()
            if file_ptr:
                kiss_amd.Context(max_n=1024, device=local_rank, lms_capacity=1024).free_dev(file_ptr)
                file_ptr = None
            torch.cuda.empty_cache()
            if not args.no_dm:
                dm = dm_leg(lambda nn: kiss_amd.Context(max_n=nn, device=local_rank), device, k=256)
                out["cpu_baseline_dm"] = dm.pop("cpu_baseline_dm")
                out["dm_size"] = dm
                if dm["hip"].get("sa_equal_to_reference_pipeline") is False:
                    status = EXIT_VERIFY_FAILED
            if not args.no_fm:
                out["fm_query"] = fm_query_leg(device, Q=args.fm_queries, n=args.fm_text_len)
        result_out.write(json.dumps(out) + "\n")
        result_out.flush()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if multi is not None:
        multi.close()
    else:
        ctx.close()
    sys.exit(status)


if __name__ == "__main__":
    main()
