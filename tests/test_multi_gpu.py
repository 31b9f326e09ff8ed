"""The sharded (one process per GPU) suffix sort of kiss_amd/multi_gpu.py.

* CPU, world_size 2, gloo: the orchestration (windows, splitters, stable partition, all-to-all, gather order,
  near-end hand-over) with a stand-in backend built from the oracle -- no GPU arithmetic involved.
* GPU: the real stage kernels, world_size 1 and world_size 2 (two ranks sharing the one GPU, gloo transport
  staged through host memory; RCCL itself needs one GPU per rank and is exercised by bench.py --gpus N).
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests import gen


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def depth_of(n, k):
    return 0 if k >= n else 125 * (k // 125 + 1)


class OracleBackend:
    """stand-in for GpuBackend made of oracle pieces + numpy (TEST ONLY)"""

    def __init__(self, S, k):
        from tests import oracle_binding
        self.orc = oracle_binding.load()
        self.S, self.k, self.n = S, k, S.size

    def _key32(self, p):
        w = np.zeros(32, dtype=np.uint64)
        seg = self.S[p:p + 32]
        w[:seg.size] = seg
        v = 0
        for b in w.tolist():
            v = (v << 2) | int(b)
        return v

    def classify(self, lo, hi):
        S, n = self.S, self.n
        lms, _ = self.orc.get_lms(S)
        lms = lms[:-1]
        typ = np.zeros(n, dtype=np.int8)
        for i in range(n - 2, -1, -1):
            typ[i] = 1 if S[i] < S[i + 1] else (typ[i + 1] if S[i] == S[i + 1] else 0)
        win = lms[(lms >= lo) & (lms < hi)]
        D = depth_of(n, self.k)
        far = win if D == 0 else win[win.astype(np.int64) + D <= n]
        self._pos = win.astype(np.uint32)
        self._mfar = far.size
        self._keys = np.array([self._key32(int(p)) for p in win], dtype=np.uint64)
        sl = slice(lo, hi)
        cnt = np.bincount(S[sl], minlength=4)
        cntS = np.bincount(S[sl][typ[sl] == 1], minlength=4)
        cntL = np.bincount(S[win], minlength=4) if win.size else np.zeros(4, np.int64)
        return [int(x) for x in list(cnt) + list(cntS) + list(cntL)] + [int(far.size)]

    def local_lms(self):
        return (torch.from_numpy(self._keys.view(np.int64).copy()), torch.from_numpy(self._pos.view(np.int32).copy()),
                self._mfar)

    def key_hist(self, keys, bits):
        top = (keys.numpy().view(np.uint64) >> np.uint64(64 - bits)).astype(np.int64)
        return torch.from_numpy(np.bincount(top, minlength=1 << bits).astype(np.int64))

    def partition(self, keys, pos, bits, splitters, groups):
        top = (keys.numpy().view(np.uint64) >> np.uint64(64 - bits)).astype(np.int64)
        g = np.zeros(top.size, dtype=np.int64)
        for s in splitters:
            g += (top >= s)
        order = np.argsort(g, kind="stable")
        return keys[torch.from_numpy(order)], pos[torch.from_numpy(order)]

    def sort(self, keys, pos):
        p = pos.numpy().view(np.uint32)
        # the received list is the concatenation, in source-rank order, of the ranks' ascending window lists filtered
        # by key range; the windows are disjoint and ordered by rank, so the whole list ascends by text position --
        # that is what keeps the index tie-break of the comparator alive across the exchange
        assert p.size < 2 or np.all(np.diff(p.astype(np.int64)) > 0), "exchange broke ascending position order"
        out = torch.from_numpy(self.orc.lms_sort(self.S, self.k, p).view(np.int32).copy())
        return out, torch.zeros_like(out)  # no context words: "gather them"

    def induce(self, far_all, near_all, counts12, SA=None, far_ctx=None):
        assert far_ctx is not None and far_ctx.numel() == far_all.numel(), "context words travel with the pieces"
        ref_sa, lms_sorted = self.orc.suffix_sort(self.S, self.k, stages=True)
        lms_sorted = lms_sorted[1:]
        D = depth_of(self.n, self.k)
        is_far = np.ones(lms_sorted.size, bool) if D == 0 else (lms_sorted.astype(np.int64) + D <= self.n)
        want_far = lms_sorted[is_far]
        got_far = far_all.numpy().view(np.uint32)
        assert np.array_equal(got_far, want_far), "sharded far order differs from the oracle's k-order"
        want_near = np.sort(lms_sorted[~is_far])
        assert np.array_equal(near_all.numpy().view(np.uint32), want_near), "near-end hand-over"
        cnt = np.bincount(self.S, minlength=4)
        assert counts12[:4] == [int(x) for x in cnt], "all-reduced character counts"
        return torch.from_numpy(ref_sa.view(np.int32).copy())

    def empty(self, count, dtype):
        return torch.empty(count, dtype=dtype)


def _cpu_worker(rank, world, port, n, k, seed, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from kiss_amd import multi_gpu
        S = gen.genome_like(n, seed) if n >= 4096 else gen.iid(n, seed)
        sa = multi_gpu.sharded_suffix_sort(OracleBackend(S, k), n)
        if rank == 0:
            from tests import oracle_binding
            ok = bool(np.array_equal(sa.numpy().view(np.uint32), oracle_binding.load().suffix_sort(S, k)))
            q.put(ok)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n,k", [(3000, 256), (20_000, 256), (20_000, 32), (5000, 0xFFFFFFFF)])
def test_orchestration_two_ranks_gloo_cpu(n, k):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=_cpu_worker, args=(r, 2, port, n, k, 5, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert q.get(timeout=5) is True


class DeepTieBackend(OracleBackend):
    """stand-in whose exact-order sort reports ties too deep on ONE rank (what kiss_hip_stage_sort does with
    KISS_HIP_E_DEEP): the orchestration has to agree on the k = 256 rerun and finish on rank 0 by refine_exact"""

    def __init__(self, S, k, deep_rank):
        super().__init__(S, k)
        self.deep_rank = deep_rank
        self.refined = 0

    def sort(self, keys, pos):
        if self.k >= self.n and dist.get_rank() == self.deep_rank:
            return None
        return super().sort(keys, pos)

    def refine_exact(self, SA, h0):
        from kiss_amd import multi_gpu
        assert h0 == multi_gpu.EXACT_H0 and dist.get_rank() == 0
        want = self.orc.suffix_sort(self.S, h0)  # the rerun delivered the h0-ordered SA
        assert np.array_equal(SA.numpy().view(np.uint32), want)
        self.refined += 1
        return torch.from_numpy(self.orc.suffix_sort(self.S, 0xFFFFFFFF).view(np.int32).copy())


class DeepTieBackendExactInduce(DeepTieBackend):
    """... whose rank-0 induction of the rerun delivers the exact order by itself (what kiss_hip_stage_induce_exact does when
    the doubling over the LMS suffixes settles everything): refine_exact must not be called then"""

    def induce(self, far_all, near_all, counts12, SA=None, far_ctx=None):
        from kiss_amd import multi_gpu
        self.last_induce_exact = False
        if getattr(self, "exact_h0", 0):
            assert self.exact_h0 == multi_gpu.EXACT_H0 and self.k == multi_gpu.EXACT_H0 and dist.get_rank() == 0
            self.last_induce_exact = True
            return torch.from_numpy(self.orc.suffix_sort(self.S, 0xFFFFFFFF).view(np.int32).copy())
        return super().induce(far_all, near_all, counts12, SA, far_ctx=far_ctx)


def _cpu_worker_deep_exact_induce(rank, world, port, n, deep_rank, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from kiss_amd import multi_gpu
        S = gen.genome_like(n, 12)
        be = DeepTieBackendExactInduce(S, 0xFFFFFFFF, deep_rank)
        sa = multi_gpu.sharded_suffix_sort(be, n)
        assert be.k == 0xFFFFFFFF and getattr(be, "exact_h0", 0) == 0  # both restored after the rerun
        if rank == 0:
            from tests import oracle_binding
            q.put(be.refined == 0 and bool(np.array_equal(sa.numpy().view(np.uint32),
                                                          oracle_binding.load().suffix_sort(S, 0xFFFFFFFF))))
    finally:
        dist.destroy_process_group()


def test_exact_order_rerun_skips_refine_when_the_induction_is_exact_gloo_cpu():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=_cpu_worker_deep_exact_induce, args=(r, 2, port, 8000, 1, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert q.get(timeout=5) is True


def _cpu_worker_deep(rank, world, port, n, deep_rank, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from kiss_amd import multi_gpu
        S = gen.genome_like(n, 11)
        be = DeepTieBackend(S, 0xFFFFFFFF, deep_rank)
        sa = multi_gpu.sharded_suffix_sort(be, n)
        assert be.k == 0xFFFFFFFF  # restored after the rerun
        if rank == 0:
            from tests import oracle_binding
            ok = be.refined == 1 and bool(np.array_equal(sa.numpy().view(np.uint32),
                                                         oracle_binding.load().suffix_sort(S, 0xFFFFFFFF)))
            q.put(ok)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("deep_rank", [0, 1])
def test_exact_order_deep_tie_fallback_two_ranks_gloo_cpu(deep_rank):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=_cpu_worker_deep, args=(r, 2, port, 8000, deep_rank, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert q.get(timeout=5) is True


def _decision_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from kiss_amd import multi_gpu
        comm = multi_gpu._Comm()
        comm.MAX_MSG_BYTES = 1000  # lowered so that only ONE pair of the matrix exceeds it
        # rank 0 sends 10 items to itself and 2000 to rank 1; rank 1 sends 5 and 7: only rank 0 / the pair (0 -> 1)
        # sees a split above the limit with 4-byte elements (the decision used to be per rank)
        send = [10, 2000] if rank == 0 else [5, 7]
        recv, largest = comm.all_to_all_counts(send)
        q.put((rank, recv, largest, comm.use_collective(4, largest), comm.use_collective(4, 200)))
    finally:
        dist.destroy_process_group()


def test_transport_decision_is_identical_on_every_rank():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=_decision_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    got = sorted(q.get(timeout=5) for _ in range(2))
    assert got[0][1] == [10, 5] and got[1][1] == [2000, 7]
    assert got[0][2] == got[1][2] == 2000
    assert got[0][3] is False and got[1][3] is False  # both take the chunked point-to-point form
    assert got[0][4] is True and got[1][4] is True


def test_splitters_balance():
    from kiss_amd.multi_gpu import choose_splitters, group_counts
    rng = np.random.default_rng(0)
    h = rng.integers(0, 50, 1 << 16)
    h[100] = 500_000  # a hot bin (poly-A like) can not be split
    for G in (1, 2, 4, 8):
        sp = choose_splitters(h, G)
        assert len(sp) == G - 1 and all(a <= b for a, b in zip(sp, sp[1:]))
        assert sum(group_counts(h, sp, G)) == int(h.sum())


# ---------------------------------------------------------------- GPU ------------------------------------------------
def _gpu_worker(rank, world, port, n, k, seed, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import kiss_amd
        from kiss_amd import multi_gpu
        S = gen.genome_like(n, seed)
        dev = torch.device("cuda", 0)
        d_S = torch.from_numpy(S).to(dev)
        ctx = kiss_amd.Context(max_n=n, device=0)
        sa = multi_gpu.sharded_suffix_sort(multi_gpu.GpuBackend(ctx, d_S, k), n)
        if rank == 0:
            from tests import oracle_binding
            ok = bool(np.array_equal(sa.cpu().numpy().view(np.uint32), oracle_binding.load().suffix_sort(S, k)))
            q.put(ok)
        ctx.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("world", [1, 2, 3])
# (65541, exact): a text that is one long (TTAGGG)n array -- ties deeper than the bounded-round exact path handles, so the
# ranks agree to run k = 256 and rank 0 finishes by rank doubling (found by tools/fuzz_sharded.py)
@pytest.mark.parametrize("n,k", [(300_000, 256), (1_000_000, 32), (200_000, 0xFFFFFFFF), (65_541, 0xFFFFFFFF)])
def test_sharded_pipeline_real_kernels(world, n, k):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=_gpu_worker, args=(r, world, port, n, k, 7, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    assert q.get(timeout=5) is True
