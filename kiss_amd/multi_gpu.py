"""Sharded suffix sort across the GPUs of one node: one process per GPU, torch.distributed (RCCL) for the
exchange, libkiss_hip.so stage calls for the arithmetic.  Design: SURVEY.md section 8(e), DESIGN.md section 7.
(The same pipeline driven by ONE process with peer copies is kiss_hip_multi_* of the C ABI, kiss_amd/csrc/multi.hip.)

  every rank : text replicated; classify text slice r  -> local ascending LMS list (key32, pos)
               ONE small all-gather: 13 counters, list sizes, the near-end positions and the 2^14-bin histogram of the
               first 14 key bits of every rank -> global counts, G key ranges balanced by LMS count, the G x G matrix
               of piece sizes -- the only host read-back of the exchange phase
               stable partition by destination, all_to_all (keys, positions) straight into the buffers the sort works on
               k-ordered sort of the received key range, in place
  rank 0     : receives the sorted pieces (positions + context words) in key-range order behind its own piece,
               runs placement + induction

The only data-path collectives are the all-to-all of the LMS list (12 bytes per LMS suffix) and the gather of the
sorted pieces (8 bytes per LMS suffix); induction is one global dependency chain and stays on one GPU.  No buffer is
copied inside a rank: the stage calls work on views of the context's own arrays (kiss_hip_stage_view), and with one
rank nothing moves at all.
`Backend` abstracts the stage calls so the orchestration can be exercised on CPU (gloo) with a stand-in backend
(tests/test_multi_gpu.py); the product backend is `GpuBackend` (no CPU fallback).
"""
import ctypes

import numpy as np

from . import _lib
from .sorter import _check

HIST_BITS = 14   # 7 bases: the histogram is private to a workgroup in LDS (64 KiB), 16 384 bins split <= 64 key ranges finely enough
EXACT_H0 = 512  # bounded order of the first phase when exact order falls back to rank doubling (KISS_EXACT_H0 of the library)
NEAR_INLINE = 1024  # near-end positions that ride in the fused all-gather (k = 256: about a hundred; more: one extra gather)


class _DevArray:
    """a raw device pointer as something torch.as_tensor wraps without a copy"""

    def __init__(self, ptr, count, typestr):
        self.__cuda_array_interface__ = {"shape": (int(count),), "typestr": typestr, "data": (int(ptr), False), "version": 2}


class GpuBackend:
    """stage calls on torch device tensors through the C ABI (include/kiss_hip.h, stage_* entry points).  The tensors
    handed between the stages are views of the context's own work arrays: no stage copies anything in or out."""

    VIEW_TYPES = {_lib.VIEW_LOCAL_KEYS: "<i8", _lib.VIEW_LOCAL_POS: "<i4", _lib.VIEW_PART_KEYS: "<i8",
                  _lib.VIEW_PART_POS: "<i4", _lib.VIEW_SORTED: "<i4", _lib.VIEW_SORTED_CTX: "<i4"}

    def __init__(self, ctx, S, k):
        import torch
        self.torch = torch
        self.ctx, self.S, self.k = ctx, S, int(k) & 0xFFFFFFFF
        self.n = int(S.numel())
        self.dev = S.device
        self.lib = getattr(ctx, "_lib", None) or _lib.load()  # the library that made the context (default or hooks build)
        self._views = {}

    def _sync(self):
        # The library works on its own HIP stream and takes raw pointers: everything torch / RCCL has queued for
        # these buffers (collectives run on RCCL's streams and only order themselves against torch's current
        # stream) must have finished before a stage call reads them.  Stage calls return synchronised.
        self.torch.cuda.synchronize(self.dev)

    def view(self, which, count):
        """the first `count` entries of one of the context's work arrays as a tensor (zero-copy)"""
        torch = self.torch
        ptr, cap = ctypes.c_void_p(), ctypes.c_uint64()
        _check(self.lib.kiss_hip_stage_view(self.ctx._ctx, which, ctypes.byref(ptr), ctypes.byref(cap)),
               "kiss_hip_stage_view", self.ctx._ctx)
        dt = torch.int64 if self.VIEW_TYPES[which] == "<i8" else torch.int32
        if count == 0 or not ptr.value:
            return torch.empty(0, dtype=dt, device=self.dev)
        assert count <= cap.value, "view beyond the capacity of the work arrays (ensure_capacity first)"
        key = (which, ptr.value, cap.value)
        full = self._views.get(key)
        if full is None:
            self._views = {k_: v for k_, v in self._views.items() if k_[0] != which}  # the array was regrown: stale view
            full = torch.as_tensor(_DevArray(ptr.value, cap.value, self.VIEW_TYPES[which]), device=self.dev)
            self._views[key] = full
        return full[:count]

    def ensure_capacity(self, count):
        """-> True when the work arrays had to regrow: their contents (the classified list) are lost, classify again"""
        cap = ctypes.c_uint64()
        ptr = ctypes.c_void_p()
        _check(self.lib.kiss_hip_stage_view(self.ctx._ctx, _lib.VIEW_LOCAL_POS, ctypes.byref(ptr), ctypes.byref(cap)),
               "kiss_hip_stage_view", self.ctx._ctx)
        if count <= cap.value:
            return False
        self._sync()
        _check(self.lib.kiss_hip_stage_reserve(self.ctx._ctx, int(count)), "kiss_hip_stage_reserve", self.ctx._ctx)
        self._views = {}
        return True

    def classify(self, lo, hi):
        self._sync()
        c = (ctypes.c_uint64 * 13)()
        _check(self.lib.kiss_hip_stage_classify(self.ctx._ctx, ctypes.c_void_p(self.S.data_ptr()), self.n, self.k, int(lo),
                                                int(hi), ctypes.byref(c), None), "kiss_hip_stage_classify", self.ctx._ctx)
        return [int(x) for x in c]

    def local_lms(self):
        m, mf = ctypes.c_uint64(), ctypes.c_uint64()
        _check(self.lib.kiss_hip_stage_local_lms(self.ctx._ctx, None, None, ctypes.byref(m), ctypes.byref(mf)),
               "kiss_hip_stage_local_lms", self.ctx._ctx)
        return self.view(_lib.VIEW_LOCAL_KEYS, m.value), self.view(_lib.VIEW_LOCAL_POS, m.value), int(mf.value)

    def key_hist(self, keys, bits):
        self._sync()
        hist = self.torch.empty(1 << bits, dtype=self.torch.int64, device=self.dev)
        _check(self.lib.kiss_hip_stage_key_hist(self.ctx._ctx, ctypes.c_void_p(keys.data_ptr()), int(keys.numel()), bits,
                                                ctypes.c_void_p(hist.data_ptr()), None), "kiss_hip_stage_key_hist",
               self.ctx._ctx)
        return hist

    def partition(self, keys, pos, bits, splitters, groups):
        self._sync()
        count = int(keys.numel())
        ko, po = self.view(_lib.VIEW_PART_KEYS, count), self.view(_lib.VIEW_PART_POS, count)
        sp = (ctypes.c_uint32 * max(1, groups - 1))(*[int(s) for s in splitters])
        _check(self.lib.kiss_hip_stage_partition(self.ctx._ctx, ctypes.c_void_p(keys.data_ptr()),
                                                 ctypes.c_void_p(pos.data_ptr()), count, bits, sp, groups,
                                                 ctypes.c_void_p(ko.data_ptr()), ctypes.c_void_p(po.data_ptr()), None),
               "kiss_hip_stage_partition", self.ctx._ctx)
        return ko, po

    def recv_buffers(self, count):
        """where the exchange delivers this rank's key range: the buffers stage_sort works on"""
        return self.view(_lib.VIEW_LOCAL_KEYS, count), self.view(_lib.VIEW_LOCAL_POS, count)

    def sorted_buffers(self, count):
        """rank 0: where the sorted pieces are gathered (its own piece, the first key range, is there already)"""
        return self.view(_lib.VIEW_SORTED, count), self.view(_lib.VIEW_SORTED_CTX, count)

    def sort(self, keys, pos):
        """-> (k-ordered positions, their context words from the key payload; 0 = to be gathered), or None when exact
        order was asked for and the ties are deeper than the 32-bases-per-round path handles (KISS_HIP_E_DEEP)"""
        self._sync()
        count = int(pos.numel())
        out, cw = self.view(_lib.VIEW_SORTED, count), self.view(_lib.VIEW_SORTED_CTX, count)
        rc = self.lib.kiss_hip_stage_sort(self.ctx._ctx, ctypes.c_void_p(keys.data_ptr()), ctypes.c_void_p(pos.data_ptr()),
                                          count, self.n, self.k, ctypes.c_void_p(out.data_ptr()),
                                          ctypes.c_void_p(cw.data_ptr()), None)
        if rc == _lib.KISS_HIP_E_DEEP:
            return None
        _check(rc, "kiss_hip_stage_sort", self.ctx._ctx)
        return out, cw

    def refine_exact(self, SA, h0):
        """h0-ordered SA -> exact SA by rank doubling (rank 0, after a pipeline run with k = h0)"""
        self._sync()
        _check(self.lib.kiss_hip_stage_refine_exact(self.ctx._ctx, self.n, int(h0), ctypes.c_void_p(SA.data_ptr()), None),
               "kiss_hip_stage_refine_exact", self.ctx._ctx)
        return SA

    def induce(self, far_all, near_all, counts12, SA=None, far_ctx=None):
        self._sync()
        torch = self.torch
        if SA is None:
            SA = torch.empty(self.n + 1, dtype=torch.int32, device=self.dev)
        c = (ctypes.c_uint64 * 12)(*[int(x) for x in counts12])
        self.last_induce_exact = False
        if getattr(self, "exact_h0", 0):
            # the list is exact_h0-ordered and the caller wants exact order: rank doubling over the LMS suffixes, then the
            # induction (kiss_hip_stage_induce_exact); whether that settled everything is left in last_induce_exact
            done = ctypes.c_int(0)
            _check(self.lib.kiss_hip_stage_induce_exact(self.ctx._ctx, self.n, int(self.exact_h0),
                                                        ctypes.c_void_p(far_all.data_ptr()),
                                                        ctypes.c_void_p(far_ctx.data_ptr()) if far_ctx is not None else None,
                                                        int(far_all.numel()), ctypes.c_void_p(near_all.data_ptr()),
                                                        int(near_all.numel()), ctypes.byref(c), ctypes.c_void_p(SA.data_ptr()),
                                                        None, ctypes.byref(done)),
                   "kiss_hip_stage_induce_exact", self.ctx._ctx)
            self.last_induce_exact = bool(done.value)
            return SA
        _check(self.lib.kiss_hip_stage_induce(self.ctx._ctx, self.n, self.k, ctypes.c_void_p(far_all.data_ptr()),
                                              ctypes.c_void_p(far_ctx.data_ptr()) if far_ctx is not None else None,
                                              int(far_all.numel()), ctypes.c_void_p(near_all.data_ptr()),
                                              int(near_all.numel()), ctypes.byref(c), ctypes.c_void_p(SA.data_ptr()), None),
               "kiss_hip_stage_induce", self.ctx._ctx)
        return SA

    def empty(self, count, dtype):
        return self.torch.empty(count, dtype=dtype, device=self.dev)


class _Comm:
    """torch.distributed wrappers; with the gloo backend device tensors are staged through host memory (tests)"""
    MAX_MSG_BYTES = 1 << 30  # keep single RCCL messages below 2^31 bytes

    def __init__(self, group=None):
        import torch
        import torch.distributed as dist
        self.torch, self.dist, self.group = torch, dist, group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.stage = dist.get_backend(group) == "gloo"

    def _h(self, t):
        return t.cpu() if (self.stage and t.is_cuda) else t

    def all_reduce(self, t):
        h = self._h(t)
        self.dist.all_reduce(h, group=self.group)
        if h is not t:
            t.copy_(h)
        return t

    def all_gather_rows(self, row):
        """row: 1-D int64 tensor of the same length on every rank -> numpy [world, len] on the host.  ONE collective and
        ONE device-to-host read, however many small things the ranks have to tell each other."""
        torch = self.torch
        h = self._h(row).contiguous()
        out = torch.empty(self.world * h.numel(), dtype=h.dtype, device=h.device)
        self.dist.all_gather_into_tensor(out, h, group=self.group)
        return out.cpu().numpy().reshape(self.world, h.numel())

    def all_gather_ints(self, value):
        t = self.torch.tensor([int(value)], dtype=self.torch.int64)
        out = [self.torch.zeros(1, dtype=self.torch.int64) for _ in range(self.world)]
        if not self.stage:
            dev = self.torch.device("cuda", self.torch.cuda.current_device())
            t = t.to(dev)
            out = [o.to(dev) for o in out]
        self.dist.all_gather(out, t, group=self.group)
        return [int(o.item()) for o in out]

    def all_to_all_counts(self, send_counts):
        """-> (recv_counts, largest count of the whole G x G matrix).  The full matrix is all-gathered (G^2 integers)
        so that every rank derives the SAME transport decision from it in all_to_all().  (The sort itself derives the
        matrix from the all-gathered histograms, sharded_suffix_sort; this stays for callers that only know their row.)"""
        mat = self.all_gather_rows(self.torch.tensor([int(c) for c in send_counts], dtype=self.torch.int64,
                                                     device=None if self.stage else
                                                     self.torch.device("cuda", self.torch.cuda.current_device())))
        return [int(mat[src][self.rank]) for src in range(self.world)], int(mat.max())

    def use_collective(self, esz, largest_count):
        """One decision for all ranks: a single all_to_all_single only if EVERY pair's message fits MAX_MSG_BYTES
        (judged on the largest entry of the all-gathered count matrix, identical everywhere)."""
        return self.world > 1 and int(largest_count) * esz <= self.MAX_MSG_BYTES

    def all_to_all(self, out, inp, out_splits, in_splits, largest_count):
        if not self.stage:
            esz = inp.element_size()
            if self.use_collective(esz, largest_count):
                self.dist.all_to_all_single(out, inp, output_split_sizes=out_splits, input_split_sizes=in_splits,
                                            group=self.group)
                return out
            # one rank, or messages too large for one collective: local copy + chunked point-to-point
            in_off = np.concatenate([[0], np.cumsum(in_splits)]).astype(np.int64)
            out_off = np.concatenate([[0], np.cumsum(out_splits)]).astype(np.int64)
            me = self.rank
            out[int(out_off[me]):int(out_off[me + 1])] = inp[int(in_off[me]):int(in_off[me + 1])]
            step = max(1, self.MAX_MSG_BYTES // esz)
            ops = []
            for peer in range(self.world):
                if peer == me:
                    continue
                for a in range(int(in_off[peer]), int(in_off[peer + 1]), step):
                    b = min(a + step, int(in_off[peer + 1]))
                    ops.append(self.dist.P2POp(self.dist.isend, inp[a:b], peer, self.group))
                for a in range(int(out_off[peer]), int(out_off[peer + 1]), step):
                    b = min(a + step, int(out_off[peer + 1]))
                    ops.append(self.dist.P2POp(self.dist.irecv, out[a:b], peer, self.group))
            if ops:
                for req in self.dist.batch_isend_irecv(ops):
                    req.wait()
            return out
        # gloo: pairwise exchange through host memory, source-rank order on the receiver
        hin = inp.cpu()
        in_off = np.concatenate([[0], np.cumsum(in_splits)]).astype(np.int64)
        out_off = np.concatenate([[0], np.cumsum(out_splits)]).astype(np.int64)
        hout = self.torch.empty(int(out_off[-1]), dtype=inp.dtype)
        reqs = []
        for peer in range(self.world):
            piece = hin[int(in_off[peer]):int(in_off[peer + 1])].contiguous()
            if peer == self.rank:
                hout[int(out_off[peer]):int(out_off[peer + 1])] = piece
            elif piece.numel():
                reqs.append(self.dist.isend(piece, peer, group=self.group))
        for peer in range(self.world):
            cnt = int(out_off[peer + 1] - out_off[peer])
            if peer != self.rank and cnt:
                buf = self.torch.empty(cnt, dtype=inp.dtype)
                self.dist.recv(buf, peer, group=self.group)
                hout[int(out_off[peer]):int(out_off[peer + 1])] = buf
        for r in reqs:
            r.wait()
        out.copy_(hout)
        return out

    def gather_to_root(self, piece, counts, make_empty, out=None):
        """variable-size gather in rank order; returns the concatenation on rank 0, None elsewhere.
        out (rank 0, optional): the destination, sum(counts) entries; when its head IS rank 0's own piece (the views of
        GpuBackend) nothing is copied for it.
        RCCL: all receives are posted at once (batch_isend_irecv), so the seven xGMI links into rank 0 carry their
        pieces concurrently instead of one after the other; messages stay below MAX_MSG_BYTES."""
        total = int(sum(counts))
        c0 = int(counts[0])

        def own(dst):
            if c0 and dst[:c0].data_ptr() != piece.data_ptr():
                dst[:c0] = piece
        if self.stage:  # gloo (tests): through host memory, one source at a time
            if self.rank == 0:
                if out is None:
                    out = make_empty(total, piece.dtype)
                own(out)
                off = c0
                for src in range(1, self.world):
                    c = int(counts[src])
                    if c:
                        buf = self.torch.empty(c, dtype=piece.dtype)
                        self.dist.recv(buf, src, group=self.group)
                        out[off:off + c] = buf.to(out.device)
                    off += c
                return out
            if piece.numel():
                self.dist.send(self._h(piece).contiguous(), 0, group=self.group)
            return None
        step = max(1, self.MAX_MSG_BYTES // piece.element_size())
        ops = []
        if self.rank == 0:
            if out is None:
                out = make_empty(total, piece.dtype)
            own(out)
            off = c0
            for src in range(1, self.world):
                c = int(counts[src])
                for a in range(0, c, step):
                    ops.append(self.dist.P2POp(self.dist.irecv, out[off + a:off + min(c, a + step)], src, self.group))
                off += c
        else:
            out = None
            piece = piece.contiguous()
            for a in range(0, int(piece.numel()), step):
                ops.append(self.dist.P2POp(self.dist.isend, piece[a:a + step], 0, self.group))
        if ops:
            for req in self.dist.batch_isend_irecv(ops):
                req.wait()
        return out

    def barrier(self):
        self.dist.barrier(group=self.group)


def choose_splitters(global_hist, groups):
    """key-range boundaries (on the first HIST_BITS key bits) that balance the LMS count over `groups` ranks"""
    h = np.asarray(global_hist, dtype=np.int64)
    total = int(h.sum())
    cum = np.cumsum(h)
    sp = []
    for g in range(1, groups):
        target = (total * g + groups - 1) // groups
        b = int(np.searchsorted(cum, target, side="left")) + 1  # bins [0, b) hold >= target items
        sp.append(min(b, h.size))
    return sp  # non-decreasing; group g = #{s in sp : s <= bin}


def group_counts(local_hist, splitters, groups):
    h = np.asarray(local_hist, dtype=np.int64)
    edges = [0] + [int(s) for s in splitters] + [h.size]
    return [int(h[edges[g]:edges[g + 1]].sum()) for g in range(groups)]


def sharded_suffix_sort(backend, n, group=None, SA=None, timings=None):
    """Runs the sharded pipeline; returns the SA tensor on rank 0 (None on the other ranks)."""
    comm = _Comm(group)
    torch = comm.torch
    G, r = comm.world, comm.rank
    lo, hi = (n * r) // G, (n * (r + 1)) // G
    # timings (a dict, optional): wall-clock ms per phase of THIS rank, each closed by a device synchronisation -- a
    # diagnostic (bench.py --sharded-timings), not for timed runs: the extra synchronisations serialise what may overlap
    import time as _time
    _t = [_time.perf_counter()]

    def phase(name):
        if timings is None:
            return
        if hasattr(backend, "_sync"):
            backend._sync()
        now = _time.perf_counter()
        timings[name] = timings.get(name, 0.0) + 1e3 * (now - _t[0])
        _t[0] = now
    counts = backend.classify(lo, hi)
    keys, pos, m_far = backend.local_lms()
    phase("classify")
    m_local = int(pos.numel())
    bins = 1 << HIST_BITS
    # ---- what the ranks have to tell each other before the exchange, in ONE all-gather and ONE host read:
    #      [13 counters | m_local | m_far | NEAR_INLINE near-end positions | 2^14-bin key histogram (G > 1)]
    near_local = m_local - m_far
    head = torch.zeros(15 + NEAR_INLINE, dtype=torch.int64)
    head[:13] = torch.tensor(counts, dtype=torch.int64)
    head[13], head[14] = m_local, m_far
    row = head.to(keys.device)
    if 0 < near_local <= NEAR_INLINE:
        row[15:15 + near_local] = pos[m_far:].to(torch.int64)
    if G > 1:
        row = torch.cat([row, backend.key_hist(keys[:m_far], HIST_BITS)])
    rows = comm.all_gather_rows(row)
    counts12 = [int(x) for x in rows[:, :12].sum(axis=0)]
    m_locals, m_fars = rows[:, 13].astype(np.int64), rows[:, 14].astype(np.int64)
    near_counts = [int(x) for x in (m_locals - m_fars)]
    m_far_total, near_total = int(m_fars.sum()), int(sum(near_counts))
    # near-end suffixes (only the rank(s) owning the end of the text have any) go to rank 0 as they are
    if max(near_counts) <= NEAR_INLINE:
        near_all = None
        if r == 0:
            near_np = np.concatenate([rows[q, 15:15 + near_counts[q]] for q in range(G)]).astype(np.int32)
            near_all = torch.from_numpy(near_np).to(keys.device) if near_np.size else backend.empty(0, torch.int32)
    else:  # a k in the thousands and beyond: one extra gather (every rank takes this branch alike)
        near_all = comm.gather_to_root(pos[m_far:].clone(), near_counts, backend.empty)
    phase("counters + near-end + histograms (one all-gather)")
    # key ranges balanced by LMS count; the G x G matrix of piece sizes follows from every rank's histogram
    if G > 1:
        hists = rows[:, 15 + NEAR_INLINE:15 + NEAR_INLINE + bins].astype(np.int64)
        splitters = choose_splitters(hists.sum(axis=0), G)
        mat = [group_counts(hists[q], splitters, G) for q in range(G)]
    else:
        splitters, mat = [], [[m_far]]
    send_counts = mat[r]
    recv_counts = [mat[q][r] for q in range(G)]
    piece_counts = [int(sum(mat[q][g] for q in range(G))) for g in range(G)]
    largest = max(max(row_) for row_ in mat)
    R = piece_counts[r]
    # capacity of the work arrays: this rank's received list, and on rank 0 the whole sorted list + the near-end suffixes
    need = max(R, m_far_total + near_total) if r == 0 else R
    if hasattr(backend, "ensure_capacity") and backend.ensure_capacity(need):
        backend.classify(lo, hi)  # the regrown arrays lost the list: classify the slice again (skewed key ranges, (AC)^n)
        keys, pos, m_far = backend.local_lms()
    keys, pos = keys[:m_far], pos[:m_far]
    phase("splitters")
    if G > 1:
        skeys, spos = backend.partition(keys, pos, HIST_BITS, splitters, G)
        phase("partition")
        # the exchange: all-to-all of (key, position) into the buffers the sort works on, receiver keeps source-rank order
        rkeys, rpos = backend.recv_buffers(R) if hasattr(backend, "recv_buffers") else (backend.empty(R, torch.int64),
                                                                                       backend.empty(R, torch.int32))
        comm.all_to_all(rkeys, skeys, recv_counts, send_counts, largest)
        comm.all_to_all(rpos, spos, recv_counts, send_counts, largest)
        phase("exchange")
    else:  # one rank: the classified list IS the list to sort, where it is
        rkeys, rpos = keys, pos
    res = backend.sort(rkeys, rpos)
    phase("sort")
    if int(backend.k) >= n:
        # exact order: a rank whose key range holds ties deeper than the bounded-round path handles (tandem arrays of
        # tens of kilobases) reports it; then every rank runs the pipeline again for k = 256 and rank 0 finishes with
        # rank doubling over the whole SA -- what the single-GPU entry point does by itself (api.hip)
        deep = sum(comm.all_gather_ints(1 if res is None else 0))
        if deep:
            if n < 4 * EXACT_H0 + 1024:
                raise RuntimeError("sharded exact sort: ties too deep on a text too short for the doubling form")
            k_exact = backend.k
            backend.k = EXACT_H0
            backend.exact_h0 = EXACT_H0  # (rank 0's induce: exact order of the LMS suffixes first, where the backend can)
            try:
                out = sharded_suffix_sort(backend, n, group=group, SA=SA, timings=timings)
            finally:
                backend.k = k_exact
                backend.exact_h0 = 0
            if r == 0 and not getattr(backend, "last_induce_exact", False):
                out = backend.refine_exact(out, EXACT_H0)
            comm.barrier()
            return out
    elif res is None:
        raise RuntimeError("kiss_hip_stage_sort: unexpected KISS_HIP_E_DEEP for a bounded k")
    sorted_piece, piece_ctx = res
    # the context words travel with the pieces (4 more bytes per LMS suffix over xGMI instead of a random text
    # gather per LMS suffix on rank 0); rank 0's own piece -- the first key range -- is where it belongs already
    if G > 1:
        dst = backend.sorted_buffers(m_far_total) if (r == 0 and hasattr(backend, "sorted_buffers")) else (None, None)
        far_all = comm.gather_to_root(sorted_piece, piece_counts, backend.empty, out=dst[0])
        ctx_all = comm.gather_to_root(piece_ctx, piece_counts, backend.empty, out=dst[1])
        phase("gather")
    else:
        far_all, ctx_all = sorted_piece, piece_ctx
    out = None
    if r == 0:
        out = backend.induce(far_all, near_all, counts12, SA, far_ctx=ctx_all)
    phase("induce")
    comm.barrier()
    return out
