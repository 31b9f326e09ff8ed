"""Host-side mirror of biovoltron::FMIndex<4, uint32_t, KISS1Sorter<uint32_t>>{.LOOKUP_LEN = 0}
(reference include/biovoltron/algo/align/exact_match/fm_index.hpp; instantiated by
include/command/fmindex_build.hpp:27-29 and fmindex_query.hpp:26-28).

Same member names and meaning: build(ref), save(path) / load(path) in the reference's `.fmi`
byte layout (fm_index.hpp:591-646), get_range(seed), get_offsets(beg, end), plus
query_batch(patterns) which replaces the per-pattern loop of fmindex_query_main
(include/command/fmindex_query.hpp:79-95).

All arithmetic runs in libkiss_hip.so through its C ABI; torch only owns the device buffers.
There is no CPU path.
"""
import ctypes
import struct

import numpy as np

from . import _lib
from .sorter import Context, _check

SA_INTV = 4          # fmindex_build.hpp:27
SORT_LEN = 32        # FMIndex::build sorts with k = 32 whatever the CLI flags say (fm_index.hpp:384-386)
OCC1_INTV, OCC2_INTV, B_OCC_INTV = 256, 16, 64


def _torch():
    import torch
    if not torch.cuda.is_available():
        raise RuntimeError("kiss_amd.FMIndex needs a HIP device (no CPU fallback)")
    return torch


class FMIndex:
    def __init__(self, device=0):
        self.device = int(device)
        self.N = 0
        self.cnt = np.zeros(4, np.uint32)
        self.pri = 0
        self.bwt = self.occ1 = self.occ2 = self.sa = self.b = self.b_occ = None  # torch tensors on the GPU
        self._ctx = None

    # ---- sizes of the .fmi arrays for an SA of N entries -------------------------------------------
    @staticmethod
    def _sizes(N):
        return {
            "bwt": (N + 3) // 4,                      # bytes
            "occ1": (N // OCC1_INTV + 1) * 4,          # u32
            "occ2": (N // OCC2_INTV + 1) * 4,          # u8
            "sa": (N + SA_INTV - 1) // SA_INTV,        # u32
            "b": (N + 63) // 64,                       # u64
            "b_occ": N // B_OCC_INTV + 1,              # u32
        }

    def _alloc(self, N):
        torch = _torch()
        dev = torch.device("cuda", self.device)
        sz = self._sizes(N)
        self.N = N
        self.bwt = torch.zeros(sz["bwt"] + 8, dtype=torch.uint8, device=dev)
        self.occ1 = torch.zeros(sz["occ1"], dtype=torch.int32, device=dev)
        self.occ2 = torch.zeros(sz["occ2"], dtype=torch.uint8, device=dev)
        self.sa = torch.zeros(sz["sa"], dtype=torch.int32, device=dev)
        self.b = torch.zeros(sz["b"] + 1, dtype=torch.int64, device=dev)
        self.b_occ = torch.zeros(sz["b_occ"], dtype=torch.int32, device=dev)

    def _context(self, max_n):
        if self._ctx is None or self._ctx.max_n < max_n:
            if self._ctx is not None:
                self._ctx.close()
            self._ctx = Context(max_n=max(max_n, 1 << 20), device=self.device)
        return self._ctx

    # ---- build (fm_index.hpp:379-451) ------------------------------------------------------------------
    def build(self, ref, sa=None):
        """ref: uint8 array with values 0..3.  Sorts with k = 32 (like the reference) unless `sa` is given."""
        torch = _torch()
        dev = torch.device("cuda", self.device)
        ref = np.ascontiguousarray(ref, dtype=np.uint8)
        n = ref.size
        ctx = self._context(n)
        d_S = torch.from_numpy(ref).to(dev)
        d_SA = torch.empty(n + 1, dtype=torch.int32, device=dev)
        if sa is None:
            ctx.suffix_sort_dev(d_S.data_ptr(), n, d_SA.data_ptr(), k=SORT_LEN)
        else:
            d_SA.copy_(torch.from_numpy(np.ascontiguousarray(sa, dtype=np.uint32).view(np.int32)))
        self._alloc(n + 1)
        cnt = (ctypes.c_uint32 * 4)()
        pri = ctypes.c_uint32()
        lib = _lib.load()
        _check(lib.kiss_hip_fmi_build_dev(ctx._ctx, ctypes.c_void_p(d_S.data_ptr()), n,
                                          ctypes.c_void_p(d_SA.data_ptr()), SA_INTV,
                                          ctypes.c_void_p(self.bwt.data_ptr()), ctypes.c_void_p(self.occ1.data_ptr()),
                                          ctypes.c_void_p(self.occ2.data_ptr()), ctypes.c_void_p(self.sa.data_ptr()),
                                          ctypes.c_void_p(self.b.data_ptr()), ctypes.c_void_p(self.b_occ.data_ptr()),
                                          ctypes.byref(cnt), ctypes.byref(pri), None),
               "kiss_hip_fmi_build_dev", ctx._ctx)
        self.cnt = np.array(list(cnt), dtype=np.uint32)
        self.pri = int(pri.value)
        return self

    # ---- .fmi serialisation (fm_index.hpp:591-646; Serializer: u64 count + raw bytes, nothing when empty) -----
    def to_bytes(self):
        sz = self._sizes(self.N)
        N = self.N
        out = [self.cnt.astype("<u4").tobytes(), struct.pack("<I", self.pri)]

        def vec(count, raw):
            if count:
                out.append(struct.pack("<Q", count))
                out.append(raw)
        vec(N, self.bwt[:sz["bwt"]].cpu().numpy().tobytes())
        vec(sz["occ1"] // 4, self.occ1.cpu().numpy().view(np.uint32).astype("<u4").tobytes())
        vec(sz["occ2"] // 4, self.occ2.cpu().numpy().tobytes())
        vec(sz["sa"], self.sa.cpu().numpy().view(np.uint32).astype("<u4").tobytes())
        vec(2, np.array([0, N], dtype="<u4").tobytes())  # lookup_ for LOOKUP_LEN = 0 (fm_index.hpp:238-258)
        vec(N, self.b[:sz["b"]].cpu().numpy().view(np.uint64).astype("<u8").tobytes())
        vec(sz["b_occ"], self.b_occ.cpu().numpy().view(np.uint32).astype("<u4").tobytes())
        return b"".join(out)

    def save(self, path):
        with open(path, "wb") as f:
            f.write(self.to_bytes())

    @classmethod
    def from_bytes(cls, buf, device=0):
        torch = _torch()
        self = cls(device)
        mv = memoryview(buf)
        self.cnt = np.frombuffer(mv[:16], dtype="<u4").astype(np.uint32)
        self.pri = struct.unpack_from("<I", mv, 16)[0]
        off = 20

        def vec(elem_bytes_of_count):
            nonlocal off
            count = struct.unpack_from("<Q", mv, off)[0]
            off += 8
            nbytes = elem_bytes_of_count(count)
            raw = bytes(mv[off:off + nbytes])
            off += nbytes
            return count, raw
        N, bwt = vec(lambda c: (c + 3) // 4)
        _, occ1 = vec(lambda c: c * 16)
        _, occ2 = vec(lambda c: c * 4)
        _, sa = vec(lambda c: c * 4)
        _, _lookup = vec(lambda c: c * 4)
        _, b = vec(lambda c: ((c + 63) // 64) * 8)
        _, b_occ = vec(lambda c: c * 4)
        if off != len(mv):
            raise ValueError("trailing bytes in .fmi (the reference asserts EOF, fm_index.hpp:642)")
        self._alloc(N)
        dev = self.bwt.device

        def put(dst, raw, dtype):
            a = np.frombuffer(raw, dtype=dtype).copy()
            t = torch.from_numpy(a.view({np.dtype("<u4"): np.int32, np.dtype("<u8"): np.int64,
                                         np.dtype("u1"): np.uint8}[np.dtype(dtype)]))
            dst[:t.numel()].copy_(t.to(dev))
        put(self.bwt, bwt, "u1")
        put(self.occ1, occ1, "<u4")
        put(self.occ2, occ2, "u1")
        put(self.sa, sa, "<u4")
        put(self.b, b, "<u8")
        put(self.b_occ, b_occ, "<u4")
        return self

    @classmethod
    def load(cls, path, device=0):
        with open(path, "rb") as f:
            return cls.from_bytes(f.read(), device)

    # ---- queries -------------------------------------------------------------------------------------------
    def _view(self):
        v = _lib.FmiView()
        v.n_sa = self.N
        for c in range(4):
            v.cnt[c] = int(self.cnt[c])
        v.pri = self.pri
        v.sa_intv = SA_INTV
        v.bwt = self.bwt.data_ptr()
        v.occ1 = self.occ1.data_ptr()
        v.occ2 = self.occ2.data_ptr()
        v.sa = self.sa.data_ptr()
        v.b = self.b.data_ptr()
        v.b_occ = self.b_occ.data_ptr()
        return v

    def query_batch(self, patterns, want_offsets=True, d_patterns=None, keep_on_device=False):
        """patterns: (Q, L) uint8 array with values 0..3 (host) or a device tensor via d_patterns.
        Returns dict(beg, end, total_hits, checksum[, offsets, offsets_index]); keep_on_device: beg / end stay device
        tensors (int32 holding the u32 values) instead of being copied to the host."""
        torch = _torch()
        dev = torch.device("cuda", self.device)
        if d_patterns is None:
            patterns = np.ascontiguousarray(patterns, dtype=np.uint8)
            d_patterns = torch.from_numpy(patterns).to(dev)
        Q, L = int(d_patterns.shape[0]), int(d_patterns.shape[1])
        ctx = self._context(max(self.N, 4 * Q))
        beg = torch.empty(Q, dtype=torch.int32, device=dev)
        end = torch.empty(Q, dtype=torch.int32, device=dev)
        tot = ctypes.c_uint64()
        chk = ctypes.c_uint64()
        lib = _lib.load()
        view = self._view()
        res = {}
        if want_offsets:
            # first pass sizes the output; the library needs a capacity, so run ranges once to learn it
            _check(lib.kiss_hip_fmi_query_batch_dev(ctx._ctx, ctypes.byref(view), ctypes.c_void_p(d_patterns.data_ptr()),
                                                    L, Q, ctypes.c_void_p(beg.data_ptr()),
                                                    ctypes.c_void_p(end.data_ptr()), ctypes.byref(tot),
                                                    ctypes.byref(chk), None, None, 0, None),
                   "kiss_hip_fmi_query_batch_dev", ctx._ctx)
            cap = int(tot.value)
            offsets = torch.empty(max(cap, 1), dtype=torch.int32, device=dev)
            index = torch.empty(Q + 1, dtype=torch.int64, device=dev)
            _check(lib.kiss_hip_fmi_query_batch_dev(ctx._ctx, ctypes.byref(view), ctypes.c_void_p(d_patterns.data_ptr()),
                                                    L, Q, ctypes.c_void_p(beg.data_ptr()),
                                                    ctypes.c_void_p(end.data_ptr()), ctypes.byref(tot),
                                                    ctypes.byref(chk), ctypes.c_void_p(offsets.data_ptr()),
                                                    ctypes.c_void_p(index.data_ptr()), cap, None),
                   "kiss_hip_fmi_query_batch_dev", ctx._ctx)
            res["offsets"] = offsets[:cap].cpu().numpy().view(np.uint32)
            res["offsets_index"] = index.cpu().numpy().view(np.uint64)
        else:
            _check(lib.kiss_hip_fmi_query_batch_dev(ctx._ctx, ctypes.byref(view), ctypes.c_void_p(d_patterns.data_ptr()),
                                                    L, Q, ctypes.c_void_p(beg.data_ptr()),
                                                    ctypes.c_void_p(end.data_ptr()), ctypes.byref(tot),
                                                    ctypes.byref(chk), None, None, 0, None),
                   "kiss_hip_fmi_query_batch_dev", ctx._ctx)
        if keep_on_device:
            res["beg"], res["end"] = beg, end
        else:
            res["beg"] = beg.cpu().numpy().view(np.uint32)
            res["end"] = end.cpu().numpy().view(np.uint32)
        res["total_hits"] = int(tot.value)
        res["checksum"] = int(chk.value)
        return res

    def get_range(self, seed):
        """(beg, end) of one pattern (fm_index.hpp:574-584)."""
        r = self.query_batch(np.asarray(seed, dtype=np.uint8)[None, :], want_offsets=False)
        return int(r["beg"][0]), int(r["end"][0])

    def get_offsets_of(self, seed):
        """hit positions of one pattern in the reference's get_offsets order (fm_index.hpp:453-501)."""
        r = self.query_batch(np.asarray(seed, dtype=np.uint8)[None, :], want_offsets=True)
        return r["offsets"]

    def close(self):
        if self._ctx is not None:
            self._ctx.close()
            self._ctx = None
