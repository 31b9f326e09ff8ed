"""Host-side mirror of the reference facade for the hot path.

Mirrors biovoltron::KISS1Sorter<uint32_t> / KISS2Sorter<uint32_t>
(reference include/biovoltron/algo/sort/kiss1_sorter.hpp:8-50, kiss2_sorter.hpp:8-50):
static `get_suffix_array_dna(S, k, num_threads)` returning an SA of n+1 entries, and
`prepare_aligned_ref`.  `num_threads` is accepted and ignored: no result depends on it
(the reference's KISS1 output is thread-count independent as well).

Everything here calls the C ABI of libkiss_hip.so; there is no CPU path.
"""
import ctypes

import numpy as np

from . import _lib

K_UNBOUNDED = 0xFFFFFFFF  # the CLI's `-k -1` (size_t max truncated to uint32_t, suffix_sort.hpp:35-37)


_CTX_LIB = {}  # context handle -> the library that made it (default or hooks build)


def _check(status, where, ctx=None):
    if status != _lib.KISS_HIP_OK:
        detail = ""
        if ctx is not None:
            msg = ctypes.c_char_p()
            lib = _CTX_LIB.get(getattr(ctx, "value", ctx)) or _lib.load()
            code = lib.kiss_hip_last_hip_error(ctx, ctypes.byref(msg))
            if code:
                detail = "hip error %d: %s" % (code, (msg.value or b"").decode())
        raise _lib.KissHipError(status, where, detail)


class Context:
    """Device workspace for texts up to max_n bases on one GPU (kiss_hip_ctx)."""

    def __init__(self, max_n, device=0, profiling=False, lms_capacity=0, hooks=None):
        """lms_capacity (optional): capacity of the per-LMS-suffix work arrays instead of 0.32 max_n -- a rank > 0 of a
        sharded sort holds about 1/G of the LMS suffixes (kiss_hip_ctx_create_sized; the arrays regrow on demand).
        hooks=True: a context of the hooks build (libkiss_hip_hooks.so), the only one whose behaviour KISS_HIP_*
        environment switches change (tests of the rare paths)."""
        self._lib = _lib.load(hooks)
        self._ctx = ctypes.c_void_p()
        _check(self._lib.kiss_hip_ctx_create_sized(ctypes.byref(self._ctx), int(device), int(max_n), int(lms_capacity)),
               "kiss_hip_ctx_create_sized")
        _CTX_LIB[self._ctx.value] = self._lib
        self._owned = True
        self.max_n = int(max_n)
        self.device = int(device)
        if profiling:
            self.set_profiling(True)

    @classmethod
    def _borrowed(cls, handle, max_n, device):
        """a kiss_hip_ctx owned by something else (a MultiContext's per-device context): never destroyed from here"""
        self = cls.__new__(cls)
        self._lib = _lib.load()
        self._ctx = ctypes.c_void_p(handle)
        self._owned = False
        self.max_n, self.device = int(max_n), int(device)
        return self

    def close(self):
        if self._ctx and self._owned:
            _CTX_LIB.pop(self._ctx.value, None)
            self._lib.kiss_hip_ctx_destroy(self._ctx)
        self._ctx = ctypes.c_void_p()

    def release_io_buffers(self):
        """gives back the device-side copies of the caller's buffers the host-pointer entry points keep between calls"""
        _check(self._lib.kiss_hip_ctx_release_io_buffers(self._ctx), "kiss_hip_ctx_release_io_buffers", self._ctx)

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def load_text_file(self, path):
        """FASTA / plain-text file -> (device pointer to base codes, n); parsed on the GPU (reference read_sequence,
        include/utils/io.hpp:6-18).  Free the buffer with `free_dev`."""
        d_S = ctypes.c_void_p()
        n = ctypes.c_uint64()
        _check(self._lib.kiss_hip_ctx_load_text_file(self._ctx, str(path).encode(), ctypes.byref(d_S), ctypes.byref(n)),
               "kiss_hip_ctx_load_text_file", self._ctx)
        return d_S.value, n.value

    def free_dev(self, ptr):
        if ptr:
            _check(self._lib.kiss_hip_free_dev(ctypes.c_void_p(ptr)), "kiss_hip_free_dev")

    def read_sequence(self, path):
        """the base codes of a FASTA / plain-text file as a host numpy array (tests, small files)"""
        ptr, n = self.load_text_file(path)
        try:
            out = np.empty(n, dtype=np.uint8)
            if n:
                _check(self._lib.kiss_hip_copy_to_host(out.ctypes.data, ctypes.c_void_p(ptr), n), "kiss_hip_copy_to_host")
        finally:
            self.free_dev(ptr)
        return out

    def set_profiling(self, on, classes=None):
        """per-kernel HIP-event timing: all classes, or only the named ones (_lib.KERNEL_CLASSES)"""
        if on and classes is not None:
            mask = 0
            for c in classes:
                mask |= 1 << _lib.KERNEL_CLASSES.index(c)
            _check(self._lib.kiss_hip_ctx_set_profiling_mask(self._ctx, mask), "kiss_hip_ctx_set_profiling_mask")
        else:
            _check(self._lib.kiss_hip_ctx_set_profiling(self._ctx, 1 if on else 0), "kiss_hip_ctx_set_profiling")

    def workspace_bytes(self):
        v = ctypes.c_uint64()
        _check(self._lib.kiss_hip_ctx_workspace_bytes(self._ctx, ctypes.byref(v)), "kiss_hip_ctx_workspace_bytes")
        return v.value

    def stats(self):
        st = _lib.Stats()
        _check(self._lib.kiss_hip_get_stats(self._ctx, ctypes.byref(st)), "kiss_hip_get_stats")
        return st.as_dict()

    def suffix_sort(self, S, k=256, algo=_lib.ALGO_PARALLEL_SORTING):
        """S: uint8 numpy array (values 0..3) in host memory -> SA (uint32, n+1)."""
        S = np.ascontiguousarray(S, dtype=np.uint8)
        n = S.size
        SA = np.empty(n + 1, dtype=np.uint32)
        _check(self._lib.kiss_hip_ctx_suffix_sort_dna_u32(self._ctx, S.ctypes.data, n, int(k) & 0xFFFFFFFF, int(algo),
                                                          SA.ctypes.data), "kiss_hip_ctx_suffix_sort_dna_u32", self._ctx)
        return SA

    def suffix_sort_host(self, S, SA, k=256, algo=_lib.ALGO_PARALLEL_SORTING):
        """host S (uint8 numpy, n) -> host SA (uint32 numpy, n+1, caller-allocated, e.g. page-locked): the reference's
        timed region; kiss_hip_ctx_suffix_sort_dna_u32 without the allocation `suffix_sort` does."""
        assert S.dtype == np.uint8 and SA.dtype == np.uint32 and SA.size == S.size + 1
        assert S.flags["C_CONTIGUOUS"] and SA.flags["C_CONTIGUOUS"]
        _check(self._lib.kiss_hip_ctx_suffix_sort_dna_u32(self._ctx, S.ctypes.data, S.size, int(k) & 0xFFFFFFFF, int(algo),
                                                          SA.ctypes.data), "kiss_hip_ctx_suffix_sort_dna_u32", self._ctx)
        return SA

    def suffix_sort_dev(self, d_S_ptr, n, d_SA_ptr, k=256, algo=_lib.ALGO_PARALLEL_SORTING, stream=None):
        """Device-resident form: raw device pointers (e.g. torch tensor .data_ptr())."""
        _check(self._lib.kiss_hip_ctx_suffix_sort_dna_u32_dev(self._ctx, ctypes.c_void_p(d_S_ptr), int(n),
                                                              int(k) & 0xFFFFFFFF, int(algo), ctypes.c_void_p(d_SA_ptr),
                                                              ctypes.c_void_p(stream or 0)),
               "kiss_hip_ctx_suffix_sort_dna_u32_dev", self._ctx)

    def verify_sa_dev(self, d_S_ptr, n, d_SA_ptr, k=256, stream=None):
        """device-side check of a suffix array (kiss_hip_ctx_verify_sa_dev) -> report dict; raises nothing on a bad SA,
        read report["ok"]"""
        rep = _lib.VerifyReport()
        _check(self._lib.kiss_hip_ctx_verify_sa_dev(self._ctx, ctypes.c_void_p(d_S_ptr), int(n), int(k) & 0xFFFFFFFF,
                                                    ctypes.c_void_p(d_SA_ptr), ctypes.byref(rep),
                                                    ctypes.c_void_p(stream or 0)), "kiss_hip_ctx_verify_sa_dev", self._ctx)
        return rep.as_dict()

    def stage_outputs(self):
        """(ascending LMS positions, k-ordered LMS positions, counts[12]) of the last sort."""
        st = self.stats()
        m = st["m"]
        asc = np.empty(m, dtype=np.uint32)
        srt = np.empty(m, dtype=np.uint32)
        counts = np.zeros(12, dtype=np.uint64)
        _check(self._lib.kiss_hip_ctx_get_stage_outputs(self._ctx, asc.ctypes.data, srt.ctypes.data,
                                                        counts.ctypes.data), "kiss_hip_ctx_get_stage_outputs", self._ctx)
        return asc, srt, counts


class MultiContext:
    """Several GPUs of one node driven by THIS process (kiss_hip_multi, include/kiss_hip.h): the LMS sort sharded by key
    range over `devices`, peer copies over xGMI, induction on devices[0].  A device may be listed more than once."""

    def __init__(self, devices, max_n):
        self._lib = _lib.load()
        self._mc = ctypes.c_void_p()
        self.devices = [int(d) for d in devices]
        arr = (ctypes.c_int * len(self.devices))(*self.devices)
        _check(self._lib.kiss_hip_multi_create(ctypes.byref(self._mc), arr, len(self.devices), int(max_n)),
               "kiss_hip_multi_create")
        self.max_n = int(max_n)

    def close(self):
        if self._mc:
            self._lib.kiss_hip_multi_destroy(self._mc)
            self._mc = ctypes.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def rank_context(self, rank=0):
        """the per-device context of one share (statistics, profiling switches); owned by this object"""
        h = self._lib.kiss_hip_multi_ctx(self._mc, int(rank))
        if not h:
            raise IndexError("no such share")
        return Context._borrowed(h, self.max_n, self.devices[rank])

    def suffix_sort(self, S, k=256, algo=_lib.ALGO_PARALLEL_SORTING):
        S = np.ascontiguousarray(S, dtype=np.uint8)
        SA = np.empty(S.size + 1, dtype=np.uint32)
        _check(self._lib.kiss_hip_multi_suffix_sort_dna_u32(self._mc, S.ctypes.data if S.size else None, S.size,
                                                            int(k) & 0xFFFFFFFF, int(algo), SA.ctypes.data),
               "kiss_hip_multi_suffix_sort_dna_u32")
        return SA

    def suffix_sort_dev(self, d_S_ptr, n, d_SA_ptr, k=256, algo=_lib.ALGO_PARALLEL_SORTING):
        """d_S / d_SA: raw device pointers on devices[0]"""
        _check(self._lib.kiss_hip_multi_suffix_sort_dna_u32_dev(self._mc, ctypes.c_void_p(d_S_ptr), int(n),
                                                                int(k) & 0xFFFFFFFF, int(algo), ctypes.c_void_p(d_SA_ptr)),
               "kiss_hip_multi_suffix_sort_dna_u32_dev")

    def stats(self):
        st = _lib.MultiStats()
        _check(self._lib.kiss_hip_multi_get_stats(self._mc, ctypes.byref(st)), "kiss_hip_multi_get_stats")
        return st.as_dict()


FNV1A64_SEED = 0xcbf29ce484222325


def fnv1a64(buf, seed=FNV1A64_SEED):
    """FNV-1a-64 over the bytes of a numpy array (host helper of the library, no device)"""
    a = np.ascontiguousarray(buf)
    return int(_lib.load().kiss_hip_fnv1a64_host(a.ctypes.data, a.nbytes, seed))


def sa_digest(SA):
    """the order-sensitive digest kiss_hip_ctx_verify_sa_dev reports, recomputed on the host"""
    a = np.ascontiguousarray(SA, dtype=np.uint32)
    return int(_lib.load().kiss_hip_sa_digest_host(a.ctypes.data, a.size))


class KISS1Sorter:
    """PARALLEL_SORTING (kiss1_sorter.hpp:8-50)."""
    algo = _lib.ALGO_PARALLEL_SORTING

    @staticmethod
    def prepare_aligned_ref(seq):
        # reference: copy into a 64-byte aligned vector<uint8_t> (kiss1_sorter.hpp:46-49); numpy owns alignment here
        return np.ascontiguousarray(np.asarray(seq, dtype=np.uint8))

    @classmethod
    def get_suffix_array_dna(cls, S, k=256, num_threads=None, device=0, devices=None):
        """devices (optional): shard the LMS sort over these GPUs (kiss_hip_suffix_sort_dna_u32_multi)"""
        S = cls.prepare_aligned_ref(S)
        SA = np.empty(S.size + 1, dtype=np.uint32)
        lib = _lib.load()
        if devices is not None:
            arr = (ctypes.c_int * len(devices))(*[int(d) for d in devices])
            _check(lib.kiss_hip_suffix_sort_dna_u32_multi(S.ctypes.data, S.size, int(k) & 0xFFFFFFFF, cls.algo,
                                                          SA.ctypes.data, arr, len(devices)),
                   "kiss_hip_suffix_sort_dna_u32_multi")
            return SA
        _check(lib.kiss_hip_suffix_sort_dna_u32(S.ctypes.data, S.size, int(k) & 0xFFFFFFFF, cls.algo, SA.ctypes.data,
                                                int(device)), "kiss_hip_suffix_sort_dna_u32")
        return SA


def suffix_array_bytes(data, device=0, hooks=None):
    """Exact suffix array (uint32, n + 1 entries, SA[0] = n) of a text over the byte alphabet: the general-alphabet
    entry of the reference facade, KISS1Sorter::get_suffix_array (kiss1_sorter.hpp:28-45 -> kiss1_core.hpp:270-311)."""
    S = np.ascontiguousarray(np.frombuffer(data, dtype=np.uint8) if isinstance(data, (bytes, bytearray)) else data,
                             dtype=np.uint8)
    SA = np.empty(S.size + 1, dtype=np.uint32)
    lib = _lib.load(hooks)  # (hooks=True: the hooks build, for the tests that switch paths through the environment)
    _check(lib.kiss_hip_suffix_sort_u8(S.ctypes.data if S.size else None, S.size, SA.ctypes.data, int(device)),
           "kiss_hip_suffix_sort_u8")
    return SA


class KISS2Sorter(KISS1Sorter):
    """PREFIX_DOUBLING (kiss2_sorter.hpp:8-50): exact suffix array for k >= n; for a bounded k the deterministic
    k-ordered array of KISS1 (the reference's own bounded-k KISS2 result depends on its thread count)."""
    algo = _lib.ALGO_PREFIX_DOUBLING
