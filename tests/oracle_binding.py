"""ctypes binding of the CPU oracle (oracle/libkiss_oracle.so).  TEST INFRASTRUCTURE ONLY."""
import ctypes
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB = os.path.join(ORACLE_DIR, "libkiss_oracle.so")


class Oracle:
    def __init__(self, lib):
        self.lib = lib
        vp = ctypes.c_void_p
        lib.ko_suffix_sort_dna.argtypes = [vp, ctypes.c_uint32, ctypes.c_uint32, vp, vp, vp]
        lib.ko_suffix_sort_dna.restype = ctypes.c_int
        lib.ko_get_lms.argtypes = [vp, ctypes.c_uint32, vp, vp]
        lib.ko_get_lms.restype = ctypes.c_uint32
        lib.ko_fnv1a64_u32.argtypes = [vp, ctypes.c_uint64]
        lib.ko_fnv1a64_u32.restype = ctypes.c_uint64
        lib.ko_num_threads.restype = ctypes.c_int

    def suffix_sort(self, S, k, stages=False):
        S = np.ascontiguousarray(S, dtype=np.uint8)
        n = S.size
        SA = np.empty(n + 1, dtype=np.uint32)
        if stages:
            lms_sorted = np.empty(n // 2 + 2, dtype=np.uint32)
            m = ctypes.c_uint32()
            rc = self.lib.ko_suffix_sort_dna(S.ctypes.data, n, int(k) & 0xFFFFFFFF, SA.ctypes.data,
                                             lms_sorted.ctypes.data, ctypes.byref(m))
            assert rc == 0
            return SA, lms_sorted[:m.value]
        rc = self.lib.ko_suffix_sort_dna(S.ctypes.data, n, int(k) & 0xFFFFFFFF, SA.ctypes.data, None, None)
        assert rc == 0
        return SA

    def lms_sort(self, S, k, positions):
        """k-ordered order of the given suffix positions (10-mer bucket, then the reference comparator)"""
        S = np.ascontiguousarray(S, dtype=np.uint8)
        lms = np.ascontiguousarray(positions, dtype=np.uint32).copy()
        tmp = np.empty_like(lms)
        self.lib.ko_lms_sort.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_void_p,
                                         ctypes.c_uint32, ctypes.c_void_p]
        self.lib.ko_lms_sort.restype = None
        if lms.size:
            self.lib.ko_lms_sort(S.ctypes.data, S.size, int(k) & 0xFFFFFFFF, lms.ctypes.data, lms.size, tmp.ctypes.data)
        return lms

    def get_lms(self, S):
        """ascending LMS positions (sentinel n appended, like the reference) and the 5x256 histogram"""
        S = np.ascontiguousarray(S, dtype=np.uint8)
        n = S.size
        lms = np.empty(n // 2 + 2, dtype=np.uint32)
        hist = np.zeros(5 * 256, dtype=np.uint32)
        m = self.lib.ko_get_lms(S.ctypes.data, n, lms.ctypes.data, hist.ctypes.data)
        return lms[:m], hist.reshape(5, 256)

    # ---- FM-index (oracle/kiss_oracle_fm.c) ----------------------------------------------------------
    def fm_build(self, S, SA):
        lib = self.lib
        vp = ctypes.c_void_p
        lib.ko_fmi_build.argtypes = [vp, ctypes.c_uint32, vp]
        lib.ko_fmi_build.restype = vp
        S = np.ascontiguousarray(S, dtype=np.uint8)
        SA = np.ascontiguousarray(SA, dtype=np.uint32)
        return OracleFmi(lib, lib.ko_fmi_build(S.ctypes.data, S.size, SA.ctypes.data))

    def read_sequence(self, raw):
        """base codes of a FASTA / plain-text file given as bytes (oracle/kiss_oracle_io.c)"""
        raw = np.frombuffer(bytes(raw), dtype=np.uint8)
        out = np.empty(max(1, raw.size), dtype=np.uint8)
        self.lib.ko_read_sequence.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p]
        self.lib.ko_read_sequence.restype = ctypes.c_uint64
        n = self.lib.ko_read_sequence(raw.ctypes.data if raw.size else None, raw.size, out.ctypes.data)
        return out[:n].copy()

    def fnv(self, a):
        a = np.ascontiguousarray(a, dtype=np.uint32)
        return int(self.lib.ko_fnv1a64_u32(a.ctypes.data, a.size))

    def num_threads(self):
        return int(self.lib.ko_num_threads())


class OracleFmi:
    """handle on a ko_fmi built by the oracle (freed on garbage collection)"""

    def __init__(self, lib, handle):
        self.lib, self.h = lib, ctypes.c_void_p(handle)
        vp = ctypes.c_void_p
        lib.ko_fmi_N.argtypes = [vp]
        lib.ko_fmi_N.restype = ctypes.c_uint64
        lib.ko_fmi_pri.argtypes = [vp]
        lib.ko_fmi_pri.restype = ctypes.c_uint32
        for name in ("cnt", "bwt", "occ1", "occ2", "sa", "b", "bocc"):
            fn = getattr(lib, "ko_fmi_" + name)
            fn.argtypes = [vp]
            fn.restype = vp
        lib.ko_fmi_free.argtypes = [vp]
        lib.ko_fmi_serialize.argtypes = [vp, vp]
        lib.ko_fmi_serialize.restype = ctypes.c_uint64
        lib.ko_fmi_query_batch.argtypes = [vp, vp, ctypes.c_uint32, ctypes.c_uint64, vp, vp,
                                           ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint64), vp, vp]
        self.N = int(lib.ko_fmi_N(self.h))
        self.pri = int(lib.ko_fmi_pri(self.h))

    def _arr(self, name, dtype, count):
        p = getattr(self.lib, "ko_fmi_" + name)(self.h)
        nbytes = count * np.dtype(dtype).itemsize
        return np.frombuffer((ctypes.c_uint8 * nbytes).from_address(p), dtype=dtype).copy()

    def arrays(self):
        N = self.N
        return {
            "cnt": self._arr("cnt", np.uint32, 4),
            "bwt": self._arr("bwt", np.uint8, (N + 3) // 4),
            "occ1": self._arr("occ1", np.uint32, (N // 256 + 1) * 4),
            "occ2": self._arr("occ2", np.uint8, (N // 16 + 1) * 4),
            "sa": self._arr("sa", np.uint32, (N + 3) // 4),
            "b": self._arr("b", np.uint64, (N + 63) // 64),
            "b_occ": self._arr("bocc", np.uint32, N // 64 + 1),
        }

    def serialize(self):
        size = int(self.lib.ko_fmi_serialize(self.h, None))
        buf = (ctypes.c_uint8 * size)()
        self.lib.ko_fmi_serialize(self.h, buf)
        return bytes(buf)

    def query_batch(self, patterns, want_offsets=True):
        patterns = np.ascontiguousarray(patterns, dtype=np.uint8)
        Q, L = patterns.shape
        beg = np.empty(Q, np.uint32)
        end = np.empty(Q, np.uint32)
        tot, chk = ctypes.c_uint64(), ctypes.c_uint64()
        res = {}
        if want_offsets:
            # two passes: size, then fill
            self.lib.ko_fmi_query_batch(self.h, patterns.ctypes.data, L, Q, beg.ctypes.data, end.ctypes.data,
                                        ctypes.byref(tot), ctypes.byref(chk), None, None)
            off = np.empty(max(1, tot.value), np.uint32)
            idx = np.empty(Q + 1, np.uint64)
            self.lib.ko_fmi_query_batch(self.h, patterns.ctypes.data, L, Q, beg.ctypes.data, end.ctypes.data,
                                        ctypes.byref(tot), ctypes.byref(chk), off.ctypes.data, idx.ctypes.data)
            res["offsets"] = off[:tot.value]
            res["offsets_index"] = idx
        else:
            self.lib.ko_fmi_query_batch(self.h, patterns.ctypes.data, L, Q, beg.ctypes.data, end.ctypes.data,
                                        ctypes.byref(tot), ctypes.byref(chk), None, None)
        res.update(beg=beg, end=end, total_hits=int(tot.value), checksum=int(chk.value))
        return res

    def __del__(self):
        try:
            if self.h:
                self.lib.ko_fmi_free(self.h)
                self.h = None
        except Exception:
            pass


def build():
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR])


_inst = None


def load():
    global _inst
    if _inst is None:
        srcs = [os.path.join(ORACLE_DIR, f) for f in os.listdir(ORACLE_DIR) if f.endswith(".c")]
        if not os.path.exists(LIB) or any(os.path.getmtime(s) > os.path.getmtime(LIB) for s in srcs):
            build()
        _inst = Oracle(ctypes.CDLL(LIB))
    return _inst
