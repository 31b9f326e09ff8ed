"""ctypes binding of oracle/_ref/libkiss_ref.so: the reference's OWN get_lms / PackedDNAString / put_lms_suffix /
induced_sort compiled unmodified from /root/reference (oracle/ref_driver.cpp says what is and is not reference code).
TEST INFRASTRUCTURE ONLY.  The .so is built in the container that holds /root/reference and travels to the GPU box as a
binary; where neither the tree nor the binary exists, available() is False and the tests that need it skip."""
import ctypes
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB = os.path.join(ORACLE_DIR, "_ref", "libkiss_ref.so")
REF_TREE = "/root/reference/include/biovoltron"


class Ref:
    def __init__(self, lib):
        self.lib = lib
        vp, u32, ci = ctypes.c_void_p, ctypes.c_uint32, ctypes.c_int
        lib.kref_get_lms.argtypes = [vp, u32, ci, vp, vp]
        lib.kref_get_lms.restype = u32
        lib.kref_prefix10.argtypes = [vp, u32, ci, vp, u32, vp]
        lib.kref_prefix10.restype = None
        lib.kref_load125.argtypes = [vp, u32, ci, vp, u32, vp]
        lib.kref_load125.restype = None
        lib.kref_suffix_sort.argtypes = [vp, u32, u32, ci, vp, vp, vp, vp]
        lib.kref_suffix_sort.restype = ci
        lib.kref_max_threads.restype = ci

    @staticmethod
    def threads_for(n, T=None):
        # the reference's chunking (utils.hpp:16-26: len = (n/T) & ~15) is fragile on tiny texts
        if T is None:
            T = 4
        return 1 if n < 10_000 else T

    def max_threads(self):
        return int(self.lib.kref_max_threads())

    def get_lms(self, S, T=None):
        S = np.ascontiguousarray(S, dtype=np.uint8)
        n = S.size
        lms = np.empty(n // 2 + 2, dtype=np.uint32)
        hist = np.zeros(5 * 256, dtype=np.uint32)
        m = self.lib.kref_get_lms(S.ctypes.data, n, self.threads_for(n, T), lms.ctypes.data, hist.ctypes.data)
        return lms[:m], hist.reshape(5, 256)

    def prefix10(self, S, idx):
        S = np.ascontiguousarray(S, dtype=np.uint8)
        idx = np.ascontiguousarray(idx, dtype=np.uint32)
        out = np.empty(idx.size, dtype=np.uint32)
        self.lib.kref_prefix10(S.ctypes.data, S.size, 1, idx.ctypes.data, idx.size, out.ctypes.data)
        return out

    def load125(self, S, idx):
        S = np.ascontiguousarray(S, dtype=np.uint8)
        idx = np.ascontiguousarray(idx, dtype=np.uint32)
        assert idx.size == 0 or int(idx.max()) + 125 <= S.size
        out = np.empty((idx.size, 32), dtype=np.uint8)
        self.lib.kref_load125(S.ctypes.data, S.size, 1, idx.ctypes.data, idx.size, out.ctypes.data)
        return out

    def suffix_sort(self, S, k, T=None, sorted_lms=None, stages=False):
        """sorted_lms given (m entries, sentinel first): only reference code runs (get_lms, put_lms_suffix,
        induced_sort); otherwise the LMS order comes from ref_driver.cpp's restated kref_lms_sort."""
        S = np.ascontiguousarray(S, dtype=np.uint8)
        n = S.size
        SA = np.empty(n + 1, dtype=np.uint32)
        lms_sorted = np.empty(n // 2 + 2, dtype=np.uint32)
        m = ctypes.c_uint32()
        sl = None
        if sorted_lms is not None:
            sl = np.ascontiguousarray(sorted_lms, dtype=np.uint32)
        rc = self.lib.kref_suffix_sort(S.ctypes.data if n else None, n, int(k) & 0xFFFFFFFF, self.threads_for(n, T),
                                       sl.ctypes.data if sl is not None else None, SA.ctypes.data, lms_sorted.ctypes.data,
                                       ctypes.byref(m))
        assert rc == 0, "kref_suffix_sort rc=%d" % rc
        return (SA, lms_sorted[:m.value]) if stages else SA


_inst = None


def available():
    return os.path.exists(LIB) or os.path.isdir(REF_TREE)


def load():
    global _inst
    if _inst is None:
        src = os.path.join(ORACLE_DIR, "ref_driver.cpp")
        if os.path.isdir(REF_TREE) and (not os.path.exists(LIB) or os.path.getmtime(src) > os.path.getmtime(LIB)):
            subprocess.check_call(["make", "-s", "-C", ORACLE_DIR, "ref"])
        _inst = Ref(ctypes.CDLL(LIB))
    return _inst
