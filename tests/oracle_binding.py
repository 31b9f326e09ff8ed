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

    def get_lms(self, S):
        """ascending LMS positions (sentinel n appended, like the reference) and the 5x256 histogram"""
        S = np.ascontiguousarray(S, dtype=np.uint8)
        n = S.size
        lms = np.empty(n // 2 + 2, dtype=np.uint32)
        hist = np.zeros(5 * 256, dtype=np.uint32)
        m = self.lib.ko_get_lms(S.ctypes.data, n, lms.ctypes.data, hist.ctypes.data)
        return lms[:m], hist.reshape(5, 256)

    def fnv(self, a):
        a = np.ascontiguousarray(a, dtype=np.uint32)
        return int(self.lib.ko_fnv1a64_u32(a.ctypes.data, a.size))

    def num_threads(self):
        return int(self.lib.ko_num_threads())


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
