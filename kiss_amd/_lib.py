"""ctypes binding of libkiss_hip.so (the C ABI declared in include/kiss_hip.h).

The product path has NO CPU fallback: if the HIP library is missing or fails to
load, importing the entry points raises.  (The CPU oracle under oracle/ is test
infrastructure and is never imported from here.)
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libkiss_hip.so")
# the hooks build of the same sources (-DKISS_HIP_HOOKS: environment switches read at every call, fault injection,
# tracing -- kiss_amd/csrc/kiss_internal.hpp KissOpts).  Test infrastructure: only what asks for it (`load(hooks=True)`,
# `Context(..., hooks=True)`, or KISS_AMD_LIB=hooks for a whole process) ever loads it.
HOOKS_LIB_PATH = os.path.join(_HERE, "libkiss_hip_hooks.so")

KISS_HIP_OK = 0
KISS_HIP_E_INVALID, KISS_HIP_E_NO_DEVICE, KISS_HIP_E_HIP, KISS_HIP_E_NOMEM = -1, -2, -3, -4
KISS_HIP_E_UNSUPPORTED, KISS_HIP_E_INTERNAL, KISS_HIP_E_IO, KISS_HIP_E_DEEP = -5, -6, -7, -8
ALGO_PARALLEL_SORTING = 0
ALGO_PREFIX_DOUBLING = 1
MAX_N = 4294963200
# kiss_hip_stage_view: the ctx's own work arrays (include/kiss_hip.h)
VIEW_LOCAL_KEYS, VIEW_LOCAL_POS, VIEW_PART_KEYS, VIEW_PART_POS, VIEW_SORTED, VIEW_SORTED_CTX = range(6)

KERNEL_CLASSES = [
    "pack", "classify", "radix_hist", "radix_scatter", "scan", "keygather", "flag_compact",
    "place", "induce_count", "induce_scatter", "induce_small", "fm_query", "fm_build", "segrank",
    "group_heads", "isa",
]


class Stats(ctypes.Structure):
    _fields_ = [
        ("n", ctypes.c_uint64), ("m", ctypes.c_uint64), ("k", ctypes.c_uint32), ("depth", ctypes.c_uint32),
        ("lms_rounds", ctypes.c_uint32), ("induce_passes", ctypes.c_uint32), ("near_end", ctypes.c_uint64),
        ("sort_item_rounds", ctypes.c_uint64), ("big_item_rounds", ctypes.c_uint64),
        ("ms_total", ctypes.c_float), ("ms_pack", ctypes.c_float), ("ms_classify", ctypes.c_float),
        ("ms_lms_sort", ctypes.c_float), ("ms_place", ctypes.c_float), ("ms_induce", ctypes.c_float),
        ("ms_kernel", ctypes.c_float * 16), ("launches_kernel", ctypes.c_uint64 * 16),
        ("items_kernel", ctypes.c_uint64 * 16),
        ("refine_items", ctypes.c_uint64), ("refine_depth", ctypes.c_uint32), ("doubling_rounds", ctypes.c_uint32),
        ("ms_refine", ctypes.c_float), ("ms_h2d", ctypes.c_float), ("ms_d2h", ctypes.c_float),
        ("refine_form", ctypes.c_uint32), ("ms_fm_range", ctypes.c_float), ("ms_fm_locate", ctypes.c_float),
        ("tie_run_retries", ctypes.c_uint32), ("reserved_tail_", ctypes.c_uint32),
    ]

    def as_dict(self):
        d = {k: getattr(self, k) for k, _ in self._fields_ if not k.endswith("_kernel")}
        d["kernels"] = {
            name: {"ms": self.ms_kernel[i], "launches": self.launches_kernel[i], "items": self.items_kernel[i]}
            for i, name in enumerate(KERNEL_CLASSES)
        }
        return d


class MultiStats(ctypes.Structure):
    _fields_ = [
        ("n", ctypes.c_uint64), ("m", ctypes.c_uint64), ("ndev", ctypes.c_uint32), ("refine_depth", ctypes.c_uint32),
        ("piece", ctypes.c_uint64 * 8),
        ("ms_total", ctypes.c_float), ("ms_pack", ctypes.c_float), ("ms_classify", ctypes.c_float),
        ("ms_partition", ctypes.c_float), ("ms_exchange", ctypes.c_float), ("ms_sort", ctypes.c_float),
        ("ms_gather", ctypes.c_float), ("ms_induce", ctypes.c_float),
    ]

    def as_dict(self):
        d = {k: getattr(self, k) for k, _ in self._fields_ if k != "piece"}
        d["piece"] = [int(x) for x in self.piece][:max(1, min(8, self.ndev))]
        return d


class VerifyReport(ctypes.Structure):
    _fields_ = [
        ("n", ctypes.c_uint64), ("k", ctypes.c_uint32), ("exact", ctypes.c_uint32), ("ok", ctypes.c_uint32),
        ("sa0_ok", ctypes.c_uint32), ("out_of_range", ctypes.c_uint64), ("duplicates", ctypes.c_uint64),
        ("order_violations", ctypes.c_uint64), ("first_violation", ctypes.c_uint64), ("tied_pairs", ctypes.c_uint64),
        ("digest", ctypes.c_uint64), ("ms", ctypes.c_float), ("reserved_", ctypes.c_uint32),
    ]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_ if k != "reserved_"}


class FmiView(ctypes.Structure):
    _fields_ = [
        ("n_sa", ctypes.c_uint64), ("cnt", ctypes.c_uint32 * 4), ("pri", ctypes.c_uint32),
        ("sa_intv", ctypes.c_uint32), ("bwt", ctypes.c_void_p), ("occ1", ctypes.c_void_p),
        ("occ2", ctypes.c_void_p), ("sa", ctypes.c_void_p), ("b", ctypes.c_void_p), ("b_occ", ctypes.c_void_p),
    ]


class KissHipError(RuntimeError):
    def __init__(self, status, where, detail=""):
        self.status = status
        msg = "%s failed: %s (%d)" % (where, strerror(status), status)
        if detail:
            msg += " [" + detail + "]"
        super().__init__(msg)


_libs = {}


def load(hooks=None):
    """Load libkiss_hip.so (hooks=True: libkiss_hip_hooks.so); raises (never falls back) when it is missing."""
    if hooks is None:
        hooks = os.environ.get("KISS_AMD_LIB", "") == "hooks"
    hooks = bool(hooks)
    if hooks in _libs:
        return _libs[hooks]
    path = HOOKS_LIB_PATH if hooks else LIB_PATH
    # A-B harnesses (tools/ab_libs.sh, tools/rx_ab.sh) name a variant build of the default library by path instead of
    # overwriting the shipped file
    if not hooks and os.environ.get("KISS_AMD_LIB_PATH"):
        path = os.environ["KISS_AMD_LIB_PATH"]
    if not os.path.exists(path):
        raise ImportError(
            "%s not built at %s -- run `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C kiss_amd/csrc`; there is no CPU fallback" % (os.path.basename(path), path))
    lib = ctypes.CDLL(path)
    if hasattr(lib, "kiss_hip_has_hooks"):  # (a variant named by KISS_AMD_LIB_PATH may be a build of an earlier round)
        lib.kiss_hip_has_hooks.restype = ctypes.c_int
        if bool(lib.kiss_hip_has_hooks()) != hooks and not os.environ.get("KISS_AMD_LIB_PATH"):
            raise ImportError("%s is not the %s build" % (path, "hooks" if hooks else "default"))
    elif hooks:
        raise ImportError("%s is not a hooks build" % path)
    vp, u8p = ctypes.c_void_p, ctypes.c_void_p
    lib.kiss_hip_version.restype = ctypes.c_int
    lib.kiss_hip_strerror.restype = ctypes.c_char_p
    lib.kiss_hip_strerror.argtypes = [ctypes.c_int]
    lib.kiss_hip_device_count.argtypes = [ctypes.POINTER(ctypes.c_int)]
    lib.kiss_hip_ctx_create.argtypes = [ctypes.POINTER(vp), ctypes.c_int, ctypes.c_uint64]
    lib.kiss_hip_ctx_create_sized.argtypes = [ctypes.POINTER(vp), ctypes.c_int, ctypes.c_uint64, ctypes.c_uint64]
    lib.kiss_hip_ctx_release_io_buffers.argtypes = [vp]
    ip = ctypes.POINTER(ctypes.c_int)
    lib.kiss_hip_multi_create.argtypes = [ctypes.POINTER(vp), ip, ctypes.c_int, ctypes.c_uint64]
    lib.kiss_hip_multi_destroy.argtypes = [vp]
    lib.kiss_hip_multi_suffix_sort_dna_u32.argtypes = [vp, vp, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_int, vp]
    lib.kiss_hip_multi_suffix_sort_dna_u32_dev.argtypes = [vp, vp, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_int, vp]
    lib.kiss_hip_multi_get_stats.argtypes = [vp, ctypes.POINTER(MultiStats)]
    lib.kiss_hip_multi_ctx.argtypes = [vp, ctypes.c_int]
    lib.kiss_hip_multi_ctx.restype = vp
    lib.kiss_hip_suffix_sort_dna_u32_multi.argtypes = [vp, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_int, vp, ip, ctypes.c_int]
    lib.kiss_hip_stage_view.argtypes = [vp, ctypes.c_int, ctypes.POINTER(vp), ctypes.POINTER(ctypes.c_uint64)]
    lib.kiss_hip_stage_reserve.argtypes = [vp, ctypes.c_uint64]
    lib.kiss_hip_debug_fail_alloc_over.argtypes = [vp, ctypes.c_uint64]
    lib.kiss_hip_debug_fail_alloc_over.restype = ctypes.c_int
    lib.kiss_hip_debug_splitters.argtypes = [vp, ctypes.c_uint64, ctypes.c_int, vp, vp]
    lib.kiss_hip_debug_splitters.restype = ctypes.c_int
    lib.kiss_hip_ctx_destroy.argtypes = [vp]
    lib.kiss_hip_ctx_set_profiling.argtypes = [vp, ctypes.c_int]
    lib.kiss_hip_ctx_set_profiling_mask.argtypes = [vp, ctypes.c_uint64]
    lib.kiss_hip_last_hip_error.argtypes = [vp, ctypes.POINTER(ctypes.c_char_p)]
    lib.kiss_hip_get_stats.argtypes = [vp, ctypes.POINTER(Stats)]
    lib.kiss_hip_ctx_workspace_bytes.argtypes = [vp, ctypes.POINTER(ctypes.c_uint64)]
    lib.kiss_hip_suffix_sort_dna_u32.argtypes = [u8p, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_int, vp, ctypes.c_int]
    lib.kiss_hip_ctx_suffix_sort_dna_u32.argtypes = [vp, u8p, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_int, vp]
    lib.kiss_hip_ctx_suffix_sort_dna_u32_dev.argtypes = [vp, vp, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_int, vp, vp]
    lib.kiss_hip_ctx_get_stage_outputs.argtypes = [vp, vp, vp, vp]
    lib.kiss_hip_ctx_verify_sa_dev.argtypes = [vp, vp, ctypes.c_uint64, ctypes.c_uint32, vp, ctypes.POINTER(VerifyReport), vp]
    lib.kiss_hip_ctx_verify_sa_dev.restype = ctypes.c_int
    lib.kiss_hip_sa_digest_host.argtypes = [vp, ctypes.c_uint64]
    lib.kiss_hip_sa_digest_host.restype = ctypes.c_uint64
    lib.kiss_hip_fnv1a64_host.argtypes = [vp, ctypes.c_uint64, ctypes.c_uint64]
    lib.kiss_hip_fnv1a64_host.restype = ctypes.c_uint64
    lib.kiss_hip_debug_radix_sort.argtypes = [vp, vp, vp, ctypes.c_uint64, ctypes.c_int]
    lib.kiss_hip_debug_scan_u32.argtypes = [vp, vp, ctypes.c_uint64]
    u64 = ctypes.c_uint64
    lib.kiss_hip_stage_classify.argtypes = [vp, vp, u64, ctypes.c_uint32, u64, u64, ctypes.POINTER(u64 * 13), vp]
    lib.kiss_hip_stage_local_lms.argtypes = [vp, vp, vp, ctypes.POINTER(u64), ctypes.POINTER(u64)]
    lib.kiss_hip_stage_key_hist.argtypes = [vp, vp, u64, ctypes.c_int, vp, vp]
    lib.kiss_hip_stage_partition.argtypes = [vp, vp, vp, u64, ctypes.c_int, vp, ctypes.c_int, vp, vp, vp]
    lib.kiss_hip_stage_sort.argtypes = [vp, vp, vp, u64, u64, ctypes.c_uint32, vp, vp, vp]
    lib.kiss_hip_stage_induce.argtypes = [vp, u64, ctypes.c_uint32, vp, vp, u64, vp, u64, ctypes.POINTER(u64 * 12), vp, vp]
    lib.kiss_hip_stage_induce_exact.argtypes = [vp, u64, ctypes.c_uint32, vp, vp, u64, vp, u64, ctypes.POINTER(u64 * 12), vp, vp,
                                                ctypes.POINTER(ctypes.c_int)]
    lib.kiss_hip_stage_refine_exact.argtypes = [vp, u64, ctypes.c_uint32, vp, vp]
    lib.kiss_hip_stage_refine_exact.restype = ctypes.c_int
    lib.kiss_hip_fmi_query_batch_dev.argtypes = [
        vp, ctypes.POINTER(FmiView), vp, ctypes.c_uint32, ctypes.c_uint64, vp, vp,
        ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint64), vp, vp, ctypes.c_uint64, vp]
    lib.kiss_hip_fmi_build_dev.argtypes = [
        vp, vp, ctypes.c_uint64, vp, ctypes.c_uint32, vp, vp, vp, vp, vp, vp,
        ctypes.POINTER(ctypes.c_uint32 * 4), ctypes.POINTER(ctypes.c_uint32), vp]
    lib.kiss_hip_file_size.argtypes = [ctypes.c_char_p, ctypes.POINTER(u64)]
    lib.kiss_hip_ctx_parse_text_dev.argtypes = [vp, vp, u64, vp, ctypes.POINTER(u64), vp]
    lib.kiss_hip_ctx_load_text_file.argtypes = [vp, ctypes.c_char_p, ctypes.POINTER(vp), ctypes.POINTER(u64)]
    lib.kiss_hip_copy_to_host.argtypes = [vp, vp, u64]
    lib.kiss_hip_suffix_sort_u8.argtypes = [vp, u64, vp, ctypes.c_int]
    lib.kiss_hip_suffix_sort_u8.restype = ctypes.c_int
    lib.kiss_hip_ctx_suffix_sort_u8_dev.argtypes = [vp, vp, u64, vp, vp]
    lib.kiss_hip_ctx_suffix_sort_u8_dev.restype = ctypes.c_int
    lib.kiss_hip_free_dev.argtypes = [vp]
    lib.kiss_hip_alloc_dev.argtypes = [ctypes.POINTER(vp), u64]
    lib.kiss_hip_alloc_dev.restype = ctypes.c_int
    for name in ("kiss_hip_file_size", "kiss_hip_ctx_parse_text_dev", "kiss_hip_ctx_load_text_file",
                 "kiss_hip_copy_to_host", "kiss_hip_free_dev"):
        getattr(lib, name).restype = ctypes.c_int
    for name in ("kiss_hip_device_count", "kiss_hip_ctx_create", "kiss_hip_ctx_destroy", "kiss_hip_ctx_set_profiling", "kiss_hip_ctx_set_profiling_mask",
                 "kiss_hip_last_hip_error", "kiss_hip_get_stats", "kiss_hip_ctx_workspace_bytes",
                 "kiss_hip_suffix_sort_dna_u32", "kiss_hip_ctx_suffix_sort_dna_u32",
                 "kiss_hip_ctx_suffix_sort_dna_u32_dev", "kiss_hip_ctx_get_stage_outputs",
                 "kiss_hip_fmi_query_batch_dev", "kiss_hip_fmi_build_dev", "kiss_hip_debug_radix_sort",
                 "kiss_hip_debug_scan_u32", "kiss_hip_stage_classify", "kiss_hip_stage_local_lms",
                 "kiss_hip_stage_key_hist", "kiss_hip_stage_partition", "kiss_hip_stage_sort", "kiss_hip_stage_induce",
                 "kiss_hip_ctx_create_sized", "kiss_hip_ctx_release_io_buffers", "kiss_hip_multi_create",
                 "kiss_hip_multi_destroy", "kiss_hip_multi_suffix_sort_dna_u32", "kiss_hip_multi_suffix_sort_dna_u32_dev",
                 "kiss_hip_multi_get_stats", "kiss_hip_suffix_sort_dna_u32_multi", "kiss_hip_stage_view",
                 "kiss_hip_stage_reserve"):
        getattr(lib, name).restype = ctypes.c_int
    _libs[hooks] = lib
    return lib


def strerror(status):
    return load().kiss_hip_strerror(int(status)).decode()


# every symbol include/kiss_hip.h declares (checked by tests/test_abi.py without a GPU)
EXPORTED_SYMBOLS = [
    "kiss_hip_version", "kiss_hip_strerror", "kiss_hip_device_count", "kiss_hip_ctx_create", "kiss_hip_ctx_destroy",
    "kiss_hip_ctx_set_profiling", "kiss_hip_ctx_set_profiling_mask", "kiss_hip_last_hip_error", "kiss_hip_get_stats", "kiss_hip_ctx_workspace_bytes",
    "kiss_hip_suffix_sort_dna_u32", "kiss_hip_ctx_suffix_sort_dna_u32", "kiss_hip_ctx_suffix_sort_dna_u32_dev",
    "kiss_hip_ctx_get_stage_outputs", "kiss_hip_ctx_verify_sa_dev", "kiss_hip_sa_digest_host", "kiss_hip_fnv1a64_host",
    "kiss_hip_fmi_query_batch_dev", "kiss_hip_fmi_build_dev",
    "kiss_hip_debug_radix_sort", "kiss_hip_debug_scan_u32",
    "kiss_hip_stage_classify", "kiss_hip_stage_local_lms", "kiss_hip_stage_key_hist", "kiss_hip_stage_partition",
    "kiss_hip_stage_sort", "kiss_hip_stage_induce", "kiss_hip_stage_induce_exact", "kiss_hip_stage_refine_exact",
    "kiss_hip_fmi_sizes_for", "kiss_hip_fmi_build_host", "kiss_hip_fmi_query_batch_host",
    "kiss_hip_file_size", "kiss_hip_ctx_parse_text_dev", "kiss_hip_ctx_load_text_file", "kiss_hip_copy_to_host",
    "kiss_hip_free_dev", "kiss_hip_alloc_dev", "kiss_hip_suffix_sort_u8", "kiss_hip_ctx_suffix_sort_u8_dev",
    "kiss_hip_ctx_create_sized", "kiss_hip_ctx_release_io_buffers", "kiss_hip_stage_view", "kiss_hip_stage_reserve",
    "kiss_hip_multi_create", "kiss_hip_multi_destroy", "kiss_hip_multi_suffix_sort_dna_u32",
    "kiss_hip_multi_suffix_sort_dna_u32_dev", "kiss_hip_multi_get_stats", "kiss_hip_multi_ctx",
    "kiss_hip_suffix_sort_dna_u32_multi", "kiss_hip_debug_splitters", "kiss_hip_debug_fail_alloc_over",
    "kiss_hip_has_hooks", "kiss_hip_release_cached_contexts", "kiss_hip_get_stats_sized",
]
