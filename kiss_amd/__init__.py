"""kiss_amd -- MI355X-native k-ordered suffix sorting (hot path of jhhung/kISS).

Only what the path needs: csrc/ (HIP kernels + the C ABI, built into libkiss_hip.so) and the
host-side mirror of the reference's sorter facade.
"""
from ._lib import ALGO_PARALLEL_SORTING, ALGO_PREFIX_DOUBLING, KissHipError, LIB_PATH, load  # noqa: F401
from .sorter import K_UNBOUNDED, Context, KISS1Sorter, KISS2Sorter, MultiContext, suffix_array_bytes  # noqa: F401
