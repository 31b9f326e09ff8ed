import os, sys
os.environ.setdefault("KISS_AMD_LIB", "hooks")  # KISS_HIP_* switches exist in the hooks build only
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import kiss_amd
from tests import gen
S = gen.genome_like(2_000_000, 11)
os.environ["KISS_HIP_DUMP_PIVOT"] = "gpurun_out/pivdump"
os.environ["KISS_HIP_PIVOT_SLOTS"] = "2"
with kiss_amd.Context(max_n=S.size) as c:
    c.suffix_sort(S, 32)
