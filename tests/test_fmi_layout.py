"""tests/fmi_layout.py (CPU): section walk of the .fmi layout and the padding-bit mask, on the oracle's serialisation."""
import numpy as np

from tests import gen
from tests.fmi_layout import canonical, sections, with_garbage_padding


def test_sections_and_padding_mask(oracle):
    for n in (1000, 1003, 4095, 100_001):
        S = gen.iid(n, n)
        buf = oracle.fm_build(S, oracle.suffix_sort(S, 32)).serialize()
        sec, N = sections(buf)
        assert N == n + 1 and sec["bwt"][1] == (N + 3) // 4 and sec["b"][1] == ((N + 63) // 64) * 8
        assert canonical(buf) == buf  # the oracle writes zeros into the padding (a choice: the reference leaves junk)
        junk = with_garbage_padding(buf)
        assert (junk != buf) == bool(N % 4 or N % 64)
        assert canonical(junk) == buf
        # only padding bits differ
        diff = np.nonzero(np.frombuffer(junk, np.uint8) != np.frombuffer(buf, np.uint8))[0]
        lo_b, nb = sec["b"]
        lo_w, nw = sec["bwt"]
        assert all((lo_b + nb - 8 <= d < lo_b + nb) or d == lo_w + nw - 1 for d in diff.tolist())
