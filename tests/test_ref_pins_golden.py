"""tests/golden/ref_pins.json: hashes produced by the REFERENCE'S OWN get_lms / put_lms_suffix / induced_sort
(oracle/_ref, see tests/golden/make_ref_golden.py) on inputs any box can regenerate from tests/gen.py.
CPU: the oracle reproduces every pin.  GPU: the HIP path (C ABI) reproduces every pin -- ascending LMS list, k-ordered
LMS list and SA -- including the reference's own test shapes (tests/kiss.cpp: 100-200 k and 10-20 M random bases)."""
import json
import os

import numpy as np
import pytest

from tests.golden.make_ref_golden import make_input

PINS = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "ref_pins.json")))["pins"]


def _id(p):
    return "%s_k%d" % ("-".join(str(x) for x in p["recipe"]), p["k"] if p["k"] < 1 << 31 else -1)


SMALL = [p for p in PINS if p["n"] <= 2_000_000]


@pytest.mark.parametrize("pin", SMALL, ids=[_id(p) for p in SMALL])
def test_oracle_reproduces_reference_pin(oracle, pin):
    S = make_input(pin["recipe"])
    assert S.size == pin["n"]
    sa, lms_sorted = oracle.suffix_sort(S, pin["k"], stages=True)
    assert "%016x" % oracle.fnv(oracle.get_lms(S)[0]) == pin["lms_asc_fnv"]
    assert lms_sorted.size == pin["m"] and "%016x" % oracle.fnv(lms_sorted) == pin["lms_sorted_fnv"]
    assert "%016x" % oracle.fnv(sa) == pin["sa_fnv"]


@pytest.fixture(scope="module")
def ctx():
    import kiss_amd
    c = kiss_amd.Context(max_n=max(p["n"] for p in PINS), device=0)
    yield c
    c.close()


@pytest.mark.gpu
@pytest.mark.parametrize("pin", PINS, ids=[_id(p) for p in PINS])
def test_hip_path_reproduces_reference_pin(ctx, oracle, pin):
    S = make_input(pin["recipe"])
    n = S.size
    sa = ctx.suffix_sort(S, pin["k"])
    assert "%016x" % oracle.fnv(sa) == pin["sa_fnv"]  # oracle.fnv: only the hash function, nothing is sorted on the CPU
    st = ctx.stats()
    if n and st["refine_depth"] == 0:  # stage outputs belong to the requested k unless the doubling fallback ran
        asc, srt, _ = ctx.stage_outputs()
        assert asc.size == pin["m"] - 1
        assert "%016x" % oracle.fnv(np.concatenate([asc, np.array([n], np.uint32)])) == pin["lms_asc_fnv"]
        assert "%016x" % oracle.fnv(np.concatenate([np.array([n], np.uint32), srt])) == pin["lms_sorted_fnv"]
