#!/usr/bin/env python3
"""Generates tests/golden/ref_pins.json from oracle/_ref/libkiss_ref.so, i.e. from the REFERENCE'S OWN get_lms /
put_lms_suffix / induced_sort compiled unmodified from /root/reference (see oracle/ref_driver.cpp for the one stage --
the LMS sort of kiss1_core.hpp, which needs spdlog -- that is restated there).  Run in the container that holds
/root/reference; the JSON (data only: generator recipe, sizes, FNV-1a-64 hashes) travels, the reference does not.

Each entry: how to regenerate the input with tests/gen.py, k, m, and the FNV-1a-64 of the u32-LE arrays
  lms_asc   : ascending LMS list incl. the sentinel          (reference get_lms)
  lms_sorted: k-ordered LMS list, sentinel first             (kref_lms_sort == oracle, asserted equal here)
  sa        : the suffix array                               (reference put_lms_suffix + induced_sort)
An entry is only written if the oracle (oracle/kiss_oracle.c) produced identical arrays."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests import gen, oracle_binding, ref_binding  # noqa: E402


def make_input(recipe):
    kind = recipe[0]
    if kind == "iid":
        return gen.iid(recipe[1], recipe[2])
    if kind == "genome_like":
        return gen.genome_like(recipe[1], recipe[2])
    if kind == "periodic":
        return gen.periodic(recipe[1], recipe[2], recipe[3], mutations=recipe[4])
    if kind == "endrepeat":  # iid text followed by a copy of its first `cut` bases
        base = gen.iid(recipe[1], recipe[2])
        return np.concatenate([base, base[:recipe[3]]])
    if kind == "const":
        return np.full(recipe[1], recipe[2], np.uint8)
    if kind == "abcd_as_dna":  # the reference's own general-alphabet test draws 'A'..'D' (tests/kiss.cpp:51-69): as codes
        return gen.iid(recipe[1], recipe[2])
    raise ValueError(kind)


RECIPES = [
    ["iid", 150_000, 42],              # tests/kiss.cpp:11-29 shape (100-200 k random bases, k = 256)
    ["iid", 15_000_000, 43],           # tests/kiss.cpp:31-49 shape (10-20 M random bases, k = 256)
    ["genome_like", 2_000_000, 4],
    ["genome_like", 12_000_000, 5],
    ["periodic", 300_000, 1, 21, 6],
    ["periodic", 300_000, 3, 22, 6],
    ["periodic", 300_000, 37, 23, 20],
    ["periodic", 300_000, 171, 24, 200],
    ["periodic", 300_000, 400, 25, 6],
    ["const", 100_000, 0],
    ["const", 100_000, 3],
    ["endrepeat", 200_000, 31, 124], ["endrepeat", 200_000, 32, 125], ["endrepeat", 200_000, 33, 126],
    ["endrepeat", 200_000, 34, 256], ["endrepeat", 200_000, 35, 257], ["endrepeat", 200_000, 36, 374],
    ["endrepeat", 200_000, 37, 375], ["endrepeat", 200_000, 38, 376], ["endrepeat", 200_000, 39, 1000],
    ["iid", 1, 1], ["iid", 2, 2], ["iid", 5, 3], ["iid", 16, 4], ["iid", 50, 5], ["iid", 1000, 6], ["iid", 100_003, 7],
    # round 2: shapes behind the code paths that changed (no merged LMS copy: the near-end rank table; round 0 in one pass;
    # the tied-segment arrays regrown after the first attempt)
    ["periodic", 300_000, 2, 27, 0],            # (AC)^n: every LMS suffix tied after round 0
    ["periodic", 600_000, 2052, 26, 6000],      # higher-order-repeat-like array: a 2052-base unit at 1 % divergence
    ["genome_like", 3_000_000, 31],
    ["iid", 60_000, 51], ["iid", 30_000, 52], ["genome_like", 400_000, 53],   # k so large that a third of the LMS
    # suffixes are near-end (KS below)
    # round 3: BASELINE.json configs[0] -- `suffix_sort example/drosophia_chr1_2.fa -k 256 -t 24` (reference README.md:85-88):
    # n = 48 800 648, the C1 stand-in of SURVEY.md section 8(d) (seed 1), pinned at the config's 24 threads
    ["genome_like", 48_800_648, 1],
]
# orders other than the default (32, 256, exact) for a recipe
KS = {("iid", 60_000, 51): (20_000,), ("iid", 30_000, 52): (10_000, 29_000), ("genome_like", 400_000, 53): (100_000,)}
# reference thread count other than the default 8 for a recipe (OpenMP oversubscribes where the host has fewer cores;
# KISS1's result does not depend on it)
THREADS = {("genome_like", 48_800_648, 1): 24}

if __name__ == "__main__":
    # --append: keep the pins already in ref_pins.json, compute only the (recipe, k) pairs that are missing
    append = "--append" in sys.argv[1:]
    out_path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ref_pins.json")
    orc, ref = oracle_binding.load(), ref_binding.load()
    T = min(8, ref.max_threads())
    pins = []
    T_all = T
    have = set()
    if append and os.path.exists(out_path):
        with open(out_path) as f:
            old = json.load(f)
        pins, T_all = old["pins"], old.get("threads", T)
        have = {(tuple(p["recipe"]), p["k"]) for p in pins}
    for recipe in RECIPES:
        ks = KS.get(tuple(recipe), (256,) if recipe[1] > 5_000_000 else (32, 256, 0xFFFFFFFF))
        if all((tuple(recipe), int(k)) in have for k in ks):
            continue
        S = make_input(recipe)
        T = THREADS.get(tuple(recipe), T_all)
        lms_asc, _ = ref.get_lms(S, T)
        if not np.array_equal(lms_asc, orc.get_lms(S)[0]):
            # seen on (TC)^n, n = 300 000: at the maximal LMS density (every other position) the reference's get_lms
            # returns a list with three entries out of place when it runs on 8 threads, and the right one on 1 or 2 --
            # its output depends on the thread count there.  Such a recipe is pinned with the single-thread run.
            T = 1
            lms_asc, _ = ref.get_lms(S, T)
            print("NOTE: reference get_lms is thread-count dependent on", recipe, "-- pinned at 1 thread", flush=True)
        assert np.array_equal(lms_asc, orc.get_lms(S)[0]), recipe
        for k in ks:
            if (tuple(recipe), int(k)) in have:
                continue
            sa_o, lms_o = orc.suffix_sort(S, k, stages=True)
            sa_r, lms_r = ref.suffix_sort(S, k, T=T, stages=True)                # restated LMS sort + reference induction
            sa_r2 = ref.suffix_sort(S, k, T=T, sorted_lms=lms_o)                  # reference code only, oracle's LMS order
            assert np.array_equal(lms_o, lms_r) and np.array_equal(sa_o, sa_r) and np.array_equal(sa_o, sa_r2), (recipe, k)
            pins.append({"recipe": recipe, "n": int(S.size), "k": int(k), "m": int(lms_r.size), "ref_threads": int(T),
                         "lms_asc_fnv": "%016x" % orc.fnv(lms_asc), "lms_sorted_fnv": "%016x" % orc.fnv(lms_r),
                         "sa_fnv": "%016x" % orc.fnv(sa_r)})
            print(recipe, k, pins[-1]["sa_fnv"], flush=True)
    with open(out_path, "w") as f:
        json.dump({"made_by": "tests/golden/make_ref_golden.py", "threads": T_all, "pins": pins}, f, indent=1)
    print("wrote", len(pins), "pins")
