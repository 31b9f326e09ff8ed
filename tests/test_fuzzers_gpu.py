"""Keeps the differential fuzzers of tools/ alive: a few seconds of each on the GPU box (the long runs are in profiles/)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
@pytest.mark.parametrize("tool,seconds", [("fuzz_parity.py", 8), ("fuzz_fm.py", 8), ("fuzz_fasta.py", 5),
                                          ("fuzz_general.py", 5)])
def test_fuzzer_runs_clean(tool, seconds):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", tool), str(seconds), "12345"], cwd=ROOT,
                       capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-1500:])
    assert " 0 failures" in p.stdout
