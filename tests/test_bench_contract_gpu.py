"""bench.py end to end on the GPU box (small text): exactly one JSON line on stdout carrying the fields the driver
reads, in the single-GPU form and in the sharded form forced onto one rank (RCCL collectives included)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REQUIRED = ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"]


def run_bench(*extra):
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    env["MASTER_ADDR"] = "127.0.0.1"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--text-len", "3000000", "--steps", "2",
                        "--warmup", "1", *extra], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, p.stdout  # stdout carries the result line and nothing else
    return json.loads(lines[0])


@pytest.mark.gpu
def test_bench_single_gpu_line():
    out = run_bench("--cpu-sample", "1000000", "--fm-text-len", "2000000", "--fm-queries", "20000")
    for k in REQUIRED:
        assert k in out, k
    # the run checks itself: the last SA verified on the device, host S -> host SA timed, FM queries/s with parity
    assert out["verified"] is True and out["verify"]["order_violations"] == 0 and len(out["sa_fnv1a64"]) == 16
    e = out["end_to_end"]
    assert e["pinned"]["ms"] > 0 and e["pageable"]["ms"] > 0 and e["pcie_h2d_GBps"] > 1 and e["floor_ms"] > 0
    fq = out["fm_query"]
    assert fq["value"] > 0 and fq["parity_vs_oracle_all_patterns"] is True and fq["roofline"]["bound"] == "infinity_cache"
    assert out["n_gpus"] == 1 and out["steps"] == 2 and out["warmup"] == 1 and out["unit"] == "bases/s"
    assert out["value"] > 0 and out["ms_per_step"] > 0 and out["vs_baseline"] is None and out["dtype"] == "u8"
    assert "workload" in out["config"] and "model" not in out["config"]
    r = out["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    # only the dominant class is timed inside the timed region; the line says where each figure was measured
    assert r["measured_in"].startswith(("timed region", "profiled steps")) and r["kernel_ms_per_step"]
    assert len(r["kernel_ms_per_step"]) > 3 and "extra steps" in r["kernel_ms_per_step_from"]
    c = out["cpu_baseline"]
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1 and c["value"] > 0 and "sample" in c


@pytest.mark.gpu
def test_bench_sharded_path_on_one_rank():
    out = run_bench("--cpu-sample", "0", "--force-sharded", "--no-profile")
    assert out["scaling"] == "strong" and out["value"] > 0 and out["verified"] is True and out["n_gpus"] == 1
    assert "sharded_error" not in out["config"]
