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
    assert out["scaling"] == "strong"  # the N = 1 line of the default (sharded) series
    # BASELINE configs[3]: exact order through PREFIX_DOUBLING on the same text, proven on the device
    x = out["exact_order"]
    assert x["verified"] is True and x["verify"]["exact"] == 1 and x["ms_per_step"] > 0 and x["value"] > 0
    assert x["doubling_over"] in ("lms_suffixes", "suffix_array", "nothing")
    # BASELINE configs[0]: dm-size text -- reference-code pipeline on the host cores beside the HIP path, same SA
    d = out["dm_size"]
    assert d["hip"]["ms_per_step"] > 0 and d["published"]["threads"] == 24
    cd = out["cpu_baseline_dm"]
    if cd is not None:
        assert cd["kind"] == "reference" and cd["value"] > 0 and d["hip"]["sa_equal_to_reference_pipeline"] is True
    # FM leg: range and locate kernels timed apart; the index build (configs[2] names fmindex_build) timed and its .fmi
    # bytes equal to the oracle's serialisation
    assert fq["range_kernel_ms"] > 0 and fq["locate_kernel_ms"] >= 0
    fb = fq["fm_build"]
    assert fb["ms"] > 0 and fb["fmi_equal_to_oracle"] is True and fb["fmi_bytes"] > 0 and fb["cpu_baseline"]["value"] > 0
    # the satellite-rich text beside the headline one, verified like it
    h = out["sensitivity"]["harsh"]
    assert h["verified"] is True and h["ms_per_step"] > 0 and h["workspace_bytes"] > 0 and len(h["sa_digest"]) == 16


@pytest.mark.gpu
def test_bench_sharded_path_on_one_rank():
    out = run_bench("--cpu-sample", "100000", "--force-sharded")
    assert out["scaling"] == "strong" and out["value"] > 0 and out["verified"] is True and out["n_gpus"] == 1
    assert "sharded_error" not in out["config"]
    # the sharded line is as complete as the single-GPU one: roofline from rank 0's launches, CPU baseline from rank 0's host
    assert out["roofline"] and out["roofline"]["kernel"] == "radix_scatter" and out["roofline"]["frac"] > 0
    assert out["cpu_baseline"] and out["cpu_baseline"]["value"] > 0


@pytest.mark.gpu
def test_bench_multi_abi_two_shares_on_one_gpu():
    out = run_bench("--cpu-sample", "0", "--multi-abi", "0,0")
    assert out["verified"] is True and out["value"] > 0 and "kiss_hip_multi" in out["config"]["parallelism"]
    ph = out["config"]["multi_phase_ms"]
    assert ph["ms_sort"] > 0 and ph["ms_induce"] > 0 and ph["ms_total"] >= ph["ms_sort"]


@pytest.mark.gpu
def test_bench_takes_a_fasta_path(tmp_path):
    import numpy as np
    from tests import gen
    S = gen.genome_like(1_500_000, 3)
    fa = tmp_path / "g.fa"
    with open(fa, "w") as f:
        txt = "".join("ACGT"[c] for c in S)
        for rec in range(3):
            f.write(">chr%d some description\n" % rec)
            part = txt[rec * 500_000:(rec + 1) * 500_000]
            for a in range(0, len(part), 60):
                f.write(part[a:a + 60] + "\n")
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--fasta", str(fa), "--steps", "2", "--warmup", "1",
                        "--cpu-sample", "200000", "--no-fm", "--no-dm", "--no-e2e"], cwd=ROOT, env=env, capture_output=True,
                       text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    out = json.loads([ln for ln in p.stdout.splitlines() if ln.strip()][0])
    assert out["data"] == "file" and out["config"]["n"] == S.size and "g.fa" in out["config"]["workload"]
    assert out["verified"] is True and out["value"] > 0 and out["exact_order"]["verified"] is True


@pytest.mark.gpu
def test_sharded_rehearsal_passes_on_one_rank_and_reports_a_failure(monkeypatch):
    """bench.rehearse_sharded: the small sharded sort in front of an N > 1 measurement.  One rank (gloo group of one, the
    only form a one-GPU box can run): the real stages sort 8 M bases, rank 0's device-side check passes -> None; with the
    pipeline made to raise, the verdict is the reason (and every rank of a real job would get one)."""
    import importlib
    import torch
    import torch.distributed as dist
    import kiss_amd
    from kiss_amd import multi_gpu
    sys.path.insert(0, ROOT)
    sys.modules.pop("bench", None)
    bench = importlib.import_module("bench")
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(29950 + os.getpid() % 40)
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        dev = torch.device("cuda", 0)
        n = 9_000_000
        S = bench.gen_text_device(n, 2, dev)
        assert bench.rehearse_sharded(torch, dist, kiss_amd, S, n, 256, 0, 0, dev) is None

        def boom(*a, **kw):
            raise RuntimeError("boom")
        monkeypatch.setattr(multi_gpu, "sharded_suffix_sort", boom)
        err = bench.rehearse_sharded(torch, dist, kiss_amd, S, n, 256, 0, 0, dev)
        assert err is not None and "boom" in err and "rank 0" in err
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("launcher", ["self", "torchrun"])
def test_the_two_rank_code_path_of_bench_with_both_ranks_on_the_one_gpu(launcher):
    """`python bench.py --gpus 2` end to end on a one-GPU box: KISS_BENCH_SHARE_GPU=1 puts both ranks on GPU 0 and makes them
    talk gloo (RCCL refuses two ranks on one device).  Everything else is the code the driver's N = 2 run executes: the
    self-launching parent, the same text on both ranks, the smaller reservation of rank 1, the rehearsal and its agreement,
    the timed loop between barriers, the extra step with phase times, rank 0's verification, the assembled line."""
    env = dict(os.environ, KISS_BENCH_SHARE_GPU="1", MASTER_PORT=str(29700 + os.getpid() % 200))
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    bench_args = [os.path.join(ROOT, "bench.py"), "--gpus", "2", "--text-len", "30000000", "--steps", "2", "--warmup", "1",
                  "--cpu-sample", "0", "--no-fnv", "--rank-timeout", "600"]
    if launcher == "self":  # bench.py starts its two ranks itself
        cmd = [sys.executable] + bench_args
    else:  # the driver's form: one rank per process under torch.distributed.run
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
               "--master-port", env["MASTER_PORT"]] + bench_args
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1, r.stdout
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["scaling"] == "strong" and j["value"] and j["verified"] is True
    cfg = j["config"]
    assert "sharded_error" not in cfg and "ranks_share_one_gpu" in cfg
    assert set(cfg["sharded_phase_ms_rank0"]) >= {"sort", "induce"} or len(cfg["sharded_phase_ms_rank0"]) >= 4
    assert cfg["scaling_model"]["gpus"] == 2
    assert j["value"] == pytest.approx(30000000 * 2 / (2 * j["ms_per_step"] / 1e3), rel=1e-6)
