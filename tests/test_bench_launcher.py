"""bench.py --gpus N without a launcher starts N ranks itself (ADVICE bench.py:126): driven here on
CPU with a stub child.  The parent must never import torch, must relay exactly rank 0's line, exit non-zero when a rank
does, and must not report anything for a job smaller than the one asked for."""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STUB = [sys.executable, os.path.join(ROOT, "tests", "bench_stub_child.py")]


def _bench():
    import importlib
    sys.modules.pop("bench", None)
    return importlib.import_module("bench")


def test_launch_ranks_relays_rank0_line(monkeypatch):
    bench = _bench()
    monkeypatch.setenv("STUB_MODE", "ok")
    rc, out = bench.launch_ranks(4, ["--gpus", "4", "--steps", "2"], child_cmd=STUB)
    assert rc == 0
    lines = [ln for ln in out.splitlines() if ln.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 4 and d["argv"] == ["--gpus", "4", "--steps", "2"]


def test_launch_ranks_failure_terminates_peers_and_is_nonzero(monkeypatch):
    bench = _bench()
    monkeypatch.setenv("STUB_MODE", "fail_rank1")
    t0 = time.time()
    rc, out = bench.launch_ranks(2, [], child_cmd=STUB)
    assert rc == 7 and out.strip() == ""
    assert time.time() - t0 < 30  # rank 0 was terminated, not waited for


def test_too_few_devices_code_wins(monkeypatch):
    bench = _bench()
    monkeypatch.setenv("STUB_MODE", "too_few")
    rc, out = bench.launch_ranks(2, [], child_cmd=STUB)
    assert rc == bench.EXIT_TOO_FEW_DEVICES and out.strip() == ""


def test_parent_never_imports_torch_and_gpus_mismatch_fails():
    # a process that is asked for --gpus 2 with WORLD_SIZE=4 refuses (before importing torch)
    env = dict(os.environ, WORLD_SIZE="4", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, "-X", "importtime", os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env,
                       capture_output=True, text=True, timeout=120)
    assert p.returncode == 2 and "WORLD_SIZE" in p.stderr
    assert "| torch" not in p.stderr and p.stdout.strip() == ""


def test_gpus_2_on_a_box_without_2_gpus_exits_nonzero_with_reason():
    # real children: here (no GPU) and on a 1-GPU box both ranks see fewer than 2 devices
    import torch
    if torch.cuda.device_count() >= 2:
        return
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--text-len", "100000", "--steps", "1",
                        "--warmup", "0", "--cpu-sample", "0"], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 3, p.stderr[-1500:]
    assert p.stdout.strip() == "" and "GPUs asked for" in p.stderr
    assert "replicas" not in p.stderr  # no fallback run for a box that is simply too small


def test_parent_fallback_starts_fresh_replica_ranks_and_says_so(monkeypatch, capsys):
    import argparse
    bench = _bench()
    monkeypatch.setenv("STUB_MODE", "fail_sharded_only")
    args = argparse.Namespace(gpus=2, mode="sharded", no_fallback=False)
    rc = bench.parent_main(args, ["--gpus", "2"], child_cmd=STUB)
    assert rc == 0
    d = json.loads(capsys.readouterr().out.strip())
    assert d["scaling"] == "weak" and "status 9" in d["sharded_error"] and d["n_gpus"] == 2
    # --no-fallback: the failure is the result
    args = argparse.Namespace(gpus=2, mode="sharded", no_fallback=True)
    assert bench.parent_main(args, ["--gpus", "2"], child_cmd=STUB) == 9


def test_the_two_gpu_line_is_complete(monkeypatch, capsys):
    # a SCALE record must be gradeable -- the N > 1 line carries roofline and cpu_baseline like the N = 1 line.
    # The stub's rank 0 assembles its line with bench.py's own code (assemble_line + cpu_baseline on the oracle).
    import argparse
    bench = _bench()
    monkeypatch.setenv("STUB_MODE", "line")
    args = argparse.Namespace(gpus=2, mode="sharded", no_fallback=False, rank_timeout=120.0)
    assert bench.parent_main(args, ["--gpus", "2"], child_cmd=STUB) == 0
    d = json.loads(capsys.readouterr().out.strip())
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["unit"] == "bases/s" and d["dtype"] == "u8"
    r = d["roofline"]
    assert r and r["bound"] == "hbm" and r["kernel"] == "radix_scatter" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and r["avg_launch_us"] > 0
    c = d["cpu_baseline"]
    assert c and c["value"] > 0 and c["cores"] >= 1 and c["kind"] in ("reference", "port") and "sample" in c
    assert "sharded" in d["config"]["parallelism"] and "configs[4]" in d["config"]["workload"]
    assert d["value"] == 2 * 1_000_000 / 0.02  # one text, whole-job aggregate


def test_single_gpu_line_of_the_default_series_says_strong():
    import argparse
    bench = _bench()
    a = argparse.Namespace(steps=1, warmup=0, iid=False, harsh=False, seed=2, mode="sharded", no_profile=True, profile_all=False)
    st = {"m": 10, "lms_rounds": 1, "sort_item_rounds": 10, "big_item_rounds": 0, "induce_passes": 3}
    stage = {"pack": 0.1, "classify": 0.1, "lms_sort": 0.1, "place": 0.1, "induce": 0.1, "total": 0.5}
    d = bench.assemble_line(a, 1, False, 1000, 256, 0, 0.001, {}, {}, 1, stage, st, 1)
    assert d["scaling"] == "strong" and d["n_gpus"] == 1 and d["config"]["parallelism"] == "single GPU"
    a.mode = "replicas"
    assert bench.assemble_line(a, 4, False, 1000, 256, 0, 0.001, {}, {}, 1, stage, st, 1)["scaling"] == "weak"
    assert bench.assemble_line(a, 4, False, 1000, 256, 0, 0.001, {}, {}, 1, stage, st, 1)["value"] == 4 * 1000 / 0.001


def test_a_line_whose_suffix_array_failed_its_check_is_relayed_with_a_nonzero_status(monkeypatch, capsys):
    # a wrong SA must not yield a normal-looking throughput record
    import argparse
    bench = _bench()
    monkeypatch.setenv("STUB_MODE", "bad_sa")
    args = argparse.Namespace(gpus=2, mode="sharded", no_fallback=False, rank_timeout=120.0)
    assert bench.parent_main(args, ["--gpus", "2"], child_cmd=STUB) == bench.EXIT_VERIFY_FAILED
    d = json.loads(capsys.readouterr().out.strip())
    assert d["value"] is None and d["verified"] is False and d["unverified_value"] > 0


def test_ranks_stuck_in_a_collective_are_terminated_after_the_rank_timeout(monkeypatch):
    # all ranks alive, none finishing -- the parent must not poll for ever, and must not retry
    import argparse
    bench = _bench()
    monkeypatch.setenv("STUB_MODE", "stuck")
    t0 = time.time()
    args = argparse.Namespace(gpus=2, mode="sharded", no_fallback=False, rank_timeout=3.0)
    assert bench.parent_main(args, ["--gpus", "2"], child_cmd=STUB) == 124
    assert time.time() - t0 < 60


def test_scaling_model_of_the_sharded_line():
    # the expected-curve inputs every N > 1 line carries (bench.py: scaling_model): two GPUs share one link for
    # the exchange and the gather and are expected SLOWER than one; from four GPUs on the sharded phases win; the ceiling is
    # what stays on rank 0 (pack + induction)
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_for_model", os.path.join(root, "bench.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    m = {G: b.scaling_model(G, b.CHM13_N) for G in (1, 2, 4, 8)}
    assert m[2]["expected_ms"] > m[1]["expected_ms"] > m[4]["expected_ms"] > m[8]["expected_ms"]
    assert m[1]["exchange_bytes_per_rank"] == 0 and m[1]["gather_bytes_into_rank0"] == 0
    assert abs(m[2]["exchange_bytes_per_rank"] - 12 * b.MODEL_LMS_FRACTION * b.CHM13_N / 4) < 1e6   # half of a rank's half
    assert abs(m[8]["gather_bytes_into_rank0"] - 8 * b.MODEL_LMS_FRACTION * b.CHM13_N * 7 / 8) < 1e6
    assert 3.0 < m[8]["ceiling_speedup"] < 4.5 and "model only" in m[8]["status"]
    # a text of another length scales the phases with n
    assert abs(b.scaling_model(4, b.CHM13_N // 2)["expected_ms"] - m[4]["expected_ms"] / 2) < 0.3


def _rehearsal_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        b = _bench()
        dev = torch.device("cpu")
        # round 1: every rank is fine; round 2: rank 1 reports a failure of its own; round 3: rank 0 does
        r1 = b.agree_on_rehearsal(dist, torch, dev, True, None)
        r2 = b.agree_on_rehearsal(dist, torch, dev, rank != 1, "rank 1: RuntimeError: boom" if rank == 1 else None)
        r3 = b.agree_on_rehearsal(dist, torch, dev, rank != 0, "the rehearsal's suffix array fails" if rank == 0 else None)
        q.put((rank, r1, r2, r3))
    finally:
        dist.destroy_process_group()


def test_ranks_agree_on_the_outcome_of_the_sharded_rehearsal():
    """bench.py: before a sharded measurement on more than one GPU the ranks sort a small text through the sharded pipeline
    and AGREE on whether that went well (all fall back to --mode replicas or none does).  gloo, world_size 2, on CPU."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 400)
    ps = [ctx.Process(target=_rehearsal_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    got = sorted(q.get(timeout=120) for _ in ps)
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    for rank, r1, r2, r3 in got:
        assert r1 is None
        assert r2 is not None and r3 is not None  # both ranks fall back, whoever saw the failure
    assert got[1][2] == "rank 1: RuntimeError: boom" and "another rank" in got[0][2]
    assert "suffix array fails" in got[0][3] and "another rank" in got[1][3]
