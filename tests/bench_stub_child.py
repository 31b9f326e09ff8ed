"""Stand-in for one bench.py rank (no torch, no GPU): lets the CPU suite drive bench.launch_ranks / parent_main.
Behaviour is chosen by STUB_MODE: ok | fail_rank1 | too_few | fail_sharded_only | hang_rank1 | stuck | line | bad_sa.
line / bad_sa: rank 0 builds its result line with bench.py's OWN assembly code (assemble_line, cpu_baseline on a tiny
sample through the oracle) from made-up measurements -- what the N > 1 line of a real run carries, without a GPU."""
import json
import os
import sys
import time

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
assert os.environ["LOCAL_RANK"] == str(rank) and os.environ["MASTER_ADDR"] == "127.0.0.1" and os.environ["MASTER_PORT"]
mode = os.environ.get("STUB_MODE", "ok")
argv = sys.argv[1:]
if mode == "too_few":
    sys.exit(3)
if mode == "fail_rank1" and rank == 1:
    sys.exit(7)
if mode == "fail_sharded_only" and "replicas" not in argv and rank == 1:
    sys.exit(9)
if mode == "hang_rank1" and rank == 1:
    sys.exit(5)
if mode == "stuck":
    time.sleep(600)  # every rank alive, none finishing: a collective that never completes
if mode in ("line", "bad_sa"):
    if rank == 0:
        import argparse
        ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        sys.path.insert(0, ROOT)
        import numpy as np
        import bench
        a = argparse.Namespace(steps=2, warmup=1, iid=False, harsh=False, seed=2, mode="sharded", no_profile=False,
                               profile_all=False)
        n, m = 1_000_000, 290_000
        agg = {"radix_scatter": {"ms": 3.0, "launches": 20, "items": 10 * m}}
        stage = {"pack": 0.0, "classify": 0.0, "lms_sort": 0.0, "place": 0.0, "induce": 0.0, "total": 0.0}
        st = {"m": m, "lms_rounds": 5, "sort_item_rounds": 2 * m, "big_item_rounds": 1000, "induce_passes": 40}
        out = bench.assemble_line(a, world, True, n, 256, 0, 0.02, agg, agg, a.steps, stage, st, 123456)
        out["cpu_baseline"] = bench.cpu_baseline(np.random.default_rng(1).integers(0, 4, 20_000, dtype=np.uint8), 256)
        if mode == "bad_sa":
            out["verified"], out["unverified_value"], out["value"] = False, out["value"], None
        print(json.dumps(out), flush=True)
    sys.exit(4 if mode == "bad_sa" else 0)
if mode in ("fail_rank1", "hang_rank1") or (mode == "fail_sharded_only" and "replicas" not in argv):
    time.sleep(60)  # rank 0 waits in a "collective" until the launcher terminates it
if rank == 0:
    err = argv[argv.index("--sharded-error") + 1] if "--sharded-error" in argv else None
    print(json.dumps({"n_gpus": world, "argv": argv, "scaling": "weak" if "replicas" in argv else "strong",
                      "sharded_error": err}), flush=True)
