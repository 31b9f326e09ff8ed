"""Stand-in for one bench.py rank (no torch, no GPU): lets the CPU suite drive bench.launch_ranks / parent_main.
Behaviour is chosen by STUB_MODE: ok | fail_rank1 | too_few | fail_sharded_only | hang_rank1."""
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
if mode in ("fail_rank1", "hang_rank1") or (mode == "fail_sharded_only" and "replicas" not in argv):
    time.sleep(60)  # rank 0 waits in a "collective" until the launcher terminates it
if rank == 0:
    err = argv[argv.index("--sharded-error") + 1] if "--sharded-error" in argv else None
    print(json.dumps({"n_gpus": world, "argv": argv, "scaling": "weak" if "replicas" in argv else "strong",
                      "sharded_error": err}), flush=True)
