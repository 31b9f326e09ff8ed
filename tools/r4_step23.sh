#!/bin/bash
# classify: LDS word window + partial rows: parity tests, then the headline
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_suffix_sort_gpu.py tests/test_verify_gpu.py tests/test_ref_pins_golden.py -m gpu -x -q > gpurun_out/r4_s23_tests.out 2>&1
rc=$?; tail -5 gpurun_out/r4_s23_tests.out
[ $rc -ne 0 ] && exit $rc
timeout -k 10 200 python bench.py --steps 10 --warmup 2 --cpu-sample 0 --no-e2e --no-fm --no-exact --no-dm --no-sensitivity > gpurun_out/r4_s23_bench.json 2> gpurun_out/r4_s23_bench.err || { tail -5 gpurun_out/r4_s23_bench.err; exit 1; }
python3 - <<'PY'
import json
j=json.loads(open("gpurun_out/r4_s23_bench.json").read().strip().splitlines()[-1])
print(j["ms_per_step"], j["roofline"]["kernel_ms_per_step"], j.get("sa_matches_pinned_hash"), j.get("verified"))
PY
