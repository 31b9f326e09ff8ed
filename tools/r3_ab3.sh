set -o pipefail
B="python bench.py --steps 8 --warmup 3 --no-fm --no-e2e --no-exact --no-dm --cpu-sample 0 --no-fnv"
KISS_HIP_WIDE_STORES=1 python -m pytest tests/test_primitives_gpu.py tests/test_ref_pins_golden.py tests/test_cli_gpu.py tests/test_multi_abi.py -m gpu -x -q > gpurun_out/t10.log 2>&1; echo "rc=$?" >> gpurun_out/t10.log
grep -q "rc=0" gpurun_out/t10.log || { tail -30 gpurun_out/t10.log; exit 1; }
$B > gpurun_out/ab3_a1.json 2> gpurun_out/ab3_a1.err
KISS_HIP_WIDE_STORES=1 $B > gpurun_out/ab3_w1.json 2> gpurun_out/ab3_w1.err
$B > gpurun_out/ab3_a2.json 2> gpurun_out/ab3_a2.err
KISS_HIP_WIDE_STORES=1 $B > gpurun_out/ab3_w2.json 2> gpurun_out/ab3_w2.err
python tools/cli_e2e.py 3117292070 --devices 0,0 > gpurun_out/r03_cli_e2e_two_shares.log 2>&1
tail -n 2 gpurun_out/t10.log
