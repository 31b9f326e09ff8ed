set -o pipefail
B="python bench.py --steps 8 --warmup 3 --no-fm --no-e2e --no-exact --no-dm --cpu-sample 0 --no-fnv"
KISS_HIP_FC0_HALF=1 python -m pytest tests/test_ref_pins_golden.py tests/test_multi_abi.py -m gpu -x -q > gpurun_out/t12.log 2>&1; echo "rc=$?" >> gpurun_out/t12.log
grep -q "rc=0" gpurun_out/t12.log || { tail -30 gpurun_out/t12.log; exit 1; }
$B > gpurun_out/ab5_a1.json 2> gpurun_out/ab5_a1.err
KISS_HIP_FC0_HALF=1 $B > gpurun_out/ab5_h1.json 2> gpurun_out/ab5_h1.err
$B > gpurun_out/ab5_a2.json 2> gpurun_out/ab5_a2.err
KISS_HIP_FC0_HALF=1 $B > gpurun_out/ab5_h2.json 2> gpurun_out/ab5_h2.err
tail -n 2 gpurun_out/t12.log
