set -o pipefail
B="python bench.py --steps 8 --warmup 3 --no-fm --no-e2e --no-exact --no-dm --cpu-sample 0 --no-fnv"
python -m pytest tests/test_suffix_sort_gpu.py tests/test_ref_pins_golden.py tests/test_multi_abi.py tests/test_fuzzers_gpu.py -m gpu -x -q > gpurun_out/t13.log 2>&1; echo "rc=$?" >> gpurun_out/t13.log
grep -q "rc=0" gpurun_out/t13.log || { tail -30 gpurun_out/t13.log; exit 1; }
$B > gpurun_out/ab6_a1.json 2> gpurun_out/ab6_a1.err
KISS_HIP_FC0_SMALL_TILES=1 $B > gpurun_out/ab6_s1.json 2> gpurun_out/ab6_s1.err
$B > gpurun_out/ab6_a2.json 2> gpurun_out/ab6_a2.err
KISS_HIP_FC0_SMALL_TILES=1 $B > gpurun_out/ab6_s2.json 2> gpurun_out/ab6_s2.err
tail -n 2 gpurun_out/t13.log
