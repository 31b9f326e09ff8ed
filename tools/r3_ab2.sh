set -o pipefail
B="python bench.py --steps 6 --warmup 2 --no-fm --no-e2e --no-exact --no-dm --cpu-sample 0 --no-fnv"
python -m pytest tests/test_suffix_sort_gpu.py tests/test_ref_pins_golden.py tests/test_multi_abi.py tests/test_cli_gpu.py tests/test_fuzzers_gpu.py -m gpu -x -q > gpurun_out/t9.log 2>&1; echo "rc=$?" >> gpurun_out/t9.log
grep -q "rc=0" gpurun_out/t9.log || { tail -30 gpurun_out/t9.log; exit 1; }
$B > gpurun_out/ab2_new.json 2> gpurun_out/ab2_new.err
KISS_HIP_PAIR_KEYS=1 $B > gpurun_out/ab2_pairkeys.json 2> gpurun_out/ab2_pairkeys.err
$B > gpurun_out/ab2_new2.json 2> gpurun_out/ab2_new2.err
tail -n 2 gpurun_out/t9.log
