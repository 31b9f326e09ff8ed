set -o pipefail
B="python bench.py --steps 5 --warmup 2 --no-fm --no-e2e --no-exact --no-dm --cpu-sample 0 --no-fnv"
python -m pytest tests/test_suffix_sort_gpu.py tests/test_ref_pins_golden.py tests/test_primitives_gpu.py tests/test_multi_abi.py -m gpu -x -q > gpurun_out/t4.log 2>&1; echo "rc=$?" >> gpurun_out/t4.log
grep -q "rc=0" gpurun_out/t4.log || { tail -30 gpurun_out/t4.log; exit 1; }
$B > gpurun_out/ab_new.json 2> gpurun_out/ab_new.err
KISS_HIP_EMIT_KEYS=1 $B > gpurun_out/ab_emitkeys.json 2> gpurun_out/ab_emitkeys.err
KISS_HIP_NO_FC0_ONEPASS=1 $B > gpurun_out/ab_nofc0.json 2> gpurun_out/ab_nofc0.err
$B --multi-abi 0 > gpurun_out/ab_multi1.json 2> gpurun_out/ab_multi1.err
$B --multi-abi 0,0 > gpurun_out/ab_multi2.json 2> gpurun_out/ab_multi2.err
python bench.py --steps 3 --warmup 1 --force-sharded --sharded-timings --no-fnv --cpu-sample 0 > gpurun_out/ab_sharded.json 2> gpurun_out/ab_sharded.err
python bench.py --steps 3 --warmup 1 --force-sharded --no-fnv --cpu-sample 0 > gpurun_out/ab_sharded_plain.json 2> gpurun_out/ab_sharded_plain.err
tail -2 gpurun_out/t4.log
