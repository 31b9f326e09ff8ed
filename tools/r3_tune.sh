set -o pipefail
B="python bench.py --steps 6 --warmup 3 --no-fm --no-e2e --no-exact --no-dm --cpu-sample 0 --no-fnv --no-verify"
$B > gpurun_out/tune_base.json 2> gpurun_out/tune_base.err
for v in 16 32 48; do KISS_HIP_SMALL_SEG=$v $B > gpurun_out/tune_seg$v.json 2> gpurun_out/tune_seg$v.err; done
for v in 524288 1048576 4194304 8388608; do KISS_HIP_COLLAPSE_N=$v $B > gpurun_out/tune_col$v.json 2> gpurun_out/tune_col$v.err; done
for v in 2 4; do KISS_HIP_PIVOT_SLOTS=$v $B > gpurun_out/tune_slots$v.json 2> gpurun_out/tune_slots$v.err; done
$B > gpurun_out/tune_base2.json 2> gpurun_out/tune_base2.err
echo done
