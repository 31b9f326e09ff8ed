set -o pipefail
F="python bench.py --steps 2 --warmup 1 --text-len 50000000 --no-e2e --no-exact --no-dm --cpu-sample 0 --no-fnv --no-verify"
for v in 64 128 256 512 1024 4096; do KISS_HIP_FM_HEAVY=$v $F > gpurun_out/tune3_fmh$v.json 2> gpurun_out/tune3_fmh$v.err; done
X="python bench.py --steps 3 --warmup 1 --algo prefix_doubling --no-fm --no-e2e --no-dm --cpu-sample 0 --no-fnv"
for v in 128 256 512 1024; do KISS_HIP_DOUBLING_H0=$v $X > gpurun_out/tune3_h0_$v.json 2> gpurun_out/tune3_h0_$v.err; done
B="python bench.py --steps 6 --warmup 3 --no-fm --no-e2e --no-exact --no-dm --cpu-sample 0 --no-fnv --no-verify"
for v in 2048 32768 131072; do KISS_HIP_INDUCE_SMALL_MAX=$v $B > gpurun_out/tune3_sm$v.json 2> gpurun_out/tune3_sm$v.err; done
$B > gpurun_out/tune3_base.json 2> gpurun_out/tune3_base.err
echo done
