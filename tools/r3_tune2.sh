set -o pipefail
B="python bench.py --steps 6 --warmup 3 --no-fm --no-e2e --no-exact --no-dm --cpu-sample 0 --no-fnv --no-verify"
for v in 16384 32768 65536 131072 262144 524288; do KISS_HIP_COLLAPSE_N=$v $B > gpurun_out/tune2_col$v.json 2> gpurun_out/tune2_col$v.err; done
KISS_HIP_COLLAPSE_N=131072 python tools/stress_verify.py 1000000000 allA,period7,long_runs > gpurun_out/tune2_stress.log 2>&1
echo done
