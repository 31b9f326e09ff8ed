set -o pipefail
python -m pytest tests/test_fm_gpu.py tests/test_fuzzers_gpu.py -m gpu -x -q > gpurun_out/t6.log 2>&1; echo "rc=$?" >> gpurun_out/t6.log
B="python bench.py --steps 3 --warmup 1 --text-len 50000000 --no-e2e --no-exact --no-dm --cpu-sample 0 --no-fnv --no-verify"
for L in 4 8 16 32 64 256; do KISS_HIP_FM_LIGHT=$L $B > gpurun_out/fm_light_$L.json 2> gpurun_out/fm_light_$L.err; done
tail -n 3 gpurun_out/t6.log
