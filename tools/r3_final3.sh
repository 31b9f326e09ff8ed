set -o pipefail
python tools/fuzz_parity.py 90 305 > gpurun_out/fuzz_parity_final3.log 2>&1; echo "fuzz_parity rc=$?"
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r3_final3.json 2> gpurun_out/r3_final3.err; echo "bench rc=$?"
bash tools/prof.sh r03c --steps 3 --warmup 1 --no-e2e --no-fm --no-exact --no-dm --cpu-sample 0 --profile-steps 0 --no-verify --no-fnv > gpurun_out/r03c_summary.txt 2>&1
bash tools/prof.sh r03d --algo prefix_doubling --k 4294967295 --steps 3 --warmup 1 --no-e2e --no-fm --no-exact --no-dm --no-fnv --cpu-sample 0 > gpurun_out/r03d_summary.txt 2>&1
tail -n 2 gpurun_out/fuzz_parity_final3.log
