set -o pipefail
X="python bench.py --steps 3 --warmup 1 --algo prefix_doubling --no-fm --no-e2e --no-dm --cpu-sample 0 --no-fnv"
$X > gpurun_out/ab8_a1.json 2> gpurun_out/ab8_a1.err
KISS_HIP_ISA_ONE_LEVEL=1 $X > gpurun_out/ab8_o1.json 2> gpurun_out/ab8_o1.err
$X > gpurun_out/ab8_a2.json 2> gpurun_out/ab8_a2.err
KISS_HIP_ISA_ONE_LEVEL=1 $X > gpurun_out/ab8_o2.json 2> gpurun_out/ab8_o2.err
echo done
