#!/bin/bash
# one GPU-box call: parity suite, default bench line, the C3 / C4 workloads, and the torchrun (N = 1) launch path
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r1_pytest_gpu.log
python bench.py > gpurun_out/r1_bench.json 2> gpurun_out/r1_bench.err
timeout 600 python bench.py --workload C3 --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-events > gpurun_out/r1_bench_C3.json 2> gpurun_out/r1_bench_C3.err
timeout 600 python bench.py --workload C4 --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-events > gpurun_out/r1_bench_C4.json 2> gpurun_out/r1_bench_C4.err
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-events > gpurun_out/r1_bench_torchrun1.json 2> gpurun_out/r1_bench_torchrun1.err
tail -3 gpurun_out/r1_pytest_gpu.log
cut -c1-300 gpurun_out/r1_bench.json
cut -c1-300 gpurun_out/r1_bench_C3.json; tail -3 gpurun_out/r1_bench_C3.err
cut -c1-300 gpurun_out/r1_bench_C4.json; tail -3 gpurun_out/r1_bench_C4.err
cut -c1-300 gpurun_out/r1_bench_torchrun1.json; tail -3 gpurun_out/r1_bench_torchrun1.err
