#!/bin/bash
# Round-end artefacts in one GPU-box round trip: bench lines (default, 432x768, bf16 correlation, torchrun single rank),
# per-layer direct-vs-Winograd sweep, rocprofv3 kernel stats of the bench command and the PMC passes.
mkdir -p gpurun_out; export TMPDIR=/tmp
bash tools/gpu_check.sh bench
timeout 300 python bench.py --steps 30 --warmup 5 --hw 432x768 > gpurun_out/bench_432x768.json 2> gpurun_out/bench_432x768.err; echo "bench 432x768 rc=$?"
timeout 300 python bench.py --steps 40 --warmup 5 --corr bf16 --no-cpu-baseline > gpurun_out/bench_bf16.json 2> gpurun_out/bench_bf16.err; echo "bench bf16 rc=$?"
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 40 --warmup 5 --no-cpu-baseline > gpurun_out/bench_torchrun1.json 2> gpurun_out/bench_torchrun1.err; echo "bench torchrun rc=$?"
timeout 400 python tools/conv_algo_sweep.py > gpurun_out/conv_algo_sweep.txt 2>&1; echo "sweep rc=$?"; tail -1 gpurun_out/conv_algo_sweep.txt
bash tools/gpu_check.sh prof > gpurun_out/prof_stdout.txt 2>&1; echo "prof done"
bash tools/gpu_check.sh pmc > gpurun_out/pmc_stdout.txt 2>&1; echo "pmc done"
