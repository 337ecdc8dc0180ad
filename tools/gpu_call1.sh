#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 400 python tools/conv_algo_sweep.py > gpurun_out/conv_algo_sweep.txt 2>&1; echo "sweep rc=$?"
timeout 200 python tools/_wino_dbg.py > gpurun_out/wino_dbg.txt 2>&1; echo "dbg rc=$?"
for a in auto direct winograd; do
  DVC_CONV_ALGO=$a timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline > gpurun_out/bench_$a.json 2> gpurun_out/bench_$a.err; echo "bench $a rc=$?"
done
bash tools/gpu_check.sh tests
