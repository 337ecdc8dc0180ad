#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k winograd -p no:cacheprovider > gpurun_out/pytest_wino.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/pytest_wino.log
timeout 400 python tools/conv_algo_sweep.py > gpurun_out/conv_algo_sweep.txt 2>&1; echo "sweep rc=$?"; cut -c1-150 gpurun_out/conv_algo_sweep.txt
timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline > gpurun_out/bench_auto.json 2> gpurun_out/bench_auto.err; echo "bench rc=$?"; cut -c1-400 gpurun_out/bench_auto.json
rm -rf gpurun_out/pmc2
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc2 -o p -- python tools/prof_kernels.py > gpurun_out/pmc2.log 2>&1; echo "pmc2 rc=$?"
