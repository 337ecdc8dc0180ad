#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 500 python tools/conv_wino_probe.py > gpurun_out/conv_wino_probe.txt 2>&1; echo "probe rc=$?"; tail -16 gpurun_out/conv_wino_probe.txt | cut -c1-260
bash tools/gpu_check.sh tests
