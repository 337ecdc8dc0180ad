#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
rm -f gpurun_out/test_report.txt
timeout 900 python -m pytest tests -m gpu -q -x -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -8 gpurun_out/pytest_gpu.log
timeout 300 python bench.py --steps 48 --warmup 6 --no-cpu-baseline > gpurun_out/bench_auto.json 2> gpurun_out/bench_auto.err; echo "bench rc=$?"
python -c "import json; d=json.load(open('gpurun_out/bench_auto.json')); print(d['value'], d['ms_per_step'], d['config']['per_frame_api_frames_per_s'], d['roofline']['frac'], d['roofline']['avg_launch_us'])"
timeout 200 python tools/corr_temperature_probe.py 2>&1 | tail -8
