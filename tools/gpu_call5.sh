#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
rm -f gpurun_out/test_report.txt
timeout 900 python -m pytest tests -m gpu -q -x -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -8 gpurun_out/pytest_gpu.log
for fb in 1 2 3 4; do
  timeout 300 python bench.py --steps 48 --warmup 6 --no-cpu-baseline --front-batch $fb > gpurun_out/bench_fb$fb.json 2> gpurun_out/bench_fb$fb.err; echo "bench fb=$fb rc=$?"
  python -c "import json; d=json.load(open('gpurun_out/bench_fb$fb.json')); print(d['value'], d['ms_per_step'], d['config']['per_frame_api_frames_per_s'])"
done
