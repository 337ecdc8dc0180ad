mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 300 python tools/corr_ab_probe.py > gpurun_out/corr_ab.txt 2>&1; cat gpurun_out/corr_ab.txt
