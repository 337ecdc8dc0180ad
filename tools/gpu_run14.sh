mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider -x -k "corr or warpnet or e2e or config" > gpurun_out/pytest_gpu14.log 2>&1; echo "pytest rc=$?"; tail -n 6 gpurun_out/pytest_gpu14.log
timeout 300 python tools/corr_ab_probe.py > gpurun_out/corr_ab.txt 2>&1; cat gpurun_out/corr_ab.txt
