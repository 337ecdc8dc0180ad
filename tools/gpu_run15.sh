mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_corr_backward.py -m gpu -q --timeout 600 -p no:cacheprovider -x -s > gpurun_out/pytest_gpu15.log 2>&1; echo "pytest rc=$?"; tail -n 30 gpurun_out/pytest_gpu15.log
