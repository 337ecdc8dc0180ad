mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider -k "nets or e2e or conv2d or cli or tail" > gpurun_out/pytest_gpu23.log 2>&1; echo "pytest rc=$?"; tail -n 8 gpurun_out/pytest_gpu23.log; grep "batch-of-2" gpurun_out/test_report.txt
