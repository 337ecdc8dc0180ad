mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider -x -k "stream_k" > gpurun_out/pytest_gpu4.log 2>&1; echo "pytest rc=$?"; tail -n 8 gpurun_out/pytest_gpu4.log
SWEEP_REPS=8 timeout 600 python tools/conv_sk_sweep.py > gpurun_out/conv_sk_sweep.txt 2>&1; echo "sweep rc=$?"; cat gpurun_out/conv_sk_sweep.txt
