mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider -x -k "stream_k or e2e or nccl or cli or corr_ or conv2d_split" > gpurun_out/pytest_gpu2.log 2>&1; echo "pytest rc=$?"; tail -n 15 gpurun_out/pytest_gpu2.log
timeout 300 python tools/corr_ab_probe.py > gpurun_out/corr_ab.txt 2>&1; echo "corr rc=$?"; cat gpurun_out/corr_ab.txt
timeout 600 python tools/conv_sk_sweep.py > gpurun_out/conv_sk_sweep.txt 2>&1; echo "sweep rc=$?"; cat gpurun_out/conv_sk_sweep.txt
