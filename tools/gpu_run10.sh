mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider -x -k "stream_k and (54 or 55)" > gpurun_out/pytest_gpu10.log 2>&1; echo "pytest rc=$?"; tail -n 5 gpurun_out/pytest_gpu10.log
timeout 300 python tools/conv_sk_parts_probe.py > gpurun_out/conv_sk_parts.txt 2>&1; cat gpurun_out/conv_sk_parts.txt
