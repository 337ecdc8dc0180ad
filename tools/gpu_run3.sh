mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider -k "stream_k or e2e or nccl or cli or corr_ or conv2d_split" > gpurun_out/pytest_gpu2.log 2>&1; echo "pytest rc=$?"; tail -n 15 gpurun_out/pytest_gpu2.log
timeout 300 python tools/conv_dma_probe.py > gpurun_out/conv_dma_probe.txt 2>&1; cat gpurun_out/conv_dma_probe.txt
timeout 300 python tools/conv_dispatch_probe.py > gpurun_out/conv_dispatch_probe.txt 2>&1; cat gpurun_out/conv_dispatch_probe.txt
