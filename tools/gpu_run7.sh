mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 300 python tools/conv_sk_parts_probe.py > gpurun_out/conv_sk_parts.txt 2>&1; cat gpurun_out/conv_sk_parts.txt
