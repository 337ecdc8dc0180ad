mkdir -p gpurun_out; export TMPDIR=/tmp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 tools/hwprobe/mfma_f32_chain.hip -o /tmp/mfma_f32_chain 2>/dev/null && /tmp/mfma_f32_chain > gpurun_out/mfma_f32_chain.txt 2>&1; cat gpurun_out/mfma_f32_chain.txt
