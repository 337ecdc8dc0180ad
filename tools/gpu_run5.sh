mkdir -p gpurun_out; export TMPDIR=/tmp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 tools/hwprobe/buffer_lds_oob.hip -o /tmp/buffer_lds_oob 2>/dev/null && /tmp/buffer_lds_oob > gpurun_out/buffer_lds_oob.txt 2>&1; cat gpurun_out/buffer_lds_oob.txt
timeout 300 python tools/corr_ab_probe.py > gpurun_out/corr_ab.txt 2>&1; echo "corr rc=$?"; cat gpurun_out/corr_ab.txt
