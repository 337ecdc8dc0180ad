mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider -x -k "conv2d_ws or routes_direct or colorvidnet" > gpurun_out/pytest_sel.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/pytest_sel.log
timeout 400 python tools/conv_algo_sweep.py > gpurun_out/conv_algo_sweep.txt 2>&1; echo "sweep rc=$?"; grep -E "ws|per frame" gpurun_out/conv_algo_sweep.txt
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-speed-leg --refs 0 --clips 0 --other-steps 0"
for i in 1 2; do
  DVC_WS_CONV=1 timeout 300 $B > gpurun_out/ab_w1_$i.json 2> gpurun_out/ab_w1_$i.err; echo "w1 rc=$?"
  DVC_WS_CONV=0 timeout 300 $B > gpurun_out/ab_w0_$i.json 2> gpurun_out/ab_w0_$i.err; echo "w0 rc=$?"
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/ab_w*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, d["value"], d["config"]["per_frame_api_frames_per_s"], d["parity"]["gpu_over_cpu32"] if d.get("parity") else None)
    except Exception as e: print(f, "ERR", e)
PY
