mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider -x -k "grouped or group_and or verify_mode or per_image_filters or fgs_edge or inference_mode or set_exemplar_then or warpnet_stages" > gpurun_out/pytest_sel.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/pytest_sel.log
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-speed-leg --refs 0 --clips 0 --other-steps 0 --no-parity"
for i in 1 2; do
  DVC_GROUP_HEADS=1 timeout 300 $B > gpurun_out/ab_g1_$i.json 2> gpurun_out/ab_g1_$i.err; echo "g1 rc=$?"
  DVC_GROUP_HEADS=0 timeout 300 $B > gpurun_out/ab_g0_$i.json 2> gpurun_out/ab_g0_$i.err; echo "g0 rc=$?"
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/ab_g*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, d["value"], d["config"]["per_frame_api_frames_per_s"], d["config"]["dropin_unmodified_frames_per_s"])
    except Exception as e: print(f, "ERR", e)
PY
bash tools/busy_probe.sh 2>&1 | tail -3
