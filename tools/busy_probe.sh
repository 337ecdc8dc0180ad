#!/bin/bash
# How busy is the GPU under the per-frame loop (one stream)?  rocprofv3 kernel trace of `bench.py --lookahead 0`, then the
# union of kernel intervals over the last second of the trace.
export TMPDIR=/tmp; mkdir -p gpurun_out; rm -rf gpurun_out/busy
timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/busy -o t -- python bench.py --lookahead 0 --steps 40 --warmup 5 --no-cpu-baseline --no-speed-leg > gpurun_out/busy_bench.json 2> gpurun_out/busy.err
python - <<'PY'
import csv, glob, json
f = glob.glob("gpurun_out/busy/**/t_kernel_trace.csv", recursive=True)[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))]
rows.sort()
# the timed region = the longest run of launches without a gap > 2 ms that contains >= 40 x 100 kernels: take the last 40 frames' worth
end = rows[-1][1]
# walk back over kernels until 40 corr_fwd launches of the per-frame pass are covered (skip the roofline leg: corr back to back)
sel = [r for r in rows]
# find windows: split at gaps > 3 ms
wins, cur = [], [sel[0]]
for a, b in zip(sel, sel[1:]):
    if b[0] - a[1] > 3_000_000:
        wins.append(cur); cur = []
    cur.append(b)
wins.append(cur)
best = None
for w in wins:
    nconv = sum("conv_wino" in k for _, _, k in w)
    ncorr = sum("corr_fwd" in k for _, _, k in w)
    if ncorr >= 38 and nconv > 30 * ncorr:
        best = w
if best is None:
    best = max(wins, key=len)
span = best[-1][1] - best[0][0]
busy = 0; cur_s, cur_e = best[0][0], best[0][1]
for s, e, _ in best[1:]:
    if s > cur_e:
        busy += cur_e - cur_s; cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
ncorr = sum("corr_fwd" in k for _, _, k in best)
print(f"window: {len(best)} kernels, {ncorr} frames, span {span/1e6:.2f} ms ({span/1e3/max(ncorr,1):.0f} us/frame), GPU busy {busy/span*100:.1f} %, "
      f"gaps {(span-busy)/1e3/max(ncorr,1):.0f} us/frame, {len(best)/max(ncorr,1):.0f} launches/frame")
print(open("gpurun_out/busy_bench.json").read()[:120])
PY
