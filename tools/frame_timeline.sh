#!/bin/bash
# Where one frame of the per-frame loop goes, launch by launch: rocprofv3 kernel trace of `bench.py --lookahead 0`, the frames cut at
# the correlation kernel, then per position in the frame's launch sequence the median duration and the median gap to the launch before.
export TMPDIR=/tmp; mkdir -p gpurun_out; rm -rf gpurun_out/ftl
timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/ftl -o t -- python bench.py --lookahead 0 --steps 40 --warmup 5 --no-cpu-baseline --no-speed-leg --other-steps 0 > gpurun_out/ftl_bench.json 2> gpurun_out/ftl.err
python - <<'PY'
import csv, glob, re, statistics as st
f = glob.glob("gpurun_out/ftl/**/t_kernel_trace.csv", recursive=True)[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0), int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", 1)) or 1)) for r in csv.DictReader(open(f))]
rows.sort()
idx = [i for i, r in enumerate(rows) if "corr_fwd_kernel" in r[2]]
# frames = runs between consecutive correlation launches with the modal length
lens = [b - a for a, b in zip(idx, idx[1:])]
L = st.mode(lens)
frames = [rows[a:b] for a, b in zip(idx, idx[1:]) if b - a == L]
# keep frames whose kernel-name sequence is the modal one
key = lambda fr: tuple(r[2] for r in fr)
seqs = {}
for fr in frames:
    seqs.setdefault(key(fr), []).append(fr)
fr_list = max(seqs.values(), key=len)
print(f"{len(fr_list)} frames of {L} launches (cut at corr_fwd)")
def short(n):
    n = re.sub(r"^void ", "", n)
    n = re.sub(r"\(.*$", "", n)
    return n[:58]
tot = 0
for p in range(L):
    d = st.median(fr[p][1] - fr[p][0] for fr in fr_list) / 1e3
    g = st.median((fr[p][0] - fr[p - 1][1]) for fr in fr_list) / 1e3 if p else 0.0
    wg = fr_list[0][p][3] // max(fr_list[0][p][4], 1)
    tot += d + max(g, 0)
    print(f"{p:3d} {short(fr_list[0][p][2]):58s} wgs {wg:6d} x{fr_list[0][p][4]:4d}  {d:7.1f} us  gap {g:5.1f}   cum {tot:7.1f}")
PY
