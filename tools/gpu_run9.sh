mkdir -p gpurun_out; export TMPDIR=/tmp; cd /tmp
rm -rf /tmp/ctrace; timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/ctrace -o t -- python $GRAFT_REPO_ROOT/tools/conv_trace_probe.py > /tmp/ctrace.log 2>&1
f=$(find /tmp/ctrace -name "*kernel_trace.csv" | head -1); echo $f; true
python - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
out = []
prev_end = None
seq = []
for r in rows:
    n = r["Kernel_Name"]
    short = "sk" if "conv_sk_kernel" in n else "fix" if "fixup" in n else "old" if "conv_mfma" in n else "red" if "splitk" in n else None
    if short is None:
        prev_end = None
        continue
    st, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    seq.append((short, n[:60], (en - st) / 1e3, None if prev_end is None else (st - prev_end) / 1e3, r.get("Grid_Size_X"), r.get("Workgroup_Size_X")))
    prev_end = en
# group consecutive identical patterns
import itertools
i = 0
stats = collections.OrderedDict()
for s in seq:
    key = (s[0], s[1], s[4])
    d = stats.setdefault(key, {"dur": [], "gap": []})
    d["dur"].append(s[2])
    if s[3] is not None: d["gap"].append(s[3])
for k, d in stats.items():
    dur = sorted(d["dur"]); gap = sorted(d["gap"]) or [0]
    print(f"{k[0]:4s} grid {k[2]:>8s} n={len(dur):3d} dur med {dur[len(dur)//2]:7.1f} us min {dur[0]:7.1f} | gap before med {gap[len(gap)//2]:5.1f} us  {k[1]}")
PY
