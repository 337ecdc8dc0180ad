"""Reads a rocprofv3 --kernel-trace CSV of `tools/graph_overlap_probe.py trace`: which queues the kernels ran on and how
much of the busy time had kernels of more than one queue in flight."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ev = []
q = collections.Counter()
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    qid = r.get("Queue_Id", "?")
    q[qid] += 1
    ev.append((s, 1, qid)); ev.append((e, -1, qid))
ev.sort()
active = collections.Counter()
last = None
busy = multi = 0
for t, d, qid in ev:
    if last is not None:
        n = sum(1 for v in active.values() if v > 0)
        if n >= 1: busy += t - last
        if n >= 2: multi += t - last
    active[qid] += d
    last = t
print(f"{len(rows)} kernels on queues {dict(q)}; busy {busy / 1e6:.2f} ms, of which >= 2 queues active {multi / 1e6:.2f} ms ({100.0 * multi / max(busy, 1):.1f} %)")
