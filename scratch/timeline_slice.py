"""rocprofv3 kernel_trace.csv -> a slice of the device timeline (start, duration, queue, kernel) from the middle of the run,
plus the fraction of wall time in which 0 / 1 / 2+ kernels were running.  usage: timeline_slice.py <kernel_trace.csv> [n]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 70
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?"), r["Kernel_Name"]) for r in rows))
mid = len(ev) * 2 // 3
t0 = ev[mid][0]
def short(k):
    k = k.split("(")[0].replace("void ", "").replace("ed::", "")
    return k[-46:]
for s, e, q, k in ev[mid:mid + n]:
    print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:7.1f}  q{q:>3s}  {short(k)}")
# concurrency histogram over the slice [ev[mid].start, ev[mid+2000].end]
sl = ev[mid:mid + 4000]
pts = sorted([(s, 1) for s, e, _, _ in sl] + [(e, -1) for s, e, _, _ in sl])
cur, last, acc = 0, pts[0][0], {}
for t, d in pts:
    acc[min(cur, 3)] = acc.get(min(cur, 3), 0) + (t - last)
    cur += d
    last = t
tot = sum(acc.values())
print("concurrency (kernels in flight: share of wall time):", {k: round(v / tot, 3) for k, v in sorted(acc.items())})
