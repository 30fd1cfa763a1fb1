"""rocprofv3 counter_collection.csv -> per-kernel mean of every counter + mean duration."""
import csv, collections, sys
rows = list(csv.DictReader(open(sys.argv[1])))
pat = sys.argv[2] if len(sys.argv) > 2 else ""
agg = collections.defaultdict(lambda: collections.defaultdict(list)); dur = collections.defaultdict(list)
for r in rows:
    k = r["Kernel_Name"]
    if pat not in k: continue
    key = k.split("(")[0][-40:]
    agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
    dur[key].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for key, c in agg.items():
    d = sum(dur[key]) / len(dur[key]) / 1e3
    print(f"{key}  launches={len(dur[key])}  mean_duration_us={d:.1f}")
    for n, v in sorted(c.items()):
        print(f"    {n:32s} {sum(v) / len(v):.6g}")
