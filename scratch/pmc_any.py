import csv, collections, sys
rows=list(csv.DictReader(open(sys.argv[1])))
agg=collections.defaultdict(lambda: collections.defaultdict(list)); dur=collections.defaultdict(list)
for r in rows:
    k=r['Kernel_Name']
    if 'gemm' not in k: continue
    key=(k.split('(')[0][-28:], r['Grid_Size'])
    agg[key][r['Counter_Name']].append(float(r['Counter_Value']))
    dur[key].append(int(r['End_Timestamp'])-int(r['Start_Timestamp']))
for key,c in agg.items():
    d=sum(dur[key])/len(dur[key])/1e3
    print(f"{key[0]} grid={key[1]} dur={d:.1f}us")
    for n,v in sorted(c.items()): print(f"    {n:36s} {sum(v)/len(v):.5g}")
