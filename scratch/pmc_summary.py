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
    g=sum(c['GRBM_GUI_ACTIVE'])/len(c['GRBM_GUI_ACTIVE'])/8 if 'GRBM_GUI_ACTIVE' in c else 0
    mf=sum(c['SQ_VALU_MFMA_BUSY_CYCLES'])/len(c['SQ_VALU_MFMA_BUSY_CYCLES']) if 'SQ_VALU_MFMA_BUSY_CYCLES' in c else 0
    print(f"{key[0]:30s} grid={key[1]:>8s} dur={d:8.1f}us clk={g/d/1e3:5.2f}GHz mfma_busy={mf/(g*1024) if g else 0:6.3f}")
