import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows=[r for r in rows if 'rk::' in r['Kernel_Name']]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# last fit: take the last 21*? kernels... find iterations by cd_mfma_kernel<2 (H solve) occurrences
idx=[i for i,r in enumerate(rows) if 'cd_mfma_kernel<' in r['Kernel_Name']]
print("H solves:",len(idx))
# analyse the last 10 iterations of the final fit
sel=idx[-11:]
for a,b in zip(sel[:-1],sel[1:]):
    seg=rows[a:b]
    busy=sum(int(r['End_Timestamp'])-int(r['Start_Timestamp']) for r in seg)
    wall=int(rows[b]['Start_Timestamp'])-int(rows[a]['Start_Timestamp'])
    gaps=[(int(seg[i+1]['Start_Timestamp'])-int(seg[i]['End_Timestamp']))/1e3 for i in range(len(seg)-1)]
    gaps.append((int(rows[b]['Start_Timestamp'])-int(seg[-1]['End_Timestamp']))/1e3)
    big=sorted([(g,seg[i]['Kernel_Name'][:40]) for i,g in enumerate(gaps)],reverse=True)[:3]
    print("iter wall %.1f us busy %.1f us kernels %d; biggest gaps after: %s"%(wall/1e3,busy/1e3,len(seg),big))
