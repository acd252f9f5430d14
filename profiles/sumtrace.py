import csv,sys,collections
rows=list(csv.DictReader(open(sys.argv[1])))
d=collections.defaultdict(list)
for r in rows:
    n=r['Kernel_Name']
    if 'rhs' in n:
        d[(n.split('(')[0][-50:], r.get('Grid_Size_X'), r.get('Workgroup_Size_X'), r.get('VGPR_Count'))].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
for k,v in d.items():
    v2=sorted(v)
    print(k, len(v), 'median %.1f us'%v2[len(v2)//2], 'min %.1f'%min(v))
