import csv,sys,collections
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
per=collections.defaultdict(list)
for r in rows:
    n=r['Kernel_Name']
    if 'blasr_hip' not in n: continue
    per[n.split('(')[0][:70]].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6)
tot=[0,0,0]
for n,v in sorted(per.items(), key=lambda kv:-sum(kv[1])):
    if len(v)%3==0 and len(v)>=9:
        k=len(v)//3  # calls per step
        lv=[sum(v[i] for i in range(len(v)) if (i%k)*3//k==L)/3 for L in range(3)] if k%3==0 else None
        if lv:
            for i in range(3): tot[i]+=lv[i]
            print("%-72s L0 %7.2f L1 %7.2f L2 %7.2f"%(n,lv[0],lv[1],lv[2]))
            continue
    print("%-72s calls %d per-step %.2f"%(n,len(v),sum(v)/3))
print("level totals",tot)
