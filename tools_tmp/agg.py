import csv,sys,glob,collections
f=sorted(glob.glob(sys.argv[1]))[-1]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
idx=[i for i,r in enumerate(rows) if 'fri_pow_kernel' in r['Kernel_Name']]
a,b=idx[-2],idx[-1]
seg=rows[a+1:b+1]
agg=collections.Counter(); cnt=collections.Counter()
for r in seg:
    k=r['Kernel_Name'].split('(')[0][:48]
    agg[k]+=int(r['End_Timestamp'])-int(r['Start_Timestamp']); cnt[k]+=1
tot=sum(agg.values())
for k,v in agg.most_common(14): print("%-50s %4d launches %8.1f us %5.1f %%"%(k,cnt[k],v/1e3,100*v/tot))
print("busy %.1f us span %.1f us"%(tot/1e3,(int(seg[-1]['End_Timestamp'])-int(seg[0]['Start_Timestamp']))/1e3))
