import csv,sys,glob
f=sorted(glob.glob(sys.argv[1]))[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
idx=[i for i,r in enumerate(rows) if 'fri_pow_kernel' in r['Kernel_Name']]
a,b=idx[-2],idx[-1]
seg=rows[a+1:b+1]
# proof starts after the gather/copy tail of the previous proof: find first ntt_small or from_values kernel
t0=int(seg[0]['Start_Timestamp']); prev_end=None; tot=0
for r in seg:
    s=int(r['Start_Timestamp']); e=int(r['End_Timestamp'])
    gap=(s-prev_end)/1e3 if prev_end else 0
    tot+=e-s
    if (e-s)/1e3>=float(sys.argv[2]) or gap>=float(sys.argv[2]):
        print("%8.1f us  gap %6.1f  dur %6.1f  %s grid %s"%((s-t0)/1e3,gap,(e-s)/1e3,r['Kernel_Name'][:64],r.get('Grid_Size_X','')))
    prev_end=e
print("kernels",len(seg),"busy us",tot/1e3,"span us",(int(seg[-1]['End_Timestamp'])-t0)/1e3)
