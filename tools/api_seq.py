import csv,sys
d=sys.argv[1]
rows=list(csv.DictReader(open(d+'_hip_api_trace.csv')))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
idx=[i for i,r in enumerate(rows) if r['Function']=='hipMemGetInfo']
i0=idx[-1]
t0=int(rows[i0]['Start_Timestamp'])
last=None;cnt=0;tl=0
for r in rows[i0:]:
    f=r['Function']
    if f.startswith('__hip') or f in('hipFree','hipGetDevice'): continue
    s=(int(r['Start_Timestamp'])-t0)/1e6; d_=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3
    if s>40: break
    if f=='hipLaunchKernel':
        cnt+=1; tl+=d_
        if last!='L': first=s
        last='L'; lastend=s; continue
    if last=='L': print('   ... %d launches %.3f .. %.3f ms (in the calls: %.0f us)'%(cnt,first,lastend,tl)); cnt=0; tl=0
    last=f
    print('%8.3f ms %-28s %8.1f us'%(s,f,d_))
ks=list(csv.DictReader(open(d+'_kernel_trace.csv')))
ks=[(int(r['Start_Timestamp'])-t0,int(r['End_Timestamp'])-t0) for r in ks if int(r['Start_Timestamp'])>=t0]
print('kernels after t0: %d, first start %.3f ms, last end %.3f ms, sum %.3f ms'%(len(ks),min(a for a,b in ks)/1e6,max(b for a,b in ks)/1e6,sum(b-a for a,b in ks)/1e6))
