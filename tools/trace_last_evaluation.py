import csv,glob,sys,collections
f=glob.glob(sys.argv[1]+'/**/*kernel_trace.csv', recursive=True)[0]
rows=sorted(csv.DictReader(open(f)), key=lambda r:int(r['Start_Timestamp']))
# last evaluation: from last kmat launch to end
idx=[i for i,r in enumerate(rows) if 'kmat' in r['Kernel_Name']]
# evaluations are separated by kmat launches; take the last full one
i0=idx[-1]
ev=rows[i0:]
t0=int(ev[0]['Start_Timestamp']); t1=max(int(r['End_Timestamp']) for r in ev)
print('launches',len(ev),'span us',(t1-t0)/1e3)
tot=collections.defaultdict(lambda:[0,0.0])
busy=[]
for r in ev:
    n=r['Kernel_Name'].split('(')[0][-40:]
    tot[n][0]+=1; tot[n][1]+=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3
    busy.append((int(r['Start_Timestamp']),int(r['End_Timestamp'])))
for n,(c,t) in sorted(tot.items(), key=lambda x:-x[1][1]): print(f"{n:42s} {c:4d} launches {t:9.1f} us  avg {t/c:7.1f}")
busy.sort(); cov=0; cur_s,cur_e=busy[0]
for s,e in busy[1:]:
    if s>cur_e: cov+=cur_e-cur_s; cur_s,cur_e=s,e
    else: cur_e=max(cur_e,e)
cov+=cur_e-cur_s
print('union of kernel time us',cov/1e3,'idle us',(t1-t0-cov)/1e3)
