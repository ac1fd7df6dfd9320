import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
nd=[r for r in rows if 'k_nd_' in r['Kernel_Name']]
nd.sort(key=lambda r:int(r['Start_Timestamp']))
groups=[];cur=[];prev=None
for r in nd:
    fwd = 'fwd' in r['Kernel_Name'] or 'forward' in r['Kernel_Name']
    if prev is not None and fwd and not prev:
        groups.append(cur);cur=[]
    cur.append(r);prev=fwd
groups.append(cur)
g=groups[-2]
t0=int(g[0]['Start_Timestamp'])
tot=0
import re,collections
agg=collections.defaultdict(float)
for r in g:
    d=int(r['End_Timestamp'])-int(r['Start_Timestamp']);tot+=d
    nm=re.sub(r'void \(anonymous namespace\)::','',r['Kernel_Name'])[:40]
    agg[nm.split('(')[0]]+=d/1e3
    print("%8.1f %7.1f us  wgs %6d x %4s  %s"%((int(r['Start_Timestamp'])-t0)/1e3,d/1e3,int(r['Grid_Size_X'])//int(r['Workgroup_Size_X']),r['Workgroup_Size_X'],nm))
print(len(g),'launches, kernel time',tot/1e3,'span',(int(g[-1]['End_Timestamp'])-t0)/1e3)
for k,v in sorted(agg.items(),key=lambda x:-x[1]): print("%9.1f us %s"%(v,k))
