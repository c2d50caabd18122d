import csv,glob,sys,collections,re
import numpy as np
for d in sys.argv[1:]:
    f=glob.glob(d+'/**/*kernel_trace.csv',recursive=True)[0]
    rows=list(csv.DictReader(open(f)))
    by=collections.defaultdict(list)
    for r in rows: by[re.sub(r'\(anonymous namespace\)::','',r['Kernel_Name']).split('(')[0][-60:]].append((int(r['Start_Timestamp']),int(r['End_Timestamp']),r.get('Queue_Id'),r.get('Stream_Id')))
    print(d)
    for k,v in by.items():
        if 'serve' in k:
            v.sort(); dur=np.array([e-s for s,e,_,_ in v]); st=np.array([s for s,_,_,_ in v])
            gaps=np.diff(st)
            print('  %-60s n=%5d dur med %8.0f ns  start-to-start med %8.0f ns  queues %s streams %s'%(k,len(v),np.median(dur),np.median(gaps) if len(gaps) else 0,sorted(set(q for _,_,q,_ in v))[:4],sorted(set(s for _,_,_,s in v))[:4]))
    # what else ran while the LAST serve_kernel dispatch was alive?
    sk=[(int(r['Start_Timestamp']),int(r['End_Timestamp'])) for r in rows if 'serve_kernel' in r['Kernel_Name']]
    if sk:
        s0,e0=sk[len(sk)//2]
        cnt=collections.Counter()
        for r in rows:
            s,e=int(r['Start_Timestamp']),int(r['End_Timestamp'])
            if e>s0 and s<e0: cnt[re.sub(r'\(anonymous namespace\)::','',r['Kernel_Name']).split('(')[0][-70:]]+=1
        print('  concurrent with one serve_kernel dispatch (%d us):'%((e0-s0)//1000), dict(cnt))
