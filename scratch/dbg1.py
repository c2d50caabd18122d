import sys, numpy as np
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import torch
from gpu_util import make_pair, step_both
rng=np.random.default_rng(11)
for mode in ['float32','float32_rn','float64']:
    n=4096+37
    env,orc=make_pair('lander3d',n,mode,seed=5)
    env.reset(options={"forces": np.zeros((3,n),np.float32)}); orc.reset(forces=np.zeros((3,n)))
    x=rng.standard_normal((12,n))*np.array([4,2,4,2,6,2,.4,.5,.4,.5,2,1])[:,None]; x[4]-=6
    x[0,:64]=9.99+0.02*rng.random(64); x[6,64:128]=np.pi/4-1e-3+2e-3*rng.random(64); x[4,128:512]=np.abs(x[4,128:512])*0.01
    status=rng.integers(0,4,n).astype(np.uint8); steps=rng.integers(1,1002,n).astype(np.int32)
    steps[:16]=1000
    prev=-rng.random(n)*300; prev[::97]=np.nan; force=rng.uniform(-30,30,(3,n)); flags=(rng.random(n)<0.3).astype(np.uint8)
    orc.x[:]=orc._round(x); orc.status[:]=status; orc.steps[:]=steps; orc.prev_shaping[:]=prev.astype(orc.T); orc.force[:]=force.astype(orc.T); orc.pending[:]=flags.astype(bool)
    env.set_state(x=orc.x.astype(np.float64),status=status,steps=steps,prev_shaping=orc.prev_shaping.astype(np.float64),force=orc.force.astype(np.float64),flags=flags)
    a=rng.uniform(-0.5,1.5,(n,4)).astype(np.float32)
    a[::5]=(0.01656*(1+0.01*rng.standard_normal((len(a[::5]),4)))).astype(np.float32)
    got,want,_=step_both(env,orc,a)
    dr=np.abs(got[1].astype(np.float64)-want[1]); i=int(np.argmax(dr))
    st=env.get_state()
    print(mode,'max dr',dr.max(),'lane',i,'isnan prev',np.isnan(prev[i]),'flags',flags[i],'steps',steps[i],'got r',got[1][i],'want',want[1][i],'status0',status[i],'prev',prev[i], 'x gpu',st['x'][:,i],'x orc',orc.x[:,i].astype(float), 'ps gpu',st['prev_shaping'][i],'orc',orc.prev_shaping[i])
    print('  num bad', (dr>1e-3+1e-5*np.abs(want[1])).sum())
