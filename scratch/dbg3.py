import sys, numpy as np
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import torch
from gpu_util import make_pair, step_both
mode=sys.argv[1]
for n in [4096, 4133, 8192+5, 2048, 2049, 1000, 65536]:
    rng=np.random.default_rng(1)
    env,orc=make_pair('lander3d',n,mode,seed=5)
    z=np.zeros((3,n),np.float32); env.reset(options={"forces":z}); orc.reset(forces=z)
    prev=-rng.random(n)*300
    orc.prev_shaping[:]=prev.astype(orc.T)
    env.set_state(prev_shaping=orc.prev_shaping.astype(np.float64))
    a=np.full((n,4),0.0165,np.float32)
    got,want,_=step_both(env,orc,a)
    dr=np.abs(got[1].astype(np.float64)-want[1]); bad=np.flatnonzero(dr>1e-3)
    print(mode,'n',n,'bad',len(bad), (bad.min(),bad.max()) if len(bad) else '')
    env.close()
