import sys, os, numpy as np
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import torch
from gpu_util import make_pair, step_both
HOVER = float(np.load("/root/repo/tests/golden/meta.npz")["hover_motor"])
task, mode = 'lander3d', sys.argv[1] if len(sys.argv)>1 else 'float32_rn'
rng = np.random.default_rng(11)
n = 4096 + 37
env, orc = make_pair(task, n, mode, seed=5)
env.reset(options={"forces": np.zeros((3, n), np.float32)})
orc.reset(forces=np.zeros((3, n)))
x = rng.standard_normal((12, n)) * np.array([4, 2, 4, 2, 6, 2, .4, .5, .4, .5, 2, 1])[:, None]
x[4] -= 6
x[0, :64] = 9.99 + 0.02 * rng.random(64)
x[6, 64:128] = np.pi / 4 - 1e-3 + 2e-3 * rng.random(64)
x[4, 128:512] = np.abs(x[4, 128:512]) * 0.01
status = rng.integers(0, 4, n).astype(np.uint8)
steps = rng.integers(1, 1002, n).astype(np.int32)
steps[:16] = 1000
prev = -rng.random(n) * 300
prev[::97] = np.nan
force = rng.uniform(-30, 30, (3, n))
flags = (rng.random(n) < 0.3).astype(np.uint8)
orc.x[:] = orc._round(x); orc.status[:] = status; orc.steps[:] = steps
orc.prev_shaping[:] = prev.astype(orc.T); orc.force[:] = force.astype(orc.T); orc.pending[:] = flags.astype(bool)
x_before = orc.x[:, :].astype(np.float64).copy()
env.set_state(x=orc.x.astype(np.float64), status=status, steps=steps, prev_shaping=orc.prev_shaping.astype(np.float64), force=orc.force.astype(np.float64), flags=flags)
st0 = env.get_state(); print('prev readback equal:', np.array_equal(np.nan_to_num(st0['prev_shaping'],nan=7.0), np.nan_to_num(orc.prev_shaping.astype(np.float64),nan=7.0)), 'x equal', np.array_equal(st0['x'], orc.x.astype(np.float64)))
actions = rng.uniform(-0.5, 1.5, (n, 4)).astype(np.float32)
actions[::5] = (HOVER * (1 + 0.01 * rng.standard_normal((len(actions[::5]), 4)))).astype(np.float32)
env._reward.fill_(12345.0); env._term.fill_(77); env._trunc.fill_(77); env._obs.fill_(-777.0)
got, want, _ = step_both(env, orc, actions)
print('sentinel reward lanes', np.flatnonzero(got[1]==12345.0)[:5], (got[1]==12345.0).sum(), 'term sentinel', (got[2].view(np.uint8)==77).sum(), 'trunc', (got[3].view(np.uint8)==77).sum(), 'obs sentinel rows', (got[0]==-777.0).any(axis=1).sum())
dr = np.abs(got[1].astype(np.float64) - want[1]); bad = np.flatnonzero(dr > 1e-2 + 1e-5*np.abs(want[1]))
print('bad lanes', bad)
st = env.get_state()
np.set_printoptions(linewidth=200, precision=8)
for i in bad[:3]:
    print('lane', i, 'status0', status[i], 'steps0', steps[i], 'prev', prev[i], 'flags', flags[i], 'action', actions[i])
    print(' x before', x_before[:, i]); print(' x gpu   ', st['x'][:, i]); print(' x orc   ', orc.x[:, i].astype(float))
    print(' r gpu', got[1][i], 'orc', want[1][i], 'ps gpu', st['prev_shaping'][i], 'ps orc', orc.prev_shaping[i], 'status gpu', st['status'][i], orc.status[i])
print('n bad', len(bad), 'min', bad.min() if len(bad) else None, 'max', bad.max() if len(bad) else None)
pf = prev.astype(np.float32).astype(np.float64)
for i in bad[:8]:
    implied = st['prev_shaping'][i] - float(got[1][i])
    j = int(np.nanargmin(np.abs(pf - implied)))
    print('lane', i, 'implied prev', implied, 'nearest prev lane', j, pf[j], 'reset-shaping?', -250.0)
newps = orc.prev_shaping.astype(np.float64)
for i in bad[:6]:
    implied = st['prev_shaping'][i] - float(got[1][i])
    j = int(np.nanargmin(np.abs(newps - implied)))
    print('lane', i, 'implied', implied, 'nearest NEW ps lane', j, newps[j], 'diff', newps[j]-implied)
wr = want[1]
for i in bad[:8]:
    j = int(np.argmin(np.abs(wr - float(got[1][i]))))
    print('lane', i, 'r_gpu', got[1][i], 'nearest oracle reward lane', j, wr[j], 'status0', status[i], 'pend', flags[i])
print('reward equal count', np.sum(np.abs(got[1]-wr) < 1e-3+1e-5*np.abs(wr)), 'of', n)
print('status0 of bad', status[bad][:40], 'flags', flags[bad][:40])
