#!/usr/bin/env python3
"""Batch counterpart of the reference's top-level `lander.py` demo (reference lander.py:25-70,
78-107) without the GUI: runs the constant-thrust (or `--random`) action law on a batch of
Lander-v0 environments on the GPU and, like `lander.py --save`, writes the trajectory of ONE
environment of the batch as CSV in the reference's format

    t,m1,m2,m3,m4,X,dX,Y,dY,Z,dZ,Phi,dPhi,Theta,dTheta        ('%f' formatting, lander.py:33-38,48-54)

which `utils/copter-plot.py` of the reference reads.

    python -m gym_copter_amd.demo --num-envs 65536 --save traj.csv [--random] [--env-index 0]
"""
import argparse

import numpy as np

MOTORVAL = 1.625e-2          # reference lander.py:21


def write_csv_header(f, state_names):
    f.write('t,' + ','.join([('m%d' % k) for k in range(1, 5)]))
    f.write(',' + ','.join(state_names) + '\n')


def write_csv_row(f, t, action, state):
    f.write('%f' % t)
    f.write((',%f' * 4) % tuple(action))
    f.write(((',%f' * len(state)) + '\n') % tuple(state))


def heuristic(env, csvfilename=None, random=False, env_index=0, seed=None, max_steps=2000,
              forces=None, verbose=True):
    """The reference's heuristic loop (lander.py:25-70) for a whole batch; follows env
    `env_index` for the CSV trace and the printed running reward.  Returns
    (steps, total_reward) of that env's first episode."""
    import torch
    n = env.num_envs
    rng = np.random.default_rng(seed)
    options = None if forces is None else {"forces": forces}
    env.reset(seed=seed, options=options)
    dt = 1. / env.unwrapped.FRAMES_PER_SECOND
    csvfile = None
    if csvfilename is not None:
        csvfile = open(csvfilename, 'w')
        write_csv_header(csvfile, env.STATE_NAMES)
    total_reward, steps = 0.0, 0
    while steps < max_steps:
        a = MOTORVAL * (rng.standard_normal((n, 4)) if random else np.ones((n, 4)))
        a = torch.from_numpy(a.astype(np.float32)).to(env.device)
        obs, reward, term, trunc, _ = env.step(a)
        total_reward += float(reward[env_index])
        if csvfile is not None:
            write_csv_row(csvfile, dt * steps, a[env_index].tolist(), obs[env_index].tolist())
        steps += 1
        if verbose:
            print('steps =  %04d    total_reward = %+0.2f' % (steps, total_reward))
        if bool(term[env_index]) or bool(trunc[env_index]):
            break
    if csvfile is not None:
        csvfile.close()
    return steps, total_reward


def main():
    import gym_copter_amd
    p = argparse.ArgumentParser(formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    p.add_argument('--save', dest='csvfilename', help='Save trajectory of one env in CSV file')
    p.add_argument('--random', action='store_true', help='Use random motor values for comparison')
    p.add_argument('--num-envs', type=int, default=1024)
    p.add_argument('--env-index', type=int, default=0)
    p.add_argument('--seed', type=int, default=0)
    args = p.parse_args()
    env = gym_copter_amd.make('gym_copter:Lander-v0', num_envs=args.num_envs, seed=args.seed,
                              autoreset_mode='disabled')
    heuristic(env, args.csvfilename, args.random, args.env_index, args.seed)
    env.close()


if __name__ == '__main__':
    main()
