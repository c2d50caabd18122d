"""Multi-GPU sharding of the env batch: one process per GPU (torch.distributed, backend
"nccl" = RCCL over xGMI on ROCm; "gloo" on CPU for tests).

The reference has a single environment and no distributed layer, so nothing here
translates upstream code.  Environments are fully independent (reference
gym_copter/envs/task.py:161 builds one Dynamics per env), so the batch shards trivially:

  * rank r owns the contiguous global env ids [r*n_local, (r+1)*n_local);
  * every random draw is keyed by the GLOBAL env id, so trajectories do not depend on the
    number of GPUs;
  * stepping needs no communication at all.  The only exchange is the optional
    concatenated return (observations, and reward/terminated/truncated if asked for):
    one all-gather per array per step, issued on the current stream right behind the step
    kernel.  A caller whose policy is replicated per GPU should leave gather off.
"""
import numpy as np


def shard_bounds(total_envs, world_size, rank):
    """(first global env id, number of local envs) of `rank`.  Equal shards are required so
    that all_gather_into_tensor can write the concatenation directly."""
    if total_envs % world_size != 0:
        raise ValueError("total_envs (%d) must be divisible by the world size (%d)"
                         % (total_envs, world_size))
    n_local = total_envs // world_size
    return rank * n_local, n_local


class ShardGather:
    """Pre-allocated all-gather of per-env rows: local [n_local, ...] -> global [N, ...]
    in global env-id order (rank-major)."""

    def __init__(self, n_local, world_size, group=None):
        self.n_local, self.world, self.group = n_local, world_size, group
        self._out = {}

    def __call__(self, name, local):
        import torch
        import torch.distributed as dist
        if self.world == 1:
            return local
        key = (name, tuple(local.shape[1:]), local.dtype, local.device)
        out = self._out.get(key)
        if out is None:
            out = torch.empty((self.world * self.n_local,) + tuple(local.shape[1:]),
                              dtype=local.dtype, device=local.device)
            self._out[key] = out
        dist.all_gather_into_tensor(out, local.contiguous(), group=self.group)
        return out


class ShardedCopterVecEnv:
    """CopterVecEnv over `total_envs` environments sharded across the ranks of a process
    group.  step()/reset() take and return LOCAL rows unless gather is enabled."""

    def __init__(self, task="lander3d", total_envs=1, gather="none", group=None, device=None,
                 local_env_factory=None, **env_kwargs):
        import torch.distributed as dist
        if gather not in ("none", "obs", "all"):
            raise ValueError("gather must be 'none', 'obs' or 'all'")
        if dist.is_available() and dist.is_initialized():
            self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        else:
            self.rank, self.world = 0, 1
        self.total_envs = int(total_envs)
        self.env_id_base, self.n_local = shard_bounds(self.total_envs, self.world, self.rank)
        if local_env_factory is None:
            from .vecenv import CopterVecEnv
            local_env_factory = CopterVecEnv
        if device is None:
            import os
            device = int(os.environ.get("LOCAL_RANK", 0))
        self.local = local_env_factory(task=task, num_envs=self.n_local, device=device,
                                       env_id_base=self.env_id_base, **env_kwargs)
        self.gather = gather
        self._gather = ShardGather(self.n_local, self.world, group)
        self.num_envs = self.total_envs if gather != "none" else self.n_local
        self.obs_dim = self.local.obs_dim
        self.single_observation_space = self.local.single_observation_space
        self.single_action_space = self.local.single_action_space

    def local_slice(self):
        return slice(self.env_id_base, self.env_id_base + self.n_local)

    def _local_actions(self, actions):
        if actions.shape[0] == self.n_local:
            return actions
        if actions.shape[0] == self.total_envs:
            return actions[self.local_slice()]
        raise ValueError("actions must have %d (local) or %d (global) rows, got %d"
                         % (self.n_local, self.total_envs, actions.shape[0]))

    def reset(self, seed=None, options=None):
        obs, info = self.local.reset(seed=seed, options=options)
        if self.gather != "none":
            obs = self._gather("obs", obs)
        return obs, info

    def step(self, actions):
        obs, reward, term, trunc, infos = self.local.step(self._local_actions(actions))
        if self.gather != "none":
            obs = self._gather("obs", obs)
        if self.gather == "all":
            import torch
            reward = self._gather("reward", reward)
            term = self._gather("term", term.view(torch.uint8)).view(torch.bool)
            trunc = self._gather("trunc", trunc.view(torch.uint8)).view(torch.bool)
        return obs, reward, term, trunc, infos

    def close(self):
        self.local.close()
