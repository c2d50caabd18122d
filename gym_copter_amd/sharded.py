"""Multi-GPU sharding of the env batch: one process per GPU (torch.distributed, backend
"nccl" = RCCL over xGMI on ROCm; "gloo" on CPU for tests).

The reference has a single environment and no distributed layer, so nothing here
translates upstream code.  Environments are fully independent (reference
gym_copter/envs/task.py:161 builds one Dynamics per env), so the batch shards trivially:

  * rank r owns the contiguous global env ids [r*n_local, (r+1)*n_local);
  * every random draw is keyed by the GLOBAL env id, so trajectories do not depend on the
    number of GPUs;
  * stepping needs no communication at all.  The only exchange is the optional
    concatenated return: gather="obs" ships the observation rows, gather="all" ships
    observations, rewards and both flags in ONE all-gather per step -- the step kernel writes
    its outputs straight into one packed per-rank buffer (PackedOutputs), so nothing is copied
    before the collective.  It is issued on the current stream right behind the step kernel.
    A caller whose policy is replicated per GPU should leave gather off.
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
        dist.all_gather_into_tensor(out, local if local.is_contiguous() else local.contiguous(), group=self.group)
        return out


class PackedOutputs:
    """One byte buffer per rank holding [obs f32 | reward f32 | terminated u8 | truncated u8]
    (each section 16-byte aligned), and its all-gathered counterpart [world, bytes].  The local
    sections are the tensors the step kernel writes; the global ones are strided views of the
    gathered buffer, so one collective moves everything."""

    def __init__(self, n_local, obs_dim, world_size, device, group=None):
        import torch
        self.n, self.od, self.world, self.group = n_local, obs_dim, world_size, group
        up = lambda b: (b + 15) // 16 * 16
        self.off_obs = 0
        self.off_rew = up(n_local * obs_dim * 4)
        self.off_term = self.off_rew + up(n_local * 4)
        self.off_trunc = self.off_term + up(n_local)
        self.nbytes = self.off_trunc + up(n_local)
        self.local = torch.zeros(self.nbytes, dtype=torch.uint8, device=device)
        self.gathered = torch.zeros((world_size, self.nbytes), dtype=torch.uint8, device=device)
        self.obs, self.reward, self.term, self.trunc = self._sections(self.local.view(1, -1), 1)
        self.obs, self.reward = self.obs[0], self.reward[0]
        self.term, self.trunc = self.term[0], self.trunc[0]

    def _sections(self, buf2d, rows):
        import torch
        n, od = self.n, self.od
        obs = buf2d[:, self.off_obs:self.off_obs + n * od * 4].view(torch.float32).view(rows, n, od)
        rew = buf2d[:, self.off_rew:self.off_rew + n * 4].view(torch.float32)
        term = buf2d[:, self.off_term:self.off_term + n]
        trunc = buf2d[:, self.off_trunc:self.off_trunc + n]
        return obs, rew, term, trunc

    def all_gather(self):
        """-> (obs [world, n, od], reward [world, n], terminated [world, n] u8, truncated u8):
        views of the gathered buffer, rank-major = global env-id order."""
        import torch.distributed as dist
        if self.world == 1:
            self.gathered[0].copy_(self.local)
        else:
            dist.all_gather_into_tensor(self.gathered.view(-1), self.local, group=self.group)
        return self._sections(self.gathered, self.world)


class ShardedCopterVecEnv:
    """CopterVecEnv over `total_envs` environments sharded across the ranks of a process
    group.  step()/reset() take and return LOCAL rows unless gather is enabled."""

    def __init__(self, task="lander3d", total_envs=1, gather="none", group=None, device=None,
                 **env_kwargs):
        import torch.distributed as dist
        if gather not in ("none", "obs", "all"):
            raise ValueError("gather must be 'none', 'obs' or 'all'")
        if dist.is_available() and dist.is_initialized():
            self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        else:
            self.rank, self.world = 0, 1
        self.total_envs = int(total_envs)
        self.env_id_base, self.n_local = shard_bounds(self.total_envs, self.world, self.rank)
        from . import vecenv
        if device is None:
            import os
            device = int(os.environ.get("LOCAL_RANK", 0))
        self.local = vecenv.CopterVecEnv(task=task, num_envs=self.n_local, device=device,
                                         env_id_base=self.env_id_base, **env_kwargs)
        self.gather = gather
        self._gather = ShardGather(self.n_local, self.world, group)
        self._packed = None
        if gather == "all":
            import torch
            dev = getattr(self.local, "device", torch.device("cpu"))
            self._packed = PackedOutputs(self.n_local, self.local.obs_dim, self.world, dev, group)
            if hasattr(self.local, "bind_outputs"):      # the kernel writes into the packed buffer
                pk = self._packed
                self.local.bind_outputs(pk.obs, pk.reward, pk.term, pk.trunc)
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
        if self.gather == "obs":
            obs = self._gather("obs", obs)
        elif self.gather == "all":
            import torch
            pk, N = self._packed, self.total_envs
            if obs.data_ptr() != pk.obs.data_ptr():     # a local env that owns its outputs: pack them
                pk.obs.copy_(obs)
                pk.reward.copy_(reward)
                pk.term.copy_(term.view(torch.uint8))
                pk.trunc.copy_(trunc.view(torch.uint8))
            g_obs, g_rew, g_term, g_trunc = pk.all_gather()       # ONE collective
            # [world, n_local, ...] views -> [N, ...] rows in global env-id order
            obs, reward = g_obs.reshape(N, self.obs_dim), g_rew.reshape(N)
            term, trunc = g_term.reshape(N).view(torch.bool), g_trunc.reshape(N).view(torch.bool)
        return obs, reward, term, trunc, infos

    def close(self):
        self.local.close()
