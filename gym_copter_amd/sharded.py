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
    every env's outputs as one row of a packed per-rank array (PackedOutputs: the "packed rows" of
    include/copterstep.h), so nothing is copied before the collective, nor after it.  It is issued on the current stream right behind the step kernel.
    A caller whose policy is replicated per GPU should leave gather off.
"""
import os



def _force_collective(flag):
    """With a process group of ONE rank the all-gathers below are identities and are shortcut; asked to
    (force_collective=True, or COPTERSTEP_FORCE_COLLECTIVE=1 in the environment) they are issued all the same,
    so that the RCCL path -- eager and hipGraph-captured -- can be exercised on a single GPU."""
    return bool(flag) if flag is not None else os.environ.get("COPTERSTEP_FORCE_COLLECTIVE", "0") == "1"


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

    def __init__(self, n_local, world_size, group=None, force_collective=None):
        self.n_local, self.world, self.group = n_local, world_size, group
        self.force = _force_collective(force_collective)
        self._out = {}

    def __call__(self, name, local):
        import torch
        import torch.distributed as dist
        if self.world == 1 and not (self.force and dist.is_initialized()):
            return local
        key = (name, tuple(local.shape[1:]), local.dtype, local.device)
        out = self._out.get(key)
        if out is None:
            out = torch.empty((self.world * self.n_local,) + tuple(local.shape[1:]),
                              dtype=local.dtype, device=local.device)
            self._out[key] = out
        dist.all_gather_into_tensor(out, local if local.is_contiguous() else local.contiguous(), group=self.group)
        return out


def row_views(rows, obs_dim):
    """(obs, reward, terminated u8, truncated u8) as views of packed rows [..., obs_dim + 2] float32: row =
    {observation, reward, flags word (byte 0 terminated, byte 1 truncated)} -- include/copterstep.h, "packed rows"."""
    import torch
    b = rows.view(torch.uint8)                     # [..., 4 * (obs_dim + 2)]
    f = 4 * (obs_dim + 1)
    return rows[..., :obs_dim], rows[..., obs_dim], b[..., f], b[..., f + 1]


class PackedOutputs:
    """One [n_local, obs_dim + 2] float32 array per rank -- the step kernel's "packed rows": every env's observation,
    reward and flags as ONE row -- and its all-gathered counterpart [world, n_local, obs_dim + 2].  The local columns
    are the tensors the step kernel writes (zero-copy: bind_outputs); the gathered rows are in global env-id order
    (rank-major) and contiguous, so one collective moves everything and the flat [N, ...] return needs no copy
    either."""

    def __init__(self, n_local, obs_dim, world_size, device, group=None, force_collective=None):
        import torch
        self.n, self.od, self.world, self.group = n_local, obs_dim, world_size, group
        self.force = _force_collective(force_collective)
        self.nbytes = n_local * (obs_dim + 2) * 4
        self.local = torch.zeros((n_local, obs_dim + 2), dtype=torch.float32, device=device)
        self.gathered = torch.zeros((world_size, n_local, obs_dim + 2), dtype=torch.float32, device=device)
        self.obs, self.reward, self.term, self.trunc = row_views(self.local, obs_dim)

    def all_gather(self):
        """-> (obs [world, n, od], reward [world, n], terminated [world, n] u8, truncated u8):
        views of the gathered buffer, rank-major = global env-id order."""
        import torch.distributed as dist
        if self.world == 1 and not (self.force and dist.is_initialized()):
            self.gathered[0].copy_(self.local)
        else:
            dist.all_gather_into_tensor(self.gathered.view(-1), self.local.view(-1), group=self.group)
        return row_views(self.gathered, self.od)

    def all_gather_flat(self):
        """The same as [N, ...] rows in global env-id order: views as well (the gathered rows are contiguous)."""
        self.all_gather()
        return row_views(self.gathered.view(self.world * self.n, self.od + 2), self.od)


class ShardedCopterVecEnv:
    """CopterVecEnv over `total_envs` environments sharded across the ranks of a process
    group.  step()/reset() take and return LOCAL rows unless gather is enabled."""

    def __init__(self, task="lander3d", total_envs=1, gather="none", group=None, device=None,
                 env_id_offset=0, flat=True, force_collective=None, **env_kwargs):
        import torch.distributed as dist
        if gather not in ("none", "obs", "all"):
            raise ValueError("gather must be 'none', 'obs' or 'all'")
        if dist.is_available() and dist.is_initialized():
            self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        else:
            self.rank, self.world = 0, 1
        self.total_envs = int(total_envs)
        first, self.n_local = shard_bounds(self.total_envs, self.world, self.rank)
        self.env_id_base = int(env_id_offset) + first    # global id of local row 0 (keys every random draw)
        self.first_row = first                           # row of local row 0 in this env's gathered outputs
        from . import vecenv
        if device is None:
            import os
            device = int(os.environ.get("LOCAL_RANK", 0))
        if gather == "obs":
            # the observation rows are what the collective ships: a contiguous [n_local, obs_dim] buffer that the
            # step kernel writes directly (the default packed rows would need a .contiguous() copy every step)
            env_kwargs.setdefault("contiguous_outputs", True)
        self.local = vecenv.CopterVecEnv(task=task, num_envs=self.n_local, device=device,
                                         env_id_base=self.env_id_base, **env_kwargs)
        self.gather = gather
        # gather="all": flat=True returns [N, ...] rows, flat=False [world, n_local, ...]; both are views of the
        # gathered packed rows (rank-major = global env-id order), no copy after the collective
        self.flat = bool(flat)
        self._gather = ShardGather(self.n_local, self.world, group, force_collective)
        self._packed = None
        if gather == "all":
            import torch
            dev = getattr(self.local, "device", torch.device("cpu"))
            self._packed = PackedOutputs(self.n_local, self.local.obs_dim, self.world, dev, group, force_collective)
            if hasattr(self.local, "bind_outputs"):      # the kernel writes into the packed buffer
                pk = self._packed
                self.local.bind_outputs(pk.obs, pk.reward, pk.term, pk.trunc)
        self.num_envs = self.total_envs if gather != "none" else self.n_local
        self.obs_dim = self.local.obs_dim
        self.single_observation_space = self.local.single_observation_space
        self.single_action_space = self.local.single_action_space

    def local_slice(self):
        return slice(self.first_row, self.first_row + self.n_local)

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
            pk = self._packed
            if obs.data_ptr() != pk.obs.data_ptr():     # a local env that owns its outputs: pack them
                pk.obs.copy_(obs)
                pk.reward.copy_(reward)
                pk.term.copy_(term.view(torch.uint8))
                pk.trunc.copy_(trunc.view(torch.uint8))
            if self.flat:    # [N, ...] rows in global env-id order: views of the gathered rows, no copy
                obs, reward, g_term, g_trunc = pk.all_gather_flat()              # ONE collective
            else:            # the [world, n_local, ...] views
                obs, reward, g_term, g_trunc = pk.all_gather()
            term, trunc = g_term.view(torch.bool), g_trunc.view(torch.bool)
        return obs, reward, term, trunc, infos

    def close(self):
        self.local.close()


class HalfBatchPipeline:
    """Double-buffered half-batches: the global batch is two halves, each sharded
    over the ranks, each with its own stream.  step_async(h, actions) enqueues half h's step kernel
    and its packed all-gather on stream h and returns at once; wait(h) orders the caller's stream
    behind them.  While half h's collective is on the links the caller evaluates its policy on the
    other half and steps it -- the schedule

        a0 = policy(obs0); step_async(0, a0)
        loop:  a1 = policy(obs1); step_async(1, a1); obs0.. = wait(0)
               a0 = policy(obs0); step_async(0, a0); obs1.. = wait(1)

    Global env ids are half-major: half h owns [h*T/2, (h+1)*T/2), rank r the r-th contiguous
    shard of it, so every env keeps the id (hence the random draws) it has in one T-env batch
    and the rows of half h come back in global order."""

    def __init__(self, task="lander3d", total_envs=2, gather="all", group=None, device=None, **env_kwargs):
        if total_envs % 2:
            raise ValueError("total_envs (%d) must be even" % total_envs)
        half = total_envs // 2
        self.total_envs, self.half_envs = int(total_envs), half
        self.halves = [ShardedCopterVecEnv(task=task, total_envs=half, gather=gather, group=group,
                                           device=device, env_id_offset=h * half, **env_kwargs)
                       for h in (0, 1)]
        self.n_local = self.halves[0].n_local          # local envs PER HALF
        self.rank, self.world = self.halves[0].rank, self.halves[0].world
        self.gather = gather
        self.num_envs = self.halves[0].num_envs        # rows per half that step()/wait() return
        self.obs_dim = self.halves[0].obs_dim
        self.single_observation_space = self.halves[0].single_observation_space
        self.single_action_space = self.halves[0].single_action_space
        self._streams = self._events = None
        dev = getattr(self.halves[0].local, "device", None)
        if dev is not None and getattr(dev, "type", "cpu") == "cuda":
            import torch
            self._dev = dev
            self._streams = [torch.cuda.Stream(device=dev) for _ in (0, 1)]
            self._events = [torch.cuda.Event() for _ in (0, 1)]
        self._out = [None, None]

    def reset(self, seed=None, options=None):
        """-> ([obs half 0, obs half 1], {}); `options` is one dict per half (or None)."""
        options = options or (None, None)
        if self._streams is not None:
            # a step_async(h) that was never wait()ed for may still be running on its side stream: the
            # reset kernels, enqueued on the caller's stream, touch the same tiles and output buffers
            import torch
            cur = torch.cuda.current_stream(self._dev)
            for h in (0, 1):
                if self._out[h] is not None:
                    cur.wait_event(self._events[h])
        obs = [self.halves[h].reset(seed=seed, options=options[h])[0] for h in (0, 1)]
        self._out = [None, None]
        return obs, {}

    def step_async(self, half, actions):
        """Enqueue half `half`'s step (+ gather) behind everything already on the caller's stream;
        the returned tensors are valid for the caller's stream after wait(half)."""
        env = self.halves[half]
        if self._streams is None:                      # host tensors (tests): nothing to overlap
            self._out[half] = env.step(actions)
            return self._out[half]
        import torch
        s = self._streams[half]
        s.wait_stream(torch.cuda.current_stream(self._dev))   # the actions, and the last readers of the outputs
        with torch.cuda.stream(s):
            self._out[half] = env.step(actions)
            self._events[half].record(s)
        return self._out[half]

    def wait(self, half):
        """Order the caller's current stream behind half `half`'s last step_async; -> its outputs."""
        if self._streams is not None:
            import torch
            torch.cuda.current_stream(self._dev).wait_event(self._events[half])
        return self._out[half]

    def step(self, actions):
        """Both halves, pipelined against each other: `actions` = (half 0 rows, half 1 rows);
        -> [outputs of half 0, outputs of half 1]."""
        self.step_async(0, actions[0])
        self.step_async(1, actions[1])
        return [self.wait(0), self.wait(1)]

    def close(self):
        for e in self.halves:
            e.close()
