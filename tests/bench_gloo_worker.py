"""A rank of `bench.py --gpus N` on a machine WITHOUT GPUs: bench.main() unmodified, with the two things that need a
device swapped for doubles inside this process only -- bench.Hip (device runtime: here the CPU and the gloo backend,
no hipGraphs) and gym_copter_amd.CopterVecEnv (the local stepper: here a stand-in that fills its outputs with the
global env ids).  Everything else is bench.py's own: the launcher environment, the process group, the barrier /
MAX-over-ranks timing, the packed all-gather leg of gym_copter_amd.sharded with its deadline, the line assembly.

Started by tests/test_bench_launcher.py through `python -m torch.distributed.run` (the driver's command shape)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


RealHip = bench.Hip


class CpuRuntime(RealHip):
    backend = "gloo"
    graphs = False

    def __init__(self, torch, local):
        self.torch = torch
        self.device = torch.device("cpu")
        self.index = local

    def synchronize(self):
        pass

    def stamp(self):
        return time.perf_counter()

    def elapsed_s(self, t0, t1):
        return t1 - t0

    def cus_and_clock_hz(self):
        return 256, bench.PEAK_ENGINE_CLOCK_HZ

    def init_group(self, dist):
        RealHip.init_group(self, dist)          # (bench.py's own: the gloo control group with its timeout)

    def open_collectives(self, dist):
        return dist.new_group(backend=self.backend)


class EnvDouble:
    """CopterVecEnv's surface as bench.main() uses it; outputs = the global env id of each row."""
    obs_dim, action_dim = 10, 4

    def __init__(self, task="lander3d", num_envs=1, device=None, env_id_base=0, **_):
        import torch
        self.n, self.base, self.device = num_envs, env_id_base, torch.device("cpu")
        self.single_observation_space = self.single_action_space = None      # (read, not used, by ShardedCopterVecEnv)
        self.bind_outputs(torch.empty((num_envs, self.obs_dim)), torch.empty(num_envs),
                          torch.empty(num_envs, dtype=torch.uint8), torch.empty(num_envs, dtype=torch.uint8))
        self.steps = 0

    def bind_outputs(self, obs, reward, term, trunc):
        self._out = (obs, reward, term, trunc)

    def reset(self, seed=None, options=None):
        return self._out[0], {}

    def step(self, actions):
        import torch
        assert actions.shape == (self.n, self.action_dim)
        ids = torch.arange(self.base, self.base + self.n, dtype=torch.float32)
        self._out[0].copy_(ids[:, None].expand(self.n, self.obs_dim))
        self._out[1].copy_(ids)
        self._out[2].fill_(self.steps & 1)
        self._out[3].zero_()
        self.steps += 1
        return self._out + ({},)

    def step_many(self, block):
        for a in block:
            self.step(a)

    def configure_pid(self):
        pass

    def rollout_pid(self, k):
        self.steps += k

    def pci_address(self):
        raise RuntimeError("no device")

    def clock_probe(self, waves):
        raise RuntimeError("no device")

    def close(self):
        pass


if __name__ == "__main__":
    import gym_copter_amd
    bench.Hip = CpuRuntime
    gym_copter_amd.CopterVecEnv = EnvDouble
    gym_copter_amd.vecenv.CopterVecEnv = EnvDouble       # (what gym_copter_amd.sharded instantiates: the --gather legs)
    bench.main(sys.argv[1:])
