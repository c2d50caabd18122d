import sys, time
sys.path.insert(0, "/root/repo")
import torch, gym_copter_amd
env = gym_copter_amd.CopterVecEnv("lander3d", 64, seed=1, autoreset_mode="next_step")
env.reset()
a = torch.rand((64, 4), device=env.device) * 2 - 1
for label, fast in (("_cs_call", env._fast), ("ctypes", None)):
    env._fast = fast
    for j in range(2000): env.step(a)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for j in range(2000): env.step(a)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    print("python 64 envs (%s): host enqueue %.2f us per step" % (label, (t1 - t0) / 2000 * 1e6))
