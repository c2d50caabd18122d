#!/usr/bin/env python3
"""tools/rccl_two_ranks_one_gpu.py -- ONE bounded attempt at a 2-rank RCCL group on the hardware there is: a single
MI355X ("Next round" #8).

The parent never touches the GPU.  It starts two FRESH child processes (this same file with --rank r), both seeing the
same device 0, which try to open a 2-rank `nccl` (= RCCL) process group under a deadline.  Either outcome is the
deliverable:

  * RCCL refuses two ranks on one device  -> the error text is printed (and recorded in DESIGN.md section 6);
  * it works                              -> each rank runs an all-reduce of ones (ranks_seen), then steps a
    ShardedCopterVecEnv(gather="all") and checks the rank-major row order of PackedOutputs.all_gather_flat() against a
    plain 2n-env context on the same device, bit for bit.

Prints one JSON line: {"outcome": "refused" | "works" | "timeout" | "error", "ranks": [...]}.
"""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEADLINE_S = float(os.environ.get("RCCL_TWO_RANK_DEADLINE_S", 120.0))


def child(rank):
    sys.path.insert(0, ROOT)
    out = {"rank": rank, "stage": "import"}
    try:
        import datetime
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(0)
        out["stage"] = "init_process_group"
        dist.init_process_group("nccl", rank=rank, world_size=2, device_id=torch.device("cuda", 0),
                                timeout=datetime.timedelta(seconds=DEADLINE_S * 0.6))
        out["stage"] = "all_reduce"
        ones = torch.ones(1, device="cuda")
        dist.all_reduce(ones)
        torch.cuda.synchronize()
        out["ranks_seen"] = int(round(float(ones.item())))
        out["stage"] = "sharded env"
        import gym_copter_amd as gca
        from gym_copter_amd.sharded import ShardedCopterVecEnv
        n = 4096
        env = ShardedCopterVecEnv(task="lander3d", total_envs=2 * n, gather="all", device=0, seed=11,
                                  autoreset_mode="next_step")
        twin = gca.CopterVecEnv(task="lander3d", num_envs=2 * n, device=0, seed=11, autoreset_mode="next_step")
        env.reset()
        twin.reset()
        g = torch.Generator(device="cuda")
        g.manual_seed(5)
        ok = True
        for t in range(12):
            a = torch.rand((2 * n, 4), generator=g, device="cuda") * 2 - 1
            obs, rew, term, trunc, _ = env.step(a)
            wobs, wrew, wterm, wtrunc, _ = twin.step(a)
            torch.cuda.synchronize()
            ok = ok and torch.equal(obs, wobs) and torch.equal(rew, wrew) and torch.equal(term, wterm) \
                and torch.equal(trunc, wtrunc)
        out["gathered_rows_equal_the_unsharded_batch"] = bool(ok)
        out["stage"] = "done"
        env.close()
        twin.close()
        dist.destroy_process_group()
    except Exception as e:      # the error text is the deliverable
        out["error"] = "%s: %s" % (type(e).__name__, str(e)[-1500:])
    print("RESULT " + json.dumps(out), flush=True)


def main():
    if len(sys.argv) > 2 and sys.argv[1] == "--rank":
        child(int(sys.argv[2]))
        return
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE="2", LOCAL_RANK="0",
               HSA_ENABLE_IPC_MODE_LEGACY="0", NCCL_SOCKET_IFNAME="lo", NCCL_IB_DISABLE="1", NCCL_DEBUG="WARN")
    procs = []
    for r in (0, 1):
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), "--rank", str(r)],
                                      env=dict(env, RANK=str(r)), stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                                      text=True))
    t0 = time.time()
    outs, timed_out = [None, None], False
    for i, p in enumerate(procs):
        try:
            outs[i], _ = p.communicate(timeout=max(1.0, DEADLINE_S - (time.time() - t0)))
        except subprocess.TimeoutExpired:
            timed_out = True
            p.kill()                       # exactly the PID this script started
            outs[i], _ = p.communicate()
    ranks = []
    for i, o in enumerate(outs):
        rec = {"rank": i, "rc": procs[i].returncode}
        for line in (o or "").splitlines():
            if line.startswith("RESULT "):
                rec.update(json.loads(line[7:]))
        rec["log_tail"] = (o or "")[-1200:]
        ranks.append(rec)
    errs = " ".join(str(r.get("error", "")) + r["log_tail"] for r in ranks)
    if all(r.get("stage") == "done" and r.get("ranks_seen") == 2 and r.get("gathered_rows_equal_the_unsharded_batch")
           for r in ranks):
        outcome = "works"
    elif timed_out:
        outcome = "timeout"
    elif "uplicate GPU" in errs or "invalid usage" in errs.lower():
        outcome = "refused"
    else:
        outcome = "error"
    print(json.dumps({"outcome": outcome, "elapsed_s": round(time.time() - t0, 1), "ranks": ranks}))


if __name__ == "__main__":
    main()
