#!/usr/bin/env python3
"""bench.py -- env-steps/s of the fused copter step kernel on MI355X.

A "step" is ONE pass of the hot path (one kernel launch = _Task.step() for every env of
the batch) over one batch of synthetic actions that is already resident in HBM.  Default
workload = BASELINE.json configs[1]: Lander3D, 65 536 envs, uniform random actions in
[-1,1)^4, auto-reset on (NEXT_STEP), float32 state words, one GPU.

  python bench.py --gpus N --steps K --warmup W
      N = 1: runs in this process.
      N > 1 without a torch.distributed environment: this process starts the N ranks itself
             (python -m torch.distributed.run, one rank per GPU, rendezvous on 127.0.0.1)
             BEFORE it touches the GPU, relays their output and exits with their code.
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...   (the driver's form)

How the K steps are timed.  The K steps are captured as hipGraphs of at most --graph-chunk launches
(several passes of the K steps per graph when K is small).  A timed REGION is R back-to-back passes of
those K steps, bracketed by
barrier + torch.cuda.synchronize() on both sides and timed with the host clock; R ("repeats") is
chosen so that a region holds >= --min-region-ms of GPU work (a single 20-step pass is 0.1 ms:
graph-launch and synchronisation overhead would be a third of it).  --regions such regions are timed,
each reduced with MAX over ranks; the MEDIAN region gives ms_per_step = wall / (R * K) and
value = total envs * R * K / wall.  `single_pass` reports one bare pass for comparison.

Rank 0 prints ONE COMPACT JSON line on stdout (<= 8 000 bytes, asserted here and in tests/): the contract keys,
  roofline     : algorithmic bytes (176 B/env-step) per launch over the
                 launch duration measured with HIP events on the launch stream, vs 8 TB/s; beside it the kernel-only
                 span (kernel_span_us / kernel_frac) and the PMC traffic per launch
  cpu_baseline : the scalar NumPy port of the reference (oracle/refcpu.py), timed here on
                 the host, 1 core, bounded sample (a reported baseline, not a target); beside
                 it all usable host cores (one process and env each) and the vectorised NumPy oracle
  summary      : <= 1.5 KB: roofline fraction per sweep point, us per step of the K-step paths, config 5's bounds
and writes the FULL record (every leg with its clocks, region spreads, byte models, bounds and notes) to
--full-out (default gpurun_out/bench_full.json) and to stderr.  The default run times the headline, its kernel-only span,
the CPU baseline, Hover3D at 262 144 envs (BASELINE configs[2]), config 5 at 65 536 envs (configs[4]), the headline's
envs under the other two action laws (lander.py's constant thrust; near hover = the low-churn variant), the HBM-resident
Lander3D point (4 M envs) and the two K-steps-per-launch paths on the headline's envs (cs_step_many: open loop over the
resident action ring; cs_rollout_pid: closed loop under the on-device PID heuristic) -- reported BESIDE the headline,
never as `value`.  --full adds the rest of the sweep (262 144 / 1 M envs, Hover3D 1 M / 4 M, config 5
at 1 M, the K-step kernels at 4 M), cs_rollout_random, the caller-compiled policy, the served leg and the launch floor.
Multi-GPU: the env batch is sharded by contiguous env-id range with no data-path
collective in the timed region ("scaling": "weak").  The barriers and the MAX over ranks run on a gloo
group; the RCCL communicator is opened only for the all-gather legs: at N > 1 one packed all-gather leg
(eager launches, last, under a deadline: value_with_packed_allgather), with --gather all three legs
(obs rows, packed, double-buffered half-batches; hipGraph-captured), timed separately from `value`.
north_star's other action law (lander.py's constant thrust) is a sweep point at N = 1 and, at N > 1, its own leg
on the same shards, timed like the headline: value_constant_thrust.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ALGO_BYTES = {"lander3d": 176, "hover3d": 176}
HBM_PEAK_GBPS = 8000.0                              # MI355X_MICROARCH.md: HBM3E 8 TB/s
# MI355X_MICROARCH.md: peak FP32 vector 157.3 TFLOP/s; float64 vector FMA issues at half that rate
F64_VALU_PEAK_TFLOPS = 78.6
HOVER = 0.016560178185018043                        # motor value with thrust == weight
# Engine clock and SIMD count for the instruction-issue bound of the float64-heavy kernels: a SIMD issues one vector
# instruction of a wavefront every 4 cycles (64 lanes over 16 ALU lanes; MI355X_MICROARCH.md), so a launch whose
# SIMDs each hold W wavefronts of V executed vector instructions cannot take less than W * V * 4 cycles.
# The counts V (and the float64 flops of config 5) are NOT a model: they are rocprofv3 PMC figures of the very
# instantiations, read from profiles/pmc_counts.json (scripts/profile_gpu.sh -> scripts/collect_profiles.py), and
# withheld when the tree's kernel sources hash differently from the ones they were measured on.
PEAK_ENGINE_CLOCK_HZ = 2.4e9
SIMDS_PER_CU, LANES_PER_WAVE, ISSUE_CYCLES = 4, 64, 4
INFINITY_CACHE_BYTES = 256 << 20                    # MI355X_MICROARCH.md: 256 MiB memory-side cache
# The ONE stdout line is for a reader with a bounded buffer (the driver keeps ~8 KB of a run's stdout): contract keys,
# roofline, cpu_baseline and a short summary only.  Everything else goes to the full record (--full-out) and stderr.
LINE_BUDGET = 8000
DEGRADED_EXIT = 3                                   # exit code when the N > 1 gather leg hit its deadline or failed ("status": "degraded")
SUMMARY_BUDGET = 1500

def parse(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=2000)
    p.add_argument("--warmup", type=int, default=200)
    p.add_argument("--envs", type=int, default=65536, help="envs per GPU")
    p.add_argument("--task", default="lander3d", choices=["lander3d", "hover3d"])
    p.add_argument("--actions", default="uniform", choices=["uniform", "near_hover", "const"])
    p.add_argument("--state", default="float32")
    p.add_argument("--substeps", type=int, default=1)
    p.add_argument("--no-graph", action="store_true", help="eager launches instead of hipGraph replay")
    p.add_argument("--graph-chunk", type=int, default=100)
    p.add_argument("--ring", type=int, default=64, help="resident action batches cycled through")
    p.add_argument("--produce-actions", action="store_true",
                   help="A/B workload: every step's action batch is WRITTEN by a preceding kernel (a device copy "
                        "from the ring into one buffer) instead of being read from the resident ring")
    p.add_argument("--min-region-ms", type=float, default=50.0, help="GPU work per timed region")
    p.add_argument("--regions", type=int, default=5, help="timed regions (the median is reported)")
    p.add_argument("--gather", action="store_true",
                   help="also time with the RCCL all-gathers of the concatenated return (on one GPU it runs them in a "
                        "1-rank RCCL group with the collectives forced)")
    p.add_argument("--default-gather-leg", action="store_true",
                   help="one GPU: run the packed all-gather leg the way an N > 1 run does by default (last, under its "
                        "deadline) in a 1-rank RCCL group with the collective forced -- to exercise that path where "
                        "only one GPU is at hand")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--cpu-seconds", type=float, default=6.0,
                   help="seconds of the one-core scalar port (the other CPU rows take 2 s each)")
    p.add_argument("--no-sweep", action="store_true", help="skip the batch-size / task sweep and config 5")
    p.add_argument("--full", action="store_true",
                   help="everything: the whole batch-size / task / action-law sweep, the K-step kernels at 4 M envs, the "
                        "on-device random policy and the caller-compiled policy, the served leg, the fused caller-policy "
                        "child and the launch-floor probe.  The DEFAULT run is the headline + its kernel-only span + the "
                        "CPU baseline + one point per remaining single-GPU BASELINE config (Hover3D 262 144, config 5 at "
                        "65 536) + the HBM-resident Lander3D point (4 M envs) + cs_step_many / cs_rollout_pid")
    p.add_argument("--full-out", default=None,
                   help="where the FULL record goes (default gpurun_out/bench_full.json under the repo root); stdout "
                        "carries only the compact line (<= %d bytes)" % LINE_BUDGET)
    p.add_argument("--pid", type=int, default=100,
                   help="also time cs_rollout_pid / cs_rollout_random with this many steps per launch (0 = skip)")
    p.add_argument("--served", type=int, default=None,
                   help="also time served stepping (cs_serve_*: producers ahead of the env) with this many steps per session "
                        "(0 = skip; default: 500 under --full, else skipped)")
    p.add_argument("--no-clock-sampling", action="store_true",
                   help="do not read the device's hwmon sensors (sclk / power / temperature) around the timed regions")
    p.add_argument("--no-span", action="store_true",
                   help="skip the kernel-only span figure (a child process on the span build, before this one touches the GPU)")
    p.add_argument("--served-graph", type=int, default=-1, help="diagnostic: feeders of the served legs from a hipGraph (1) "
                   "or eagerly (0); default: as the headline")
    p.add_argument("--many", type=int, default=100,
                   help="also time cs_step_many with this many steps per launch (0 = skip)")
    p.add_argument("--master-port", type=int, default=0, help="self-launch only: rendezvous port (0 = pick a free one)")
    a = p.parse_args(argv)
    if a.served is None:
        a.served = 500 if a.full else 0
    return a


# ------------------------------------------------------------------------------------------
# N > 1 without a launcher: start the ranks ourselves, before anything touches the GPU
# ------------------------------------------------------------------------------------------
def needs_self_launch(gpus, environ):
    """True when --gpus N > 1 was asked of a process that no launcher has given a rank."""
    return gpus > 1 and "WORLD_SIZE" not in environ and "RANK" not in environ


def _free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launcher_command(gpus, argv, port=0, python=None):
    """The command that runs this script as `gpus` ranks of one node (one per GPU, RCCL)."""
    return [python or sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
            "--nproc-per-node", str(gpus), "--master-addr", "127.0.0.1",
            "--master-port", str(port or _free_port()), os.path.abspath(__file__)] + list(argv)


def self_launch(a, argv):
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC (RCCL across processes)
    env.setdefault("MASTER_ADDR", "127.0.0.1")
    cmd = launcher_command(a.gpus, argv, a.master_port)
    print("bench.py: starting %d ranks: %s" % (a.gpus, " ".join(cmd)), file=sys.stderr, flush=True)
    return subprocess.call(cmd, env=env)       # a child process: this one never initialises the GPU


class stdout_to_stderr:
    """File descriptor 1 -> 2 for the duration: RCCL prints a version banner on stdout when its first
    communicator is set up, and stdout is where the ONE JSON line goes."""

    def __enter__(self):
        sys.stdout.flush()
        self.saved = os.dup(1)
        os.dup2(2, 1)

    def __exit__(self, *exc):
        sys.stdout.flush()
        os.dup2(self.saved, 1)
        os.close(self.saved)


# ------------------------------------------------------------------------------------------
def make_actions(torch, law, ring, n, device, seed):
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    if law == "uniform":
        return torch.rand((ring, n, 4), generator=g, device=device, dtype=torch.float32) * 2 - 1
    if law == "near_hover":
        return HOVER * (1 + 0.01 * torch.randn((ring, n, 4), generator=g, device=device, dtype=torch.float32))
    return torch.full((ring, n, 4), 1.625e-2, device=device, dtype=torch.float32)


RCCL_OPEN = [False]      # set once main() has opened the RCCL communicator (collectives())


def quiesce_collectives(torch):
    """Before a capture in a process whose RCCL communicator is open: let the eager collectives finish and give
    ProcessGroupNCCL's watchdog (polling period 100 ms) time to retire them, so that it has nothing to query while
    the capture is open.  (The gloo control group has no such thread.)"""
    if not RCCL_OPEN[0]:
        return
    try:
        torch.cuda.synchronize()
        time.sleep(0.3)
    except Exception:
        pass


class Stepper:
    """Runs `count` consecutive env steps, as hipGraph replays of `chunk` captured
    launches plus eager launches for the remainder."""

    def __init__(self, torch, env, actions, use_graph, chunk, post=None, produce=False):
        self.torch, self.env, self.actions, self.post = torch, env, actions, post
        self.ring = actions.shape[0]
        # produce: the action row is written by a kernel right before the step, as a policy would
        self.produced = torch.empty_like(actions[0]) if produce else None
        self.pos = 0
        self.graph = None
        self.chunk = chunk
        if use_graph:
            s = torch.cuda.Stream(device=env.device)
            s.wait_stream(torch.cuda.current_stream(env.device))
            with torch.cuda.stream(s):
                for j in range(3):       # settle allocations outside capture
                    self._one(j)
            torch.cuda.current_stream(env.device).wait_stream(s)
            self.graph = torch.cuda.CUDAGraph()
            # thread-local capture mode: with a process group alive, RCCL's watchdog thread queries the events of
            # earlier collectives while this thread captures -- under the default (global) mode that query is an
            # error inside ProcessGroupNCCL and aborts the process (seen once in three runs with forced collectives)
            quiesce_collectives(torch)
            with torch.cuda.graph(self.graph, capture_error_mode="thread_local"):
                for j in range(chunk):
                    self._one(j)

    def _one(self, j):
        if self.produced is not None:
            self.produced.copy_(self.actions[j % self.ring])
            self.env.step(self.produced)
        else:
            self.env.step(self.actions[j % self.ring])
        if self.post is not None:
            self.post()

    def run(self, count):
        done = 0
        if self.graph is not None:
            while count - done >= self.chunk:
                self.graph.replay()
                done += self.chunk
        while done < count:
            self._one(self.pos)
            self.pos += 1
            done += 1


class PipeStepper:
    """The double-buffered half-batch schedule of gym_copter_amd.sharded.HalfBatchPipeline without a policy:
    per step, half h is waited for (where a learner would consume its rows) and stepped again, so its packed
    all-gather is on the links while the other half steps.  hipGraph replay of `chunk` steps when `graph`."""

    def __init__(self, torch, pipe, actions, device, graph, chunk):
        self.torch, self.pipe, self.actions, self.chunk_steps = torch, pipe, actions, chunk
        self.half, self.ring = actions.shape[1] // 2, actions.shape[0]
        self.graph, self.pos = None, 0
        if graph:
            cur = torch.cuda.current_stream(device)
            s = torch.cuda.Stream(device=device)
            s.wait_stream(cur)
            with torch.cuda.stream(s):
                self.chunk(3)
            cur.wait_stream(s)
            self.graph = torch.cuda.CUDAGraph()
            quiesce_collectives(torch)
            with torch.cuda.graph(self.graph, capture_error_mode="thread_local"):      # (see Stepper)
                self.chunk(chunk)

    def chunk(self, count, start=0):
        pipe, hn = self.pipe, self.half
        for j in range(count):
            row = self.actions[(start + j) % self.ring]
            for h in (0, 1):
                if j:
                    pipe.wait(h)
                pipe.step_async(h, row[h * hn:(h + 1) * hn])
        pipe.wait(0)
        pipe.wait(1)

    def run(self, count):
        done = 0
        if self.graph is not None:
            while count - done >= self.chunk_steps:
                self.graph.replay()
                done += self.chunk_steps
        if done < count:
            self.chunk(count - done, self.pos)
            self.pos += count - done


class ServedSession:
    """K env steps as ONE served session (gym_copter_amd.CopterVecEnv.serve_*): cs_serve_begin leaves a
    persistent env kernel running, `body(s)` feeds step s (a policy kernel, or submit + collect kernels),
    cs_serve_end joins.  begin / end are eager calls (HIP may serialise the branches of one hipGraph, so the env
    kernel must not sit in the same graph as its feeders); the K feeder launches are captured once and
    replayed against every session.  run(count) runs count // K sessions back to back."""

    def __init__(self, torch, env, K, body, ring, use_graph=True):
        self.torch, self.env, self.K, self.body, self.ring = torch, env, K, body, ring
        self.graph = None
        self.session()                        # the first session allocates the rings
        torch.cuda.synchronize()
        self.check()
        if use_graph:
            # captured while no session is open (torch's capture starts with a device-wide synchronize, which an
            # open session's env kernel would sit out until its timeout); valid for every session of this shape
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph, capture_error_mode="thread_local"):
                for s in range(K):
                    body(s)
            self.session()
            torch.cuda.synchronize()
            self.check()

    def session(self):
        self.env.serve_begin(self.K, ring=self.ring, timeout=1.0)
        if self.graph is not None:
            self.graph.replay()
        else:
            for s in range(self.K):
                self.body(s)
        self.env.serve_end(wait=False)

    def check(self):
        st = self.env.serve_status()
        if st != (self.K, self.K, 0):
            raise RuntimeError("served session incomplete: (min, max, timeouts) = %r of %d steps" % (st, self.K))

    def run(self, count):
        for _ in range(max(1, count // self.K)):
            self.session()


class LaunchFloor:
    """A hipGraph chain of 100 in-place adds on a small tensor: what one dependent launch costs on this box when
    the kernel does next to nothing."""

    def __init__(self, torch, probe, device):
        s = torch.cuda.Stream(device=device)
        s.wait_stream(torch.cuda.current_stream(device))
        with torch.cuda.stream(s):
            probe.add_(1.0)
        torch.cuda.current_stream(device).wait_stream(s)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph, capture_error_mode="thread_local"):
            for _ in range(100):
                probe.add_(1.0)

    def run(self, count):
        for _ in range(count // 100):
            self.graph.replay()


class Hip:
    """What main() and Timer ask of the device runtime: the device of this rank, synchronisation, HIP events on the
    current (= launch) stream, the process-group backend.  There is no CPU variant in the product: the constructor
    refuses a machine without a HIP device.  (tests/test_bench_launcher.py swaps a double in -- inside its own worker
    processes only -- to run the N > 1 line assembly under gloo at the world size of the target node.)"""
    backend = "nccl"                       # = RCCL on ROCm: the DATA path (the all-gather legs), opened when first needed
    control_backend = "gloo"               # barriers and the MAX over ranks of a region's wall time: CPU-side, so that no
    #                                        ProcessGroupNCCL watchdog thread is alive while the collective-free legs capture
    #                                        and replay hipGraphs (its event queries racing a capture aborted about one cold
    #                                        start in fifteen with ONE rank: tests/gpu_util.py -- eight ranks would not get by)
    graphs = True                          # hipGraph capture / replay of the step launches

    def __init__(self, torch, local):
        assert torch.cuda.is_available(), "bench.py needs a HIP device (no CPU fallback)"
        self.torch = torch
        self.device = torch.device("cuda", local)
        torch.cuda.set_device(self.device)

    def synchronize(self):
        self.torch.cuda.synchronize()

    def stamp(self):
        ev = self.torch.cuda.Event(enable_timing=True)
        ev.record()
        return ev

    def elapsed_s(self, ev0, ev1):
        return ev0.elapsed_time(ev1) * 1e-3

    def cus_and_clock_hz(self):
        props = self.torch.cuda.get_device_properties(self.device)
        cus = int(getattr(props, "multi_processor_count", 256) or 256)
        return cus, float(getattr(props, "clock_rate", 0) or 0) * 1e3 or PEAK_ENGINE_CLOCK_HZ     # (kHz -> Hz)

    def init_group(self, dist):
        # (a rank that dies outside the guarded gather leg must not leave the others at a barrier for the default 30 minutes)
        import datetime
        dist.init_process_group(self.control_backend,
                                timeout=datetime.timedelta(seconds=float(os.environ.get("BENCH_CONTROL_TIMEOUT_S", 600))))

    def open_collectives(self, dist):
        """The RCCL communicator over all ranks (a group of its own beside the gloo control plane)."""
        return dist.new_group(backend=self.backend)


class Timer:
    """barrier + synchronize on both sides of a region; host clock for the wall time (MAX over
    ranks), HIP events on the launch stream for the device time of the same region."""

    def __init__(self, hip, dist):
        self.hip, self.torch, self.dist, self.device = hip, hip.torch, dist, hip.device

    def barrier(self):
        if self.dist is not None:
            self.dist.barrier()
        self.hip.synchronize()

    def region(self, runner, count):
        torch = self.torch
        self.barrier()
        t0 = time.perf_counter()
        ev0 = self.hip.stamp()
        runner.run(count)
        ev1 = self.hip.stamp()
        self.hip.synchronize()
        t1 = time.perf_counter()
        self.barrier()
        wall = t1 - t0
        if self.dist is not None:
            t = torch.tensor([wall], dtype=torch.float64)            # (host tensor: the control plane is gloo)
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
            wall = float(t.item())
        return wall, self.hip.elapsed_s(ev0, ev1)

    def measure(self, runner, steps, warmup, min_region_s, regions, quantum=1, sampler=None):
        """-> dict: the median of `regions` timed regions of R * steps steps each (R chosen so that
        a region holds >= min_region_s of GPU work), the spread over the regions, and the bare single pass of
        `steps` steps.  sampler (a ClockSampler): clocks / power / temperature of the device before, during and
        after the timed regions."""
        steps = max(quantum, steps // quantum * quantum)
        if warmup > 0:
            runner.run(max(quantum, warmup // quantum * quantum))
        w1, e1 = self.region(runner, steps)                       # also the calibration pass
        if self.dist is not None:                                 # every rank must pick the same R
            t = self.torch.tensor([e1], dtype=self.torch.float64)
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
            e1 = float(t.item())
        reps = max(1, int(min_region_s / max(e1, 1e-9) + 0.999))
        walls, evs = [], []
        before = sampler.snapshot() if sampler is not None else None
        if sampler is not None:
            sampler.start()
        for _ in range(max(1, regions)):
            w, e = self.region(runner, steps * reps)
            walls.append(w)
            evs.append(e)
        during = sampler.stop() if sampler is not None else None
        after = sampler.snapshot() if sampler is not None else None
        k = sorted(range(len(walls)), key=lambda i: walls[i])[len(walls) // 2]     # the median region
        total = steps * reps
        sev = sorted(evs)
        return {"steps": steps, "repeats": reps, "regions": len(walls), "wall_s": walls[k],
                "s_per_step": walls[k] / total, "launch_s": evs[k] / total,
                "launch_s_best": min(evs) / total,
                "launch_us_regions": {"min": sev[0] / total * 1e6, "median": sev[len(sev) // 2] / total * 1e6,
                                      "max": sev[-1] / total * 1e6, "all": [e / total * 1e6 for e in evs]},
                "clocks": None if sampler is None else {"before": before, "during": during, "after": after},
                "single_pass": {"steps": steps, "wall_ms": w1 * 1e3, "ms_per_step": w1 / steps * 1e3,
                                "note": "one bare pass (the K steps, or one hipGraph of them) incl. graph-launch + "
                                        "synchronisation overhead"}}


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_topology():
    """What this process may actually use of the host's CPUs: the affinity set, the cgroup CPU quota (v2 cpu.max or
    v1 cfs quota), and how many physical cores / SMT siblings the affinity set spans."""
    visible = os.cpu_count() or 1
    try:
        aff = sorted(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        aff = list(range(visible))
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = float(q) / float(per)
    except (OSError, ValueError):
        try:
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0 and per > 0:
                quota = q / per
        except (OSError, ValueError):
            pass
    cores = set()
    for c in aff:
        try:
            sib = open("/sys/devices/system/cpu/cpu%d/topology/thread_siblings_list" % c).read().strip()
        except OSError:
            sib = str(c)
        cores.add(sib)
    usable = len(aff)
    if quota is not None:
        usable = max(1, min(usable, int(quota + 0.999)))
    return {"cores_visible": visible, "cores_in_affinity_set": len(aff), "cgroup_quota_cores": quota,
            "cores_usable": usable, "physical_cores_in_set": len(cores),
            "smt_siblings_in_set": len(cores) < len(aff)}


SAMPLER_CHILD = r"""
import glob, json, os, select, sys, time
FIELDS = (("sclk_MHz", "freq1_input", 1e-6), ("mclk_MHz", "freq2_input", 1e-6), ("power_W", "power1_input", 1e-6),
          ("power_W", "power1_average", 1e-6), ("temp_junction_C", "temp2_input", 1e-3),
          ("temp_memory_C", "temp3_input", 1e-3), ("temp_edge_C", "temp1_input", 1e-3))
files, node = {}, None
def snap():
    out = {}
    for name, (path, scale) in files.items():
        try:
            out[name] = round(float(open(path).read()) * scale, 1)
        except (OSError, ValueError):
            pass
    return out
def say(x):
    sys.stdout.write(json.dumps(x) + "\n"); sys.stdout.flush()
sampling, samples = False, []
while True:
    r, _, _ = select.select([sys.stdin], [], [], 0.005 if sampling else None)
    if r:
        line = sys.stdin.readline()
        if not line:
            break
        cmd = line.split()
        if not cmd:
            continue
        if cmd[0] == "node":
            for h in sorted(glob.glob("/sys/bus/pci/devices/%s/hwmon/hwmon*" % cmd[1])):
                node = h
                for name, fn, scale in FIELDS:
                    path = os.path.join(h, fn)
                    if name not in files and os.access(path, os.R_OK):
                        files[name] = (path, scale)
                break
            say({"node": node, "fields": sorted(files)})
        elif cmd[0] == "snap":
            say(snap())
        elif cmd[0] == "start":
            sampling, samples = True, []
        elif cmd[0] == "stop":
            sampling = False
            out = {"samples": len(samples)}
            for name in files:
                v = sorted(x[name] for x in samples if name in x)
                if v:
                    out[name] = {"min": v[0], "median": v[len(v) // 2], "max": v[-1]}
            say(out)
        elif cmd[0] == "quit":
            break
    elif sampling:
        samples.append(snap())
"""


class ClockSampler:
    """Clocks, power and temperature of ONE device from its sysfs hwmon node: snapshot() once, or start() / stop()
    around timed regions (one sample every 5 ms; stop() -> min / median / max).  freq1 = sclk, freq2 = mclk, power1 =
    socket power, temp2 = junction, temp3 = memory.  The reader is a CHILD PROCESS started before this process
    touches the GPU (plain file reads, no GPU call of its own): a sampling THREAD in this process was measured to
    cost the timed loop up to 6 % -- it takes the interpreter lock from the thread that replays the hipGraphs
    (Hover3D 262 144: 7.49 instead of 7.04 us per step, profiles/r05_ab_r4_vs_r5.txt).  sysfs sclk is the firmware's
    reading; the clock the shader actually runs at under a float64 load is measured in-kernel by cs_clock_probe
    (MI355X_MICROARCH.md: "sysfs pp_dpm_sclk is not the test")."""

    def __init__(self):
        self.node, self.fields = None, []
        try:
            self.child = subprocess.Popen([sys.executable, "-c", SAMPLER_CHILD], stdin=subprocess.PIPE,
                                          stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, bufsize=1)
        except Exception:
            self.child = None

    def _ask(self, line, reply=True):
        if self.child is None or self.child.poll() is not None:
            return None
        try:
            self.child.stdin.write(line + "\n")
            self.child.stdin.flush()
            return json.loads(self.child.stdout.readline()) if reply else None
        except Exception:
            return None

    def attach(self, pci_address):
        r = self._ask("node %s" % pci_address) or {}
        self.node, self.fields = r.get("node"), r.get("fields", [])
        return bool(self.fields)

    def available(self):
        return bool(self.fields)

    def snapshot(self):
        return self._ask("snap") or {}

    def start(self):
        self._ask("start", reply=False)

    def stop(self):
        return self._ask("stop") or {}

    def close(self):
        self._ask("quit", reply=False)
        try:
            if self.child is not None:
                self.child.wait(timeout=2)
        except Exception:
            pass


def _scalar_port(args):
    """`seconds` of the scalar NumPy port on one core -> (steps, elapsed)."""
    task, law, seconds, seed = args
    import numpy as np
    from oracle.refcpu import TaskOracle
    rng = np.random.default_rng(seed)
    o = TaskOracle(task)
    o.reset(rng=rng)
    steps = 0
    t0 = time.perf_counter()
    while True:
        for _ in range(500):
            if law == "uniform":
                a = rng.uniform(-1, 1, 4)
            elif law == "near_hover":
                a = HOVER * (1 + 0.01 * rng.standard_normal(4))
            else:
                a = 1.625e-2 * np.ones(4)
            _, _, done, _, _ = o.step(a)
            steps += 1
            if done:
                o.reset(rng=rng)
        dt = time.perf_counter() - t0
        if dt >= seconds:
            return steps, dt


def cpu_baseline(task, law, seconds):
    """Scalar NumPy port of the reference (one env per object, same NumPy call structure):
    one core, bounded sample; the same on every host core at once (one process per core, each
    stepping its own env); plus the vectorised NumPy oracle as an extra row.  Runs BEFORE the
    process touches the GPU or RCCL, so that forking the workers is safe."""
    import numpy as np
    from oracle.refvec import VecOracle
    rng = np.random.default_rng(0)
    steps, dt = _scalar_port((task, law, seconds, 0))
    # the 1-core row for both action laws of the headline workloads
    other = "const" if law != "const" else "uniform"
    o_steps, o_dt = _scalar_port((task, other, 2.0, 1))
    scalar = steps / dt
    # All usable cores, one process and one env each.  The pool is sized by what this process may USE -- the affinity
    # set capped by the cgroup CPU quota -- not by os.cpu_count() (the host's thread count: round 4 started 256
    # processes on a box that ran ~15 of them at a time).  If the pool still does not scale (a shared host), it is
    # halved until the aggregate is at least half of pool x one-core rate: `cores` is the pool that scaled.
    topo = cpu_topology()
    all_cores = None
    if topo["cores_usable"] > 1:
        import multiprocessing as mp
        per = 2.0
        tried = []
        try:
            procs = topo["cores_usable"]
            while True:
                with mp.get_context("fork").Pool(procs) as pool:
                    res = pool.map(_scalar_port, [(task, law, per, 100 + i) for i in range(procs)])
                agg = sum(r[0] for r in res) / max(r[1] for r in res)
                eff = agg / (scalar * procs)
                tried.append({"processes": procs, "value": agg, "scaling_efficiency": eff})
                if eff >= 0.5 or procs <= 2 or len(tried) >= 5:
                    break
                procs, per = max(2, procs // 2), 2.0
            # the pool that scaled (the last one tried) if there is one, else the largest aggregate seen
            best = tried[-1] if tried[-1]["scaling_efficiency"] >= 0.5 else max(tried, key=lambda t: t["value"])
            all_cores = {"value": best["value"], "unit": "env-steps/s", "cores": best["processes"],
                         "per_process_rate": best["value"] / best["processes"],
                         "scaling_efficiency": best["scaling_efficiency"],
                         "arithmetic": "scaling_efficiency = value / (one-core value x cores)",
                         "pools_tried": tried, "largest_aggregate_seen": max(t["value"] for t in tried),
                         "sample": "%d processes x %.1f s, one env each (pool = usable cores: affinity set %d, cgroup "
                                   "quota %s, %d visible)" % (best["processes"], per, topo["cores_in_affinity_set"],
                                                              topo["cgroup_quota_cores"], topo["cores_visible"])}
        except Exception as e:      # a baseline, never a reason to lose the GPU measurement
            all_cores = {"error": repr(e), "pools_tried": tried}
    nv = 65536
    v = VecOracle(task, nv, store_mode="float32", autoreset=1, seed=1)
    v.reset()
    acts = rng.uniform(-1, 1, (nv, 4))
    t0 = time.perf_counter()
    k = 0
    while time.perf_counter() - t0 < 2.0:
        v.step(acts)
        k += 1
    vec = nv * k / (time.perf_counter() - t0)
    return {"value": scalar, "unit": "env-steps/s", "cores": 1, "kind": "port",
            "sample": "oracle/refcpu.py TaskOracle (scalar NumPy, reference call structure), %s, "
                      "'%s' actions, %d steps in %.1f s on 1 of %d usable host cores (%d visible)"
                      % (task, law, steps, dt, topo["cores_usable"], topo["cores_visible"]),
            "one_core_other_law": {"actions": other, "value": o_steps / o_dt, "unit": "env-steps/s",
                                   "sample": "%d steps in %.1f s" % (o_steps, o_dt)},
            "all_cores": all_cores, "cpu_model": _cpu_model(), "cpu_topology": topo,
            "vectorised_numpy": {"value": vec, "unit": "env-steps/s", "cores": 1,
                                 "sample": "oracle/refvec.py VecOracle, %d envs x %d steps" % (nv, k)}}


def kernel_source_hash():
    """sha256 (16 hex digits) over the device sources of the step kernels: what profiles/traffic.json is stamped
    with (scripts/collect_profiles.py), so that a PMC figure is never reported for kernels it was not taken on."""
    import glob
    import hashlib
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "gym_copter_amd", "csrc")
    for f in sorted(glob.glob(os.path.join(csrc, "dev_*.h")) + [os.path.join(csrc, "copterstep_kernels.hip"),
                                                                  os.path.join(csrc, "copterstep_internal.h")]):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def load_stamped(name):
    """profiles/<name> (a JSON written by scripts/collect_profiles.py) if it was measured on THIS tree's kernel
    sources, else (None, why)."""
    path = os.path.join(ROOT, "profiles", name)
    if not os.path.exists(path):
        return None, "profiles/%s is absent" % name
    try:
        tj = json.load(open(path))
    except Exception as e:
        return None, "profiles/%s unreadable: %r" % (name, e)
    here = kernel_source_hash()
    if tj.get("kernel_source_sha16") != here:
        return None, ("profiles/%s was measured on kernel sources %s, this tree has %s: withheld"
                      % (name, tj.get("kernel_source_sha16"), here))
    return tj, ("profiles/%s: rocprofv3 PMC of these kernels at commit %s (kernel sources %s = this tree's) -- NOT "
                "measured by this run" % (name, tj.get("commit"), here))


def issue_bound(valu_per_wave_step, n_envs, s_per_step, cus, clock_hz, measured_clock_hz=None):
    """The instruction-issue floor of one step: every SIMD holds ceil(waves / SIMDs) wavefronts, each of which executes
    `valu_per_wave_step` vector instructions at 4 cycles apiece -> (floor in us, floor / measured = frac).  4 cycles per
    float64 vector instruction and SIMD is what profiles/r05_ubench_f64.txt measures with proven co-residency (4.19-4.45
    summed over two or more wavefronts, 4.6-4.9 for a lone one).  `frac` prices the floor at the PEAK engine clock (a
    roofline is a peak); `frac_at_measured_clock` at the clock this device held under a DENSE float64 FMA load in this
    run (cs_clock_probe) -- the lowest clock it runs at, so the two fractions bracket the fraction at the kernel's own
    clock."""
    simds = cus * SIMDS_PER_CU
    waves = (n_envs + LANES_PER_WAVE - 1) // LANES_PER_WAVE
    per_simd = (waves + simds - 1) // simds
    floor_s = per_simd * valu_per_wave_step * ISSUE_CYCLES / clock_hz
    out = {"bound": "valu_f64_issue", "valu_per_wavefront_step": valu_per_wave_step, "wavefronts_per_simd": per_simd,
           "floor_us": floor_s * 1e6, "achieved_us": s_per_step * 1e6, "frac": floor_s / s_per_step,
           "ceiling_env_steps_per_s": n_envs / floor_s, "simds": simds, "clock_GHz": clock_hz / 1e9,
           "arithmetic": "wavefronts_per_simd x valu_per_wavefront_step x 4 cycles / clock = floor; frac = floor / achieved"}
    if measured_clock_hz:
        out["measured_f64_clock_GHz"] = measured_clock_hz / 1e9
        out["frac_at_measured_clock"] = out["frac"] * clock_hz / measured_clock_hz
        out["measured_clock_is"] = ("cs_clock_probe: the clock under a DENSE float64 FMA stream -- the lowest this device clocks; a real "
                                    "kernel with other instructions and memory phases in it holds more (the K-step kernels at 65 536 "
                                    "envs: 2.13 GHz in-kernel, profiles/r05_kstep_phase_stamps.txt), so frac_at_measured_clock is an "
                                    "UPPER bound of the fraction at the kernel's own clock and frac (peak clock) the lower one")
    return out


def kernel_span_child(task, n, law, substeps, timeout=180):
    """The step kernel's OWN duration per launch (first wavefront start -> last wavefront end on the chip-wide 100 MHz
    clock), from the span build of the library (make span: two clock reads and one 16-byte store per wavefront, phases
    not serialised, no profiler) in a CHILD process -- started before this process initialises the GPU and finished
    before it does.  -> the JSON tools/kernel_span.py prints, or {"error": ...}."""
    lib = os.path.join(ROOT, "gym_copter_amd", "csrc", "build", "libcopterstep_span.so")
    tool = os.path.join(ROOT, "tools", "kernel_span.py")
    if not os.path.exists(lib):
        return {"error": "span build absent (make -C gym_copter_amd/csrc span)"}
    if any(k.startswith("ROCPROF") for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", ""):
        return {"error": "skipped under a profiler"}
    try:
        p = subprocess.run([sys.executable, tool, task, str(n), law, str(substeps)], capture_output=True, text=True,
                           timeout=timeout)
        if p.returncode != 0:
            return {"error": "rc %d: %s" % (p.returncode, p.stderr[-300:])}
        return json.loads(p.stdout.strip().splitlines()[-1])
    except Exception as e:
        return {"error": repr(e)}


def graph_chunk_for(steps, graph_chunk):
    """Launches per captured hipGraph: whole passes of the K steps, as many as fit `graph_chunk`
    (a 20-step graph pays its ~4 us replay boundary every 20 launches: 0.2 us per step)."""
    steps = max(1, steps)
    return steps * max(1, graph_chunk // steps) if steps < graph_chunk else graph_chunk


def roofline_block(task, n, launch_s, state, traffic=None, traffic_source=None):
    achieved = ALGO_BYTES[task] * n / launch_s / 1e9
    return {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic, "traffic_source": traffic_source,
            "kernel": "step_kernel<%s,%s>" % (task, state), "launch_us": launch_s * 1e6,
            "algorithmic_bytes_per_launch": ALGO_BYTES[task] * n}


def run_config(torch, timer, gca, a, task, n, law, substeps, device, rank, steps, warmup, ring, min_region_s,
               regions, use_graph=True, sampler=None):
    """One (task, batch, action law) point of the sweep: own env, own action ring, own graph."""
    env = gca.CopterVecEnv(task=task, num_envs=n, device=device.index, seed=1234,
                           autoreset_mode="next_step", state_dtype=a.state, substeps=substeps,
                           env_id_base=rank * n)
    actions = make_actions(torch, law, ring, n, device, 1234 + rank)
    env.reset()
    chunk = graph_chunk_for(steps, a.graph_chunk)
    st = Stepper(torch, env, actions, use_graph, chunk)
    m = timer.measure(st, steps, warmup, min_region_s, regions, quantum=chunk if use_graph else 1, sampler=sampler)
    env.close()
    del st, actions, env
    torch.cuda.empty_cache()
    return m


def _short(x, digits=6):
    """Floats to `digits` significant digits (recursively): the compact line is read by people and a bounded parser."""
    if isinstance(x, float):
        return float("%.*g" % (digits, x))
    if isinstance(x, dict):
        return {k: _short(v, digits) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_short(v, digits) for v in x]
    return x


def compact_line(full, full_path=None, budget=LINE_BUDGET, summary_budget=SUMMARY_BUDGET):
    """The ONE stdout line (a str, no newline) from the full record `assemble()` builds: the contract keys, `roofline`
    and `cpu_baseline` with their listed sub-keys only, and a `summary` of at most `summary_budget` bytes -- trimmed,
    least important key first, until the whole line is within `budget` bytes.  Never raises on a missing key: a leg
    that did not run is simply absent."""
    pick = lambda d, *ks: {k: d[k] for k in ks if isinstance(d, dict) and k in d and d[k] is not None}
    cfg = full.get("config", {})
    line = {k: full.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step",
                                     "higher_is_better", "scaling", "vs_baseline", "dtype", "data")}
    line["config"] = dict(pick(cfg, "workload", "envs_per_gpu", "total_envs", "task", "actions", "state_words",
                               "substeps", "parallelism"))
    line["config"]["workload"] = str(line["config"].get("workload", ""))[:200]
    line["status"] = full.get("status", "ok")
    line.update(pick(full, "timed_steps_total", "timed_region_s"))
    rf = full.get("roofline", {})
    r = {k: rf.get(k) for k in ("bound", "achieved", "peak", "unit", "frac", "traffic")}
    r.update(_short(pick(rf, "kernel", "launch_us", "kernel_span_us", "kernel_frac", "algorithmic_bytes_per_launch",
                         "resident")))
    r["source"] = ("achieved = algorithmic_bytes_per_launch / launch_us (HIP events on the launch stream, median region, "
                   "dependent-launch gap included); kernel_span_us = the kernel alone (span build, child process); "
                   "traffic = %s" % ("profiles/traffic.json (rocprofv3 PMC, stamped with this tree's kernel sources)"
                                     if rf.get("traffic") is not None else "withheld (no PMC figure for these sources)"))
    line["roofline"] = r
    cb = full.get("cpu_baseline")
    if isinstance(cb, dict):
        c = pick(cb, "value", "unit", "cores", "kind", "cpu_model")
        c["sample"] = str(cb.get("sample", ""))[:220]
        if isinstance(cb.get("all_cores"), dict) and "value" in cb["all_cores"]:
            c["all_cores"] = _short(pick(cb["all_cores"], "value", "cores"))
        if isinstance(cb.get("vectorised_numpy"), dict):
            c["vectorised_numpy"] = _short(pick(cb["vectorised_numpy"], "value"))
        if isinstance(cb.get("one_core_other_law"), dict):      # both action laws on one core
            c["one_core_other_law"] = _short(pick(cb["one_core_other_law"], "actions", "value"))
        line["cpu_baseline"] = c
    if full.get("rccl") is not None:
        line["rccl"] = full["rccl"]
    line.update(pick(full, "value_with_packed_allgather", "ms_per_step_with_packed_allgather", "packed_allgather_note"))
    line.update(_short(pick(full, "value_constant_thrust", "ms_per_step_constant_thrust", "constant_thrust_note"), 9))
    if "value_constant_thrust" in full and full["value_constant_thrust"] is None:
        line["value_constant_thrust"] = None
    if "value_with_packed_allgather" in full and full["value_with_packed_allgather"] is None:
        line["value_with_packed_allgather"] = None
    if full_path:
        line["full_record"] = full_path
    # the summary: most important first; dropped from the END until it fits
    sm = full.get("summary", {}) if isinstance(full.get("summary"), dict) else {}
    order = ("sweep_frac", "k_step_us", "config5", "config5_launch_us", "k_step_issue_frac", "f64_load_clock_GHz",
             "with_packed_allgather", "sweep_4m_launch_us_min_median_max", "served_us", "fused_caller_policy_us")
    summary = {k: _short(sm[k], 4) for k in order if sm.get(k) not in (None, {}, [])}
    size = lambda o: len(json.dumps(o, separators=(", ", ": ")))
    keys = list(summary)
    while keys and size(summary) > summary_budget:
        del summary[keys.pop()]
    line["summary"] = summary
    text = json.dumps(line)
    while len(text) > budget and line["summary"]:
        del line["summary"][list(line["summary"])[-1]]
        text = json.dumps(line)
    if len(text) > budget:          # (cannot happen with the fields above; a hard stop rather than a long line)
        for k in ("cpu_baseline", "roofline"):
            line[k].pop("sample", None)
            line[k].pop("source", None)
        text = json.dumps(line)
    assert len(text) <= budget, "bench.py: compact line is %d bytes (> %d)" % (len(text), budget)
    return text


def emit(full, real_stdout, full_out):
    """Full record -> the side file and stderr; compact line -> the saved stdout descriptor (ONE line, last)."""
    path = full_out or os.path.join(ROOT, "gpurun_out", "bench_full.json")
    shown = None
    try:
        os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
        with open(path, "w") as f:
            json.dump(full, f)
            f.write("\n")
        shown = os.path.relpath(path, ROOT) if os.path.abspath(path).startswith(ROOT + os.sep) else path
    except OSError as e:
        print("bench.py: full record not written to %s: %r" % (path, e), file=sys.stderr, flush=True)
    try:
        sys.stderr.write("bench.py full record: " + json.dumps(full) + "\n")
        sys.stderr.flush()
    except Exception:
        pass
    os.write(real_stdout, (compact_line(full, shown) + "\n").encode())


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    a = parse(argv)
    if needs_self_launch(a.gpus, os.environ):
        sys.exit(self_launch(a, argv))

    if os.environ.get("BENCH_DUMP_STACKS_AFTER_S"):     # diagnostic: where is the host when a run seems stuck?
        import faulthandler
        faulthandler.dump_traceback_later(float(os.environ["BENCH_DUMP_STACKS_AFTER_S"]), repeat=True, file=sys.stderr)
    # stdout carries ONE JSON line and nothing else: from here on file descriptor 1 is stderr (RCCL's version
    # banner, any library chatter, C stdio buffers flushed at exit), and the line goes to the saved descriptor
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    if a.gpus != world:
        sys.exit("bench.py: --gpus %d but WORLD_SIZE=%d" % (a.gpus, world))
    cpu = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        # before torch.cuda / RCCL are initialised in this process (the baseline forks workers)
        cpu = cpu_baseline(a.task, a.actions, a.cpu_seconds)

    span = None
    if rank == 0 and world == 1 and not a.no_span and not a.no_graph and a.state == "float32" and not a.produce_actions:
        # the kernel-only figure of the headline: a child on the span build, before this process touches the GPU
        span = kernel_span_child(a.task, a.envs, a.actions, a.substeps)

    # the clock / power / temperature reader: a child process, started before this one touches the GPU
    sampler_proc = ClockSampler() if rank == 0 and world == 1 and not a.no_clock_sampling else None

    import torch
    dist = None
    launched = world > 1 or "TORCHELASTIC_RUN_ID" in os.environ      # by torch.distributed.run
    if launched or a.gather or a.default_gather_leg:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if not launched:           # --gather on one GPU: a real RCCL group of ONE rank
            os.environ.setdefault("MASTER_PORT", str(_free_port()))
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        if world == 1:
            # world 1: gym_copter_amd.sharded would shortcut the all-gathers to a return / a device copy;
            # the legs below are there to exercise RCCL, so ask for the collective
            os.environ.setdefault("COPTERSTEP_FORCE_COLLECTIVE", "1")
    # more ranks than devices on this node (a test of the N > 1 path on the hardware there is): the ranks share the
    # devices round-robin -- the collective-free legs and the gloo control plane work; RCCL refuses two ranks on one
    # device, so the all-gather leg reports its failure ("status": "degraded") instead of a figure
    ndev = max(1, int(torch.cuda.device_count()))
    if local >= ndev:
        print("bench.py: rank %d: LOCAL_RANK %d on a node with %d device(s): sharing device %d" % (rank, local, ndev, local % ndev),
              file=sys.stderr, flush=True)
        local = local % ndev
    hip = Hip(torch, local)
    device = hip.device
    timer = Timer(hip, dist)
    rccl = None
    with stdout_to_stderr():
        if dist is not None:
            if os.environ.get("MASTER_ADDR", "127.0.0.1") in ("127.0.0.1", "localhost", "::1"):
                # one node (the contract): RCCL's bootstrap sockets belong on the loopback interface -- probing
                # whatever else the box has can cost minutes; the data path is xGMI either way
                os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")
                os.environ.setdefault("NCCL_IB_DISABLE", "1")
                os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
            hip.init_group(dist)
        timer.barrier()      # (N > 1: the control group is set up here, outside every timed region)
    coll = {"group": None}

    def collectives():
        """The RCCL communicator, opened when the first all-gather leg needs it -- after the collective-free legs have been
        timed -- and what RCCL itself saw: an all-reduce of ones over it (the driver can check N ranks took part)."""
        nonlocal rccl
        if coll["group"] is None:
            with stdout_to_stderr():
                g = hip.open_collectives(dist)
                ones = torch.ones(1, device=device, dtype=torch.float32)
                dist.all_reduce(ones, group=g)
                hip.synchronize()
            coll["group"] = g
            RCCL_OPEN[0] = True
            rccl = {"backend": dist.get_backend(g), "world_size": dist.get_world_size(g),
                    "ranks_seen": int(round(float(ones.item())))}
        return coll["group"]

    import gym_copter_amd as gca
    n = a.envs
    min_region_s = a.min_region_ms * 1e-3
    cus, clock_hz = hip.cus_and_clock_hz()
    pmc, pmc_src = load_stamped("pmc_counts.json")
    pmc = pmc or {}
    env = gca.CopterVecEnv(task=a.task, num_envs=n, device=local, seed=1234,
                           autoreset_mode="next_step", state_dtype=a.state,
                           substeps=a.substeps, env_id_base=rank * n)
    actions = make_actions(torch, a.actions, a.ring, n, device, 1234 + rank)
    env.reset()
    # clocks / power / temperature of THIS device beside the timings (sysfs hwmon of the device's PCI address), and the
    # shader clock it holds under a float64 vector load (cs_clock_probe) -- what the issue bounds are ALSO priced at
    sampler = None
    try:
        if sampler_proc is not None and sampler_proc.attach(env.pci_address()):
            sampler = sampler_proc
    except Exception:
        sampler = None
    f64_clock = {}
    try:
        for w in (1, 4):
            f64_clock[w] = env.clock_probe(w)
    except Exception as e:
        f64_clock = {"error": repr(e)}
    # streams for the served legs, created BEFORE any hipGraph is captured: on MI355X / ROCm 7 a stream created once
    # graphs have been instantiated is served several times more slowly by the hardware scheduler (DESIGN.md section 8;
    # the env kernel's own stream is created with the context for the same reason)
    served_side = [torch.cuda.Stream(device=device) for _ in (0, 1)] if a.served > 0 else None
    use_graph = hip.graphs and not a.no_graph
    chunk = graph_chunk_for(a.steps, a.graph_chunk)
    stepper = Stepper(torch, env, actions, use_graph, chunk, produce=a.produce_actions)
    m = timer.measure(stepper, a.steps, a.warmup, min_region_s, a.regions, quantum=chunk if use_graph else 1,
                      sampler=sampler)
    total_envs = n * world
    value = total_envs / m["s_per_step"]

    extra = {}
    state = {"deadline_hit": False}      # (set when the default N > 1 gather leg hits its deadline or fails: "status": "degraded")

    def assemble():
        """The ONE JSON line from what has been measured so far (also called by the deadline of the default N > 1
        gather leg)."""
        traffic, tsrc = None, None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                here = kernel_source_hash()
                if tj.get("kernel_source_sha16") != here:
                    tsrc = ("profiles/traffic.json was measured on kernel sources %s, this tree has %s: traffic withheld"
                            % (tj.get("kernel_source_sha16"), here))
                else:
                    traffic = tj.get("%s_%d" % (a.task, n))
                    if traffic is not None:
                        tsrc = ("profiles/traffic.json: rocprofv3 PMC (2*FETCH_SIZE + WRITE_SIZE, separate passes) of this "
                                "kernel and batch at commit %s (kernel sources %s = this tree's) -- NOT measured by this run"
                                % (tj.get("commit"), here))
            except Exception:
                traffic = None

        out = {
            "metric": "env-steps/sec Lander3D at 65 536 envs" if (a.task, n) == ("lander3d", 65536)
                      else "env-steps/sec %s at %d envs" % (a.task, n),
            "value": value, "unit": "env-steps/s", "n_gpus": world, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": m["s_per_step"] * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "repeats": m["repeats"], "regions": m["regions"], "timed_steps_per_region": m["steps"] * m["repeats"],
            # `steps` echoes --steps; what was actually timed (the driver's consistency arithmetic should use these):
            "timed_steps_total": m["steps"] * m["repeats"] * m["regions"],
            "timed_region_s": m["wall_s"],
            "timing": "median of %d regions of %d x %d steps (%s), each bracketed by barrier + "
                      "synchronize, MAX over ranks" % (m["regions"], m["repeats"], m["steps"],
                                                       "hipGraphs of %d launches" % chunk if use_graph else "eager launches"),
            "single_pass": m["single_pass"],
            "config": {"workload": "%s, %d envs/GPU, %s actions, auto-reset NEXT_STEP, %s state words, "
                                   "dt=%g x %d substeps, %s" % (a.task, n, a.actions, a.state,
                                                              1.0 / (100 * a.substeps), a.substeps,
                                                              "hipGraph replay of %d-step chunks" % chunk
                                                              if use_graph else "eager launches"),
                       "envs_per_gpu": n, "total_envs": total_envs, "task": a.task,
                       "actions": a.actions, "state_words": a.state, "substeps": a.substeps,
                       "action_ring": a.ring,
                       "actions_produced_by_a_preceding_kernel": bool(a.produce_actions),
                       "parallelism": "env-shard x%d" % world},
            "roofline": roofline_block(a.task, n, m["launch_s"], a.state, traffic, tsrc),
        }
        rf = out["roofline"]
        rf["resident"] = "infinity_cache" if n <= 1048576 else "hbm"
        lr = m["launch_us_regions"]
        rf["launch_us_min_median_max"] = [lr["min"], lr["median"], lr["max"]]
        # clocks / power / temperature of the device around and during the headline's timed regions (sysfs hwmon), and
        # the shader clock it holds under a float64 vector load with 1 and 4 wavefronts per SIMD (cs_clock_probe)
        out["clocks"] = {"headline": m.get("clocks"),
                         "f64_load_clock_GHz": ({str(k): v / 1e9 for k, v in f64_clock.items()}
                                                if "error" not in f64_clock else f64_clock),
                         "peak_engine_clock_GHz": clock_hz / 1e9,
                         "source": (sampler.node if sampler is not None else "no readable hwmon node for this device"),
                         "note": "sclk_MHz is the firmware's sysfs reading; f64_load_clock_GHz is delta s_memtime / delta "
                                 "s_memrealtime inside a float64 FMA kernel, by wavefronts per SIMD"}
        out["status"] = "degraded" if state["deadline_hit"] else "ok"
        # the kernel alone, without the dependent-launch gap that `frac` contains: the span build in a child process
        if span is not None and "kernel_span_ns" in span:
            ks = span["kernel_span_ns"]["median"] * 1e-9
            rf["kernel_span_us"] = ks * 1e6
            rf["kernel_frac"] = ALGO_BYTES[a.task] * n / ks / 1e9 / HBM_PEAK_GBPS
            rf["kernel_span"] = {"p10_us": span["kernel_span_ns"]["p10"] * 1e-3, "min_us": span["kernel_span_ns"]["min"] * 1e-3,
                                 "launches": span["launches"],
                                 "gap_between_eager_launches_us": span["gap_between_eager_launches_ns"]["median"] * 1e-3,
                                 "note": "first wavefront start -> last wavefront end on the 100 MHz chip clock, span build "
                                         "of the library (make span) in a child process run before this one touched the GPU "
                                         "(tools/kernel_span.py; eager launches, no profiler).  frac = with the "
                                         "inter-kernel gap of a dependent hipGraph chain, kernel_frac = the kernel alone"}
        elif span is not None:
            rf["kernel_span"] = span
        key = "%s_%d%s" % (a.task, n, "_substeps%d" % a.substeps if a.substeps > 1 else "")
        v = pmc.get("valu_per_wavefront", {}).get(key) if a.state == "float32" else None
        if v:
            rf["issue"] = dict(issue_bound(v, n, m["launch_s"], cus, clock_hz, f64_clock.get(1 if n <= 65536 else 4)), source=pmc_src)
        out.update(extra)
        if cpu is not None:
            out["cpu_baseline"] = cpu
        if rccl is not None:
            out["rccl"] = rccl
        # a compact digest as the LAST key (a log tail keeps the end of the line) and inside `roofline` (a contract key)
        pick = lambda d, *ks: {k: d[k] for k in ks if isinstance(d, dict) and k in d}
        digest = {"timed_steps_total": out["timed_steps_total"], "timed_region_s": out["timed_region_s"],
                  "headline": pick(out["roofline"], "launch_us", "frac", "kernel_span_us", "kernel_frac", "resident"),
                  "sweep_frac": {"%s_%d_%s" % (e.get("task"), e.get("envs", 0), e.get("actions")): round(e["frac"], 4)
                                 for e in extra.get("sweep", []) if "frac" in e},
                  "sweep_resident": {"%s_%d_%s" % (e.get("task"), e.get("envs", 0), e.get("actions")): e["resident"].split(",")[0][:24]
                                     for e in extra.get("sweep", []) if "resident" in e},
                  "config5": [pick(b, "bound", "frac") for b in extra.get("config5", {}).get("bounds", [])],
                  "config5_launch_us": extra.get("config5", {}).get("launch_us"),
                  "sweep_4m_launch_us_min_median_max": {
                      "%s_%d_%s" % (e.get("task"), e.get("envs", 0), e.get("actions")): [round(x, 2) for x in e["launch_us_min_median_max"]]
                      for e in extra.get("sweep", []) if e.get("envs", 0) >= 4194304 and "launch_us_min_median_max" in e},
                  "f64_load_clock_GHz": {str(k): round(v / 1e9, 3) for k, v in f64_clock.items()} if "error" not in f64_clock else None,
                  "status": "degraded" if state["deadline_hit"] else "ok",
                  "k_step_us": {k: round(extra[k]["us_per_step"], 3) for k in ("step_many", "rollout_pid", "rollout_random", "rollout_policy_linear")
                                if "us_per_step" in extra.get(k, {})},
                  "k_step_issue_frac": {k: round(extra[k]["roofline"]["frac"], 3)
                                        for k in ("step_many", "rollout_pid", "rollout_random", "rollout_policy_linear")
                                        if isinstance(extra.get(k, {}).get("roofline", {}).get("frac"), float)},
                  "with_packed_allgather": pick(extra, "value_with_packed_allgather", "ms_per_step_with_packed_allgather"),
                  "served_us": {k: round(extra[k]["us_per_step"], 3)
                                for k in ("served_producers_ahead",)
                                if "us_per_step" in extra.get(k, {})},
                  "fused_caller_policy_us": {k: extra["rollout_custom"][k] for k in
                                             ("closed_loop_law_with_state", "linear_policy_44_weights", "replay_policy")
                                             if k in extra.get("rollout_custom", {})},
                  "rccl": rccl}
        out["summary"] = digest
        return out

    # --gather: all three collective legs, right after the headline.  N > 1 without the flag (the driver's command):
    # still ONE packed all-gather leg, so that the multi-GPU line shows value_with_packed_allgather beside the
    # collective-free value -- run LAST and under a deadline (below): the N > 1 collective path has only ever run on
    # one GPU and under gloo, and a leg that hangs must not cost the line
    def run_gather_legs(gather_legs, graph=None):
        from gym_copter_amd.sharded import ShardGather, PackedOutputs
        grp = collectives()
        graph = use_graph if graph is None else (graph and use_graph)
        gather = ShardGather(n, world, group=grp)           # the product's RCCL all-gather of the obs rows

        def gather_leg(post, bind=None):
            """step + collective per step, hipGraph-captured like the headline when asked to and RCCL allows it."""
            if bind is not None:
                bind()
            mode = "graph" if graph else "eager"
            try:
                st = Stepper(torch, env, actions, graph, chunk, post=post)
            except Exception as e:      # capture of the collective refused: time it eagerly instead
                torch.cuda.synchronize()
                mode = "eager (capture failed: %s)" % type(e).__name__
                st = Stepper(torch, env, actions, False, chunk, post=post)
            g = timer.measure(st, a.steps, min(a.warmup, 50), min_region_s, a.regions,
                              quantum=chunk if mode == "graph" else 1)
            return g, mode
        modes = {}
        if "obs" in gather_legs:
            # observation rows only: a contiguous [n, obs_dim] buffer of their own (the default outputs are packed rows)
            sep = (torch.empty((n, env.obs_dim), device=device), torch.empty(n, device=device),
                   torch.empty(n, dtype=torch.uint8, device=device), torch.empty(n, dtype=torch.uint8, device=device))
            g2, modes["obs"] = gather_leg(lambda: gather("obs", sep[0]), bind=lambda: env.bind_outputs(*sep))
            extra["value_with_allgather"] = total_envs / g2["s_per_step"]
            extra["ms_per_step_with_allgather"] = g2["s_per_step"] * 1e3
        # everything a global learner needs (obs, reward, both flags) in ONE all-gather: the kernel
        # writes its outputs straight into the packed per-rank buffer
        pk = PackedOutputs(n, env.obs_dim, world, device, group=grp)
        g3, modes["packed"] = gather_leg(pk.all_gather, bind=lambda: env.bind_outputs(pk.obs, pk.reward, pk.term, pk.trunc))
        extra["value_with_packed_allgather"] = total_envs / g3["s_per_step"]
        extra["ms_per_step_with_packed_allgather"] = g3["s_per_step"] * 1e3
        extra["packed_allgather_bytes_per_rank"] = pk.nbytes
        if n % 2 == 0 and "pipelined" in gather_legs:
            # double-buffered half-batches: each half's step + packed all-gather on its own
            # stream, so one half's collective is on the links while the other half steps
            from gym_copter_amd.sharded import HalfBatchPipeline
            pipe = HalfBatchPipeline(task=a.task, total_envs=total_envs, gather="all", group=grp, device=local, seed=1234,
                                     autoreset_mode="next_step", state_dtype=a.state, substeps=a.substeps)
            pipe.reset()
            mode4 = "graph" if graph else "eager"
            try:
                pr = PipeStepper(torch, pipe, actions, device, graph, chunk)
            except Exception as e:
                torch.cuda.synchronize()
                mode4 = "eager (capture failed: %s)" % type(e).__name__
                pr = PipeStepper(torch, pipe, actions, device, False, chunk)
            g4 = timer.measure(pr, a.steps, min(a.warmup, 50), min_region_s, a.regions,
                               quantum=chunk if mode4 == "graph" else 1)
            extra["value_with_pipelined_allgather"] = total_envs / g4["s_per_step"]
            extra["ms_per_step_with_pipelined_allgather"] = g4["s_per_step"] * 1e3
            modes["pipelined"] = mode4
            del pr
            pipe.close()
        extra["allgather_launch_mode"] = modes
        extra["allgather_is_a_collective"] = bool(world > 1 or os.environ.get("COPTERSTEP_FORCE_COLLECTIVE") == "1")

    if a.gather and dist is not None:
        run_gather_legs(("obs", "packed", "pipelined"))

    def k_step_leg(name, k, call, bytes_step, note):
        class Runner:
            def run(self, count):
                for _ in range(count // k):
                    call()
        g = timer.measure(Runner(), max(k, a.steps // k * k), 2 * k, min_region_s, a.regions, quantum=k)
        extra[name] = {"steps_per_launch": k, "value": total_envs / g["s_per_step"], "unit": "env-steps/s",
                       "us_per_step": g["launch_s"] * 1e6, "repeats": g["repeats"],
                       "algorithmic_bytes_per_env_step": bytes_step,
                       "achieved_GBps": bytes_step * n / g["launch_s"] / 1e9, "note": note}
        # these kernels keep the env in registers: what bounds them is float64 instruction issue, not memory
        v = pmc.get("valu_per_wavefront_step", {}).get(name) if (a.task, n, a.state) == ("lander3d", 65536, "float32") else None
        if v:
            extra[name]["roofline"] = dict(issue_bound(v, n, g["launch_s"], cus, clock_hz, f64_clock.get(1)), source=pmc_src)
        else:
            extra[name]["roofline"] = {"bound": "valu_f64_issue", "frac": None, "source": pmc_src}

    od = env.obs_dim
    if a.many > 0:
        # K steps per launch (env state stays in registers): same envs, same resident action ring
        k = min(a.many, actions.shape[0])
        block = actions[:k].contiguous()
        k_step_leg("step_many", k, lambda: env.step_many(block), 16 + 4 * od + 4 + 2 + 136.0 / k,
                   "cs_step_many: bit-identical to K single-step launches "
                   "(tests/test_gpu_stepping_forms.py::test_step_many_is_bit_identical_to_single_steps); "
                   "open-loop actions only, so it is reported beside, not as, the headline value")
    if a.pid > 0 and a.task == "lander3d":
        # closed loop: K steps per launch with the on-device PID heuristic choosing every action
        k = a.pid
        env.configure_pid()
        env.reset()
        k_step_leg("rollout_pid", k, lambda: env.rollout_pid(k), 4 * od + 4 + 2 + (136.0 + 384.0) / k,
                   "cs_rollout_pid: closed loop, upstream's PID landing heuristic evaluated on device "
                   "(tests/test_gpu_stepping_forms.py::test_rollout_pid_policy_is_bit_exact); episodes under "
                   "upstream's gains end by tilt after ~130 steps and auto-reset")
    if a.pid > 0 and a.full:
        # random policy on device: the headline's action law with no action tensor, K steps per launch
        k = a.pid
        env.reset()
        k_step_leg("rollout_random", k, lambda: env.rollout_random(k), 4 * od + 4 + 2 + 136.0 / k,
                   "cs_rollout_random: actions ~ U[-1,1)^4 drawn in the kernel (Philox, keyed by seed / env "
                   "id / episode / step; tests/test_gpu_stepping_forms.py::test_rollout_random_is_bit_exact)")
    if a.pid > 0 and world == 1 and a.full:     # (N = 1 only: a compile that failed on ONE rank would leave the others at a barrier)
        # the CALLER'S OWN policy fused into the K-step kernel, the Python route: a linear law given as HIP source,
        # compiled with hipcc here and now (gym_copter_amd.compile_policy), K closed-loop steps per launch
        try:
            from gym_copter_amd import compile_policy
            src = """
struct Policy {
  const float* params;
  float w[ACT * OBS + ACT];
  __device__ void load(uint32_t, bool) { for (int j = 0; j < ACT * OBS + ACT; ++j) w[j] = params[j]; }
  __device__ void store(uint32_t, bool) {}
  __device__ void operator()(const float (&obs)[OBS], uint32_t, int, bool, float (&a)[ACT]) const {
    for (int m = 0; m < ACT; ++m) { float s = w[ACT * OBS + m]; for (int j = 0; j < OBS; ++j) s += w[m * OBS + j] * obs[j]; a[m] = s; }
  }
};"""
            t0 = time.perf_counter()
            pol = compile_policy(env, src)       # cached in a directory only this user can write (policy_jit)
            compile_s = time.perf_counter() - t0
            ad = env.action_dim
            W = torch.zeros(ad * od + ad, device=device)
            W[ad * od:] = HOVER * 0.98                      # a slow descent; the sink rate and body rates fed back a little
            for mtr in range(ad):
                W[mtr * od + min(5, od - 1)] = 0.002
            k = a.pid
            env.reset()
            k_step_leg("rollout_policy_linear", k, lambda: env.rollout_policy(pol, k, W), 4 * od + 4 + 2 + 136.0 / k,
                       "env.rollout_policy: the caller's own policy (a linear law, %d weights, given as HIP source and "
                       "compiled by hipcc in %.1f s%s) fused into the K-step kernel (include/copterstep_rollout.h; "
                       "tests/test_gpu_stepping_forms.py::test_python_callers_policy_source_is_compiled_and_fused)"
                       % (ad * od + ad, compile_s, "" if compile_s > 0.5 else ", cached"))
        except Exception as e:              # an extra never costs the headline
            extra["rollout_policy_linear"] = {"error": repr(e)}

    def served_legs():
        # served stepping (cs_serve_*): ONE persistent env kernel per K-step session, the env state in
        # registers throughout; action rows in and result rows out as tagged 16-byte granules through device
        # memory.  Reported beside the headline with their own byte models (what crosses memory per env-step).
        k = a.served
        ap, op = (env.action_dim + 1) // 2, (od + 2) // 2
        wire = 2 * 16 * (ap + op)                                   # every granule pair is written once and read once

        def served_leg(name, body, ring, bytes_step, note, prepare=None):
            try:
                if prepare is not None:
                    prepare()
                env.reset()
                print("bench.py: served leg %s: building" % name, file=sys.stderr, flush=True)
                ses = ServedSession(torch, env, k, body, ring, use_graph if a.served_graph < 0 else bool(a.served_graph))
                ses.run(2 * k)                     # two sessions back to back, checked before anything is timed
                torch.cuda.synchronize()
                ses.check()
                print("bench.py: served leg %s: timing" % name, file=sys.stderr, flush=True)
                g = timer.measure(ses, max(k, a.steps // k * k), 2 * k, min_region_s, a.regions, quantum=k)
                torch.cuda.synchronize()
                ses.check()
                extra[name] = {"steps_per_session": k, "ring": ring, "value": total_envs / g["s_per_step"],
                               "unit": "env-steps/s", "us_per_step": g["launch_s"] * 1e6, "repeats": g["repeats"],
                               "bytes_per_env_step": bytes_step, "achieved_GBps": bytes_step * n / g["launch_s"] / 1e9,
                               "note": note}
                del ses
            except Exception as e:          # an extra never costs the headline
                extra[name] = {"error": repr(e)}
                try:
                    torch.cuda.synchronize()
                    env._lib.cs_serve_end(env._ctx, env._stream(), None)
                except Exception:
                    pass

        side = served_side

        def submit_two_streams(s):
            # no launch dependency between consecutive producers: even / odd steps on two streams
            cur = torch.cuda.current_stream(device)
            if s < 2:
                side[s].wait_stream(cur)
            with torch.cuda.stream(side[s % 2]):
                env.serve_submit(s, actions[s % actions.shape[0]])
            if s >= k - 2:
                cur.wait_stream(side[s % 2])
        served_leg("served_producers_ahead", submit_two_streams, 8, wire + 4 * env.action_dim,
                   "producers that run ahead of the env (open-loop rows submitted from two alternating streams, only ring "
                   "back-pressure; outputs left in the output ring for a device-side consumer): what the persistent env "
                   "kernel sustains when it never waits for a policy")
    if a.served > 0 and world == 1:
        served_legs()

    def fused_policy_leg():
        # the caller's OWN policy fused into the K-step kernel (include/copterstep_rollout.h): such a kernel lives in
        # the caller's translation unit, so the figure comes from the third-party-style test program, run as a child
        # (same GPU, this process idle meanwhile); it checks itself against cs_step_many / cs_step before it times
        exe = os.path.join(ROOT, "tests", "host", "rollout_policy_host")
        if not os.path.exists(exe):
            return
        try:
            p = subprocess.run([exe, "time"], capture_output=True, text=True, timeout=120)
            import re
            m = re.search(r"(\d+) envs, (\d+) steps per launch.*?cs_rollout_random ([\d.]+)\s+cs_step_many ([\d.]+)\s+"
                          r"custom replay policy ([\d.]+)\s+custom closed-loop policy with state ([\d.]+)\s+"
                          r"custom linear policy \(44 weights\) ([\d.]+)", p.stdout)
            mm = re.search(r"custom MLP actor .*? ([\d.]+) us per env step", p.stdout)
            if p.returncode != 0 or m is None or "rollout_policy_host: OK" not in p.stdout:
                extra["rollout_custom"] = {"error": "rc %d: %s" % (p.returncode, (p.stdout + p.stderr)[-300:])}
                return
            extra["rollout_custom"] = {
                "envs": int(m.group(1)), "steps_per_launch": int(m.group(2)), "unit": "us per env step",
                "same_run": {"cs_rollout_random": float(m.group(3)), "cs_step_many": float(m.group(4))},
                "replay_policy": float(m.group(5)), "closed_loop_law_with_state": float(m.group(6)),
                "linear_policy_44_weights": float(m.group(7)),
                **({"mlp_10_32_32_4_per_lane_policy_bound": float(mm.group(1))} if mm else {}),
                "note": "tests/host/rollout_policy_host.hip: caller-side device functors instantiated into the K-step "
                        "kernel in the caller's translation unit; eager launches timed with HIP events; verified "
                        "bit-identical to cs_step_many / a twin stepped with cs_step in the same run"}
        except Exception as e:              # an extra never costs the headline
            extra["rollout_custom"] = {"error": repr(e)}
    if a.full and a.pid > 0 and world == 1 and not any(k.startswith("ROCPROF") for k in os.environ) \
            and "rocprof" not in os.environ.get("LD_PRELOAD", ""):
        fused_policy_leg()
    if world > 1 and not a.no_sweep:
        # north_star: "constant-thrust and random-action workloads at 1, 2, 4 and 8 GPUs" -- at N = 1 the constant-thrust
        # point is in the sweep; at N > 1 it is this leg: the same shards under lander.py's MOTORVAL, timed like the
        # headline (barrier + synchronize, MAX over ranks).  Before the all-gather leg, which may end the run.
        try:
            gc_ = run_config(torch, timer, gca, a, a.task, n, "const", a.substeps, device, rank, 100, 100, a.ring,
                             min_region_s, 3, use_graph=use_graph)
            extra["value_constant_thrust"] = total_envs / gc_["s_per_step"]
            extra["ms_per_step_constant_thrust"] = gc_["s_per_step"] * 1e3
        except Exception as e:              # an extra never costs the headline
            extra["value_constant_thrust"] = None
            extra["constant_thrust_note"] = ("failed: %r" % (e,))[:300]
    if (world > 1 or a.default_gather_leg) and dist is not None and not a.gather:
        import threading

        def give_up():
            # the leg did not come back: the line goes out without it (rank 0) with "status": "degraded" (a hung
            # collective must not read as a clean run), every rank says so on stderr and leaves
            state["deadline_hit"] = True
            print("bench.py: rank %d: the packed all-gather leg did not return within its deadline -- status degraded"
                  % rank, file=sys.stderr, flush=True)
            extra["value_with_packed_allgather"] = None
            extra["packed_allgather_note"] = "the default packed all-gather leg did not finish within its deadline: abandoned"
            try:
                if rank == 0:
                    for _ in range(5):          # (the main thread may be adding a key at this very moment)
                        try:
                            emit(assemble(), real_stdout, a.full_out)
                            break
                        except RuntimeError:
                            time.sleep(0.05)
            finally:
                # non-zero: a launcher must see that this run did not finish all it set out to do (the line is out)
                os._exit(DEGRADED_EXIT)
        deadline_s = float(os.environ.get("BENCH_GATHER_DEADLINE_S", 150.0))
        dog = threading.Timer(deadline_s, give_up)
        dog.daemon = True
        dog.start()
        try:
            collectives()
            if os.environ.get("BENCH_TEST_HANG_GATHER") == "1":      # diagnostic: what a leg that never returns does to the line
                while True:
                    time.sleep(1.0)
            # EAGER launches: step + all-gather per step as a learner's loop issues them -- capturing a collective of N
            # ranks into a hipGraph is the one thing here that has never run on N > 1 devices (--gather captures them)
            run_gather_legs(("packed",), graph=False)
        except Exception as e:      # the leg failed: the line says so and the exit code too (after the line is out)
            extra["value_with_packed_allgather"] = None
            extra["packed_allgather_note"] = ("failed: %r" % (e,))[:300]
            state["deadline_hit"] = True
        dog.cancel()
    env.close()
    del stepper, env, actions
    torch.cuda.empty_cache()

    # ---- the same kernel at the other single-GPU points (driver-visible; N = 1 only) ----
    if not a.no_sweep and world == 1:
        sweep = []
        points = [("hover3d", 262144, "uniform", 16),            # BASELINE configs[2] first
                  # north_star's other action law at the headline size: lander.py's own constant thrust (MOTORVAL)
                  ("lander3d", 65536, "const", 64), ("lander3d", 65536, "near_hover", 64),
                  ("lander3d", 262144, "uniform", 16), ("lander3d", 262144, "near_hover", 16),
                  ("lander3d", 1048576, "uniform", 8),
                  ("hover3d", 1048576, "uniform", 8), ("lander3d", 4194304, "uniform", 4),
                  ("hover3d", 4194304, "uniform", 4)]
        if not a.full:       # the default run: BASELINE configs[2], north_star's other action law at the headline size (lander.py's
            # constant thrust), the headline's low-churn variant (SURVEY section 8(d) C2:
            # near-hover actions, episodes that do not finish -- under uniform actions one lasts 7.8 steps and every
            # wavefront resets lanes in every step) and the HBM-resident point of the headline's kernel
            points = [("hover3d", 262144, "uniform", 16), ("lander3d", 65536, "const", 64),
                      ("lander3d", 65536, "near_hover", 64), ("lander3d", 4194304, "uniform", 4)]
        tile_bytes = 5632 if a.state != "float64" else 10752     # copterstep_internal.h: make_layout (4 groups + FE + RET + EPH)

        def resident(task, nn, ring):
            """Where a step's working set lives between launches: the state tiles (5 632 B per 64 envs in the float32
            modes, 4 KiB of it touched per step), the action ring and the output buffers against the 256 MiB Infinity
            Cache.  A point that fits is served from that cache (its `frac` is algorithmic bytes over time against the
            HBM peak, as the contract defines it, but the bytes do not come from HBM); only the others are HBM-resident."""
            obs = 12 if task == "hover3d" else 10
            ws = (nn + 63) // 64 * tile_bytes + ring * nn * 16 + nn * (4 * obs + 4 + 2)
            tiles = (nn + 63) // 64 * tile_bytes
            where = ("infinity_cache" if ws <= INFINITY_CACHE_BYTES else
                     "hbm" if tiles > INFINITY_CACHE_BYTES else "state in the infinity cache, action ring + outputs streamed from / to hbm")
            return where, ws

        def spread(g):
            """min / median / max of the launch time over the timed regions, as roofline fractions too."""
            return g["launch_us_regions"]
        for task, nn, law, ring in points:
            if (task, nn, law) == (a.task, n, a.actions):
                continue
            try:
                # the HBM-resident points vary run to run and box to box by more than a layout change is worth
                # (VERDICT round 4, weak #4): seven regions each, min / median / max and the clocks beside them
                nreg = 7 if nn >= 4194304 else 3
                g = run_config(torch, timer, gca, a, task, nn, law, 1, device, rank, 100, 100, ring,
                               min_region_s, nreg, sampler=sampler)
                r = roofline_block(task, nn, g["launch_s"], a.state)
                where, ws = resident(task, nn, ring)
                lr = spread(g)
                algo = ALGO_BYTES[task] * nn / 1e3 / HBM_PEAK_GBPS       # us at the roof
                sweep.append({"task": task, "envs": nn, "actions": law, "ring": ring,
                              "value": nn / g["s_per_step"], "unit": "env-steps/s",
                              "ms_per_step": g["s_per_step"] * 1e3, "launch_us": r["launch_us"],
                              "achieved_GBps": r["achieved"], "frac": r["frac"], "repeats": g["repeats"],
                              "regions": g["regions"],
                              "launch_us_min_median_max": [lr["min"], lr["median"], lr["max"]],
                              "frac_min_median_max": [algo / lr["max"], algo / lr["median"], algo / lr["min"]],
                              "clocks": g["clocks"],
                              "resident": where, "working_set_bytes": ws,
                              "config": "BASELINE configs[2]" if (task, nn, law) == ("hover3d", 262144, "uniform") else None})
            except Exception as e:          # a sweep point never costs the headline
                sweep.append({"task": task, "envs": nn, "actions": law, "error": repr(e)})
                torch.cuda.empty_cache()

        def config5_point(nn, ring, nreg):
            """BASELINE configs[4] (dt = 1e-3, 10 Dynamics.setMotors calls per env step) at `nn` envs with its three
            bounds: HBM, float64 vector ALU (flops executed, by PMC) and float64 instruction issue."""
            nsub = 10
            g = run_config(torch, timer, gca, a, "lander3d", nn, "near_hover", nsub, device, rank, 100, 100,
                           ring, min_region_s, nreg, sampler=sampler)
            r = roofline_block("lander3d", nn, g["launch_s"], a.state)
            key = "lander3d_%d_substeps10" % nn
            bounds = [{"bound": "hbm", "achieved": r["achieved"], "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": r["frac"]}]
            flop = pmc.get("f64_flops_per_env_step", {}).get(key) if a.state == "float32" else None
            if flop:
                tf = flop * nn / g["launch_s"] / 1e12
                bounds.append({"bound": "valu_f64", "achieved": tf, "peak": F64_VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                               "frac": tf / F64_VALU_PEAK_TFLOPS, "flop_per_env_step": flop,
                               "note": "float64 vector ALU (no MFMA on this path).  flop_per_env_step is EXECUTED float64 "
                                       "arithmetic by PMC: (SQ_INSTS_VALU_ADD_F64 + MUL_F64 + TRANS_F64 + 2 x FMA_F64) "
                                       "per wavefront = per env (one lane each)", "source": pmc_src})
            else:
                bounds.append({"bound": "valu_f64", "frac": None, "source": pmc_src})
            v = pmc.get("valu_per_wavefront", {}).get(key) if a.state == "float32" else None
            if v:
                w = 1 if nn <= 65536 else 4
                bounds.append(dict(issue_bound(v, nn, g["launch_s"], cus, clock_hz, f64_clock.get(w)), source=pmc_src))
            else:
                bounds.append({"bound": "valu_f64_issue", "frac": None, "source": pmc_src})
            lr = g["launch_us_regions"]
            return {"workload": "lander3d, %d envs, near_hover actions, dt=0.001 x 10 substeps (BASELINE configs[4]%s)"
                                % (nn, "" if nn == 65536 else " at a batch that gives every SIMD many wavefronts"),
                    "task": "lander3d", "envs": nn, "actions": "near_hover", "substeps": nsub,
                    "value": nn / g["s_per_step"], "unit": "env-steps/s", "ms_per_step": g["s_per_step"] * 1e3,
                    "launch_us": r["launch_us"], "repeats": g["repeats"], "regions": g["regions"],
                    "launch_us_min_median_max": [lr["min"], lr["median"], lr["max"]], "clocks": g["clocks"],
                    "frac": r["frac"], "resident": resident("lander3d", nn, ring)[0], "bounds": bounds}
        try:
            extra["config5"] = config5_point(65536, a.ring, 3)
        except Exception as e:
            extra["config5"] = {"error": repr(e)}
            torch.cuda.empty_cache()
        # ... and where an instruction-issue bound is a bound: 16 wavefronts per SIMD (VERDICT round 4 #1b)
        try:
            if a.full:
                c5 = config5_point(1048576, 8, 3)
                c5["actions"] = "near_hover_substeps10"
                sweep.append(c5)
        except Exception as e:
            sweep.append({"task": "lander3d", "envs": 1048576, "actions": "near_hover_substeps10", "error": repr(e)})
            torch.cuda.empty_cache()

        # the K-step kernels with MANY wavefronts per SIMD (4 194 304 envs = 64 per SIMD): here their float64
        # instruction-issue bound binds.  Counts by PMC at THIS size (above 65 536 envs the observation rows go
        # through the LDS transpose: a different instantiation from the 65 536-env legs above).
        def k_step_large(name, nn, k, make_call, bytes_step):
            envL = gca.CopterVecEnv(task="lander3d", num_envs=nn, device=local, seed=1234, autoreset_mode="next_step",
                                    state_dtype=a.state, env_id_base=rank * nn)
            try:
                envL.reset()
                call = make_call(envL)

                class Runner:
                    def run(self, count):
                        for _ in range(count // k):
                            call()
                g = timer.measure(Runner(), 2 * k, 2 * k, min_region_s, 5, quantum=k, sampler=sampler)
                lr = g["launch_us_regions"]
                e = {"leg": name, "task": "lander3d", "envs": nn, "actions": name, "steps_per_launch": k,
                     "value": nn / g["s_per_step"], "unit": "env-steps/s", "launch_us": g["launch_s"] * 1e6,
                     "us_per_step": g["launch_s"] * 1e6, "repeats": g["repeats"], "regions": g["regions"],
                     "launch_us_min_median_max": [lr["min"], lr["median"], lr["max"]], "clocks": g["clocks"],
                     "algorithmic_bytes_per_env_step": bytes_step,
                     "achieved_GBps": bytes_step * nn / g["launch_s"] / 1e9,
                     "hbm_frac_of_its_own_bytes": bytes_step * nn / g["launch_s"] / 1e9 / HBM_PEAK_GBPS,
                     "resident": "hbm"}
                v = pmc.get("valu_per_wavefront_step", {}).get("%s_%d" % (name, nn)) if a.state == "float32" else None
                if v:
                    e["roofline"] = dict(issue_bound(v, nn, g["launch_s"], cus, clock_hz, f64_clock.get(4)), source=pmc_src)
                    e["frac"] = e["roofline"]["frac"]
                    e["bound"] = "valu_f64_issue"
                else:
                    e["roofline"] = {"bound": "valu_f64_issue", "frac": None, "source": pmc_src}
                return e
            finally:
                envL.close()
                del envL
                torch.cuda.empty_cache()
        if a.full and a.many > 0 and a.state == "float32":
            nn, k = 4194304, 16
            try:
                def mk_many(envL):
                    block = make_actions(torch, "uniform", k, nn, device, 99)
                    return lambda: envL.step_many(block)
                sweep.append(k_step_large("step_many", nn, k, mk_many, 16 + 4 * 10 + 4 + 2 + 136.0 / k))
            except Exception as e:
                sweep.append({"leg": "step_many", "envs": nn, "actions": "step_many", "task": "lander3d", "error": repr(e)})
                torch.cuda.empty_cache()
        if a.full and a.pid > 0 and a.state == "float32":
            nn, k = 4194304, 16
            try:
                def mk_pid(envL):
                    envL.configure_pid()
                    envL.reset()
                    return lambda: envL.rollout_pid(k)
                sweep.append(k_step_large("rollout_pid", nn, k, mk_pid, 4 * 10 + 4 + 2 + (136.0 + 384.0) / k))
            except Exception as e:
                sweep.append({"leg": "rollout_pid", "envs": nn, "actions": "rollout_pid", "task": "lander3d", "error": repr(e)})
                torch.cuda.empty_cache()
        extra["sweep"] = sweep

    if rank == 0 and use_graph and not a.no_sweep and a.full:
        # context for the headline's latency-bound figure: what ONE dependent launch costs on this box when the
        # kernel does next to nothing -- a hipGraph chain of in-place adds on 64 Ki floats (1 024 x 64 threads)
        try:
            probe = torch.zeros(65536, dtype=torch.float32, device=device)

            fl = Timer(hip, None).measure(LaunchFloor(torch, probe, device), 2000, 200, min_region_s, 3,
                                                    quantum=100)
            extra["dependent_launch_floor"] = {
                "us_per_launch": fl["launch_s"] * 1e6,
                "note": "hipGraph chain of torch in-place adds on 65 536 floats: the per-launch cost of a dependent "
                        "kernel that does almost nothing (inter-kernel gap + one read-modify-write through the "
                        "fabric); the step kernel's launch_us contains the same floor"}
        except Exception as e:
            extra["dependent_launch_floor"] = {"error": repr(e)}

    out = assemble()
    if sampler_proc is not None:
        sampler_proc.close()
    if rank == 0:
        emit(out, real_stdout, a.full_out)
    os.close(real_stdout)
    if dist is not None:
        hip.synchronize()
        profiled = "rocprof" in os.environ.get("LD_PRELOAD", "") or any(k.startswith("ROCPROF") for k in os.environ)
        if (launched or profiled) and not state["deadline_hit"]:     # (a communicator that failed is not torn down: leave)
            dist.destroy_process_group()
        if not profiled:               # (a profiler writes its files from exit handlers: leave normally under one)
            # leave without running the interpreter's shutdown: hipGraphs that captured collectives are still alive, and
            # freeing them once RCCL is (being) torn down has been seen to end the process with SIGSEGV / SIGABRT
            # (the watchdog thread during shutdown; a garbage collection at exit in tests/test_gpu_multigpu.py's RCCL
            # child) -- after the line is out, which a launcher would report as a failed rank.  The line is written,
            # the group is closed (launched runs), nothing is left to flush
            sys.stderr.flush()
            os._exit(DEGRADED_EXIT if state["deadline_hit"] else 0)
    if state["deadline_hit"]:
        sys.exit(DEGRADED_EXIT)


if __name__ == "__main__":
    main()
