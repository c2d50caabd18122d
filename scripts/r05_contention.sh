#!/bin/bash
# GPU box: the race-prone GPU tests (served stepping, streams, graphs, the fuzz) while ANOTHER process keeps the device
# busy -- timing perturbation as a race detector.   bash scripts/r05_contention.sh <tag> [N=3]
tag=${1:-r05cont}; N=${2:-3}
out=gpurun_out/$tag; mkdir -p $out
python3 - > $out/load.log 2>&1 <<'PY' &
import time, torch, sys
sys.path.insert(0, ".")
import gym_copter_amd as gca
env = gca.CopterVecEnv(task="lander3d", num_envs=262144, seed=1, autoreset_mode="next_step")
env.reset()
a = torch.rand((262144, 4), device="cuda") * 2 - 1
t0 = time.time(); n = 0
while time.time() - t0 < 420:
    for _ in range(200):
        env.step(a)
    torch.cuda.synchronize(); n += 200
    time.sleep(0.002)
print("load: %d steps" % n)
PY
LOAD=$!
sleep 5
for i in $(seq 1 $N); do
  timeout 600 python3 -m pytest ${FILES:-tests/test_gpu_round3.py tests/test_gpu_round4.py tests/test_gpu_round5.py tests/test_gpu_fuzz.py} -q -m gpu -p no:cacheprovider \
      -k "not bench and not rccl and not span" > $out/run_$i.log 2>&1
  echo "run $i rc=$? $(tail -1 $out/run_$i.log)" | tee -a $out/summary.txt
done
kill $LOAD 2>/dev/null; wait $LOAD 2>/dev/null
grep -h "^FAILED\|Error" $out/run_*.log | head -20 >> $out/summary.txt
tail -12 $out/summary.txt
