#!/bin/bash
# 1/2/4/8-GPU weak-scaling sweep of the train step on ONE node (one process per GPU, RCCL over xGMI):
#     tools/scale.sh [workload=config4] [steps=20] [warmup=5]
# prints audio-s/s per N and the efficiency against N x the 1-GPU value.  Needs as many visible GPUs as the largest N
# (the build's gpurun boxes expose one: the driver runs this sweep on an 8-GPU node at round end).
# bench.py replays the step from HIP graphs (train.Trainer.train_step_graphed) at every N; RTG_GRAPH=0 times the eager
# step (8 ranks then issue ~770 Python-side launches per step each).
set -e
WL=${1:-config4}; STEPS=${2:-20}; WARM=${3:-5}
cd "$(dirname "$0")/.."
export HSA_ENABLE_IPC_MODE_LEGACY=0
NGPU=$(python3 -c 'import torch; print(torch.cuda.device_count())')
base=""
for N in 1 2 4 8; do
  if [ "$N" -gt "$NGPU" ]; then echo "N=$N: only $NGPU GPU(s) visible, skipped"; continue; fi
  # (bench.py --gpus N starts its N ranks itself, one process per GPU; under torchrun it checks --gpus against WORLD_SIZE)
  out=$(python3 bench.py --gpus $N --steps $STEPS --warmup $WARM --workload $WL --no-cpu-baseline --no-roofline | tail -1)
  val=$(echo "$out" | python3 -c 'import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"])')
  v=${val% *}; ms=${val#* }
  # per-rank step times, tuner-pick digests (rank 0 tunes the block shapes and broadcasts its tables: the digests are equal)
  # and the exchange policy bench.py picked after timing both
  echo "$out" | python3 -c 'import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); c = d["config"]; pr = c.get("per_rank"); print("      per rank ms/step", pr["ms_per_step"], "tuner picks", pr["tuner_picks_digest"], "exchange", c["exchange"]["policy"], c["exchange"]["trial_ms_per_step"]) if pr else None'
  [ -z "$base" ] && base=$v
  eff=$(python3 -c "print(round($v / ($N * $base), 3))")
  echo "N=$N  $v audio-s/s  $ms ms/step  efficiency $eff"
done
