#!/bin/bash
# A/B of the fused scalar loss head under the two-stream hipGraph: steady-state ms/step + per-queue timelines.
set -e
mkdir -p gpurun_out
for v in 0 1; do
  KVAE_LOSS_HEAD=$v python3 bench.py --steps 100 --warmup 5 --no-cpu-baseline 2> gpurun_out/lh$v.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('KVAE_LOSS_HEAD=$v', d['ms_per_step'], d['steady_state']['ms_per_step_median'], d['steady_state']['ms_per_step_min'])"
done
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in 0 1; do
  KVAE_LOSS_HEAD=$v rocprofv3 --kernel-trace --output-format csv -d gpurun_out/lh_prof$v -- python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-steady > gpurun_out/lh_prof$v.log 2>&1
  f=$(ls gpurun_out/lh_prof$v/*/*kernel_trace.csv | head -1)
  echo "== KVAE_LOSS_HEAD=$v"; python3 tools/prof_queues.py $f --steps 10 | head -12
done
