#!/bin/bash
# usage (GPU box, through gpurun): tools/profile_steps.sh <tag> <preset> [steps]
#   kernel trace of the replayed training-step graph of one bench preset -> gpurun_out/<tag>_step_graph_kernel_summary_<preset>.txt
set -e
tag=$1; preset=$2; steps=${3:-30}
out=gpurun_out
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
d=$out/prof_bench_$preset
rm -rf $d
rocprofv3 --kernel-trace --output-format csv -d $d -- python3 bench.py --config $preset --steps $steps --warmup 5 --no-cpu-baseline --no-steady --no-roofline --also none > $d.log 2>&1 || { tail -20 $d.log; exit 1; }
python3 tools/prof_summary.py $(ls $d/*/*kernel_trace.csv | head -1) --steps $((steps / 3)) > $out/${tag}_step_graph_kernel_summary_${preset}.txt
head -60 $out/${tag}_step_graph_kernel_summary_${preset}.txt
