#!/bin/bash
# Round profiles (run on the GPU box through gpurun): rocprofv3 --kernel-trace --stats of the LGSSM chain at the two
# BASELINE sizes and of bench.py, plus the HBM-traffic counters (FETCH_SIZE / WRITE_SIZE, separate passes).
# usage: tools/profile_round.sh r03        -> gpurun_out/<tag>_*.txt ; copy what should be judged into profiles/
set -e
tag=${1:-r03}
out=gpurun_out
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
stats() {   # stats <name> <cmd...>
  local name=$1; shift
  rm -rf $out/prof_$name
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_$name -- "$@" > $out/prof_$name.log 2>&1
  python3 tools/stats_filter.py $out/prof_$name > $out/${tag}_${name}_stats.txt
}
pmc() {     # pmc <name> <counter> <cmd...>
  local name=$1 ctr=$2; shift; shift
  rm -rf $out/pmc_${name}_$ctr
  rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $out/pmc_${name}_$ctr -- "$@" > $out/pmc_${name}_$ctr.log 2>&1
  python3 tools/pmc_summary.py $out/pmc_${name}_$ctr $ctr >> $out/${tag}_${name}_pmc.txt
}
C2="python3 tools/lgssm_chain.py --B 256 --T 50 --n 4 --iters 20"
C4="python3 tools/lgssm_chain.py --B 32 --T 100 --n 4 --K 7 --iters 20 --q-per-step"
C5="python3 tools/lgssm_chain.py --B 512 --T 200 --n 16 --iters 5"
C5S="python3 tools/lgssm_chain.py --B 512 --T 200 --n 16 --iters 5 --q-per-step"
stats lgssm_chain_c2 $C2
stats lgssm_chain_c4 $C4
stats lgssm_chain_c5 $C5
stats lgssm_chain_c5_switching $C5S
for n in lgssm_chain_c2 lgssm_chain_c4 lgssm_chain_c5 lgssm_chain_c5_switching; do rm -f $out/${tag}_${n}_pmc.txt; done
for ctr in FETCH_SIZE WRITE_SIZE; do
  pmc lgssm_chain_c2 $ctr $C2
  pmc lgssm_chain_c4 $ctr $C4
  pmc lgssm_chain_c5 $ctr $C5
  pmc lgssm_chain_c5_switching $ctr $C5S
done
for preset in c2 c4 c4-lstm c5 c5-lstm; do
  steps=30; [ "${preset#c5}" != "$preset" ] && steps=9
  tools/profile_steps.sh $tag $preset $steps > /dev/null
done
ls -la $out/${tag}_*
