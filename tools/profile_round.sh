#!/bin/bash
# Round profiles (run on the GPU box through gpurun): rocprofv3 --kernel-trace --stats of the LGSSM chain at the two
# BASELINE sizes and of bench.py, plus the HBM-traffic counters (FETCH_SIZE / WRITE_SIZE, separate passes).
# usage: tools/profile_round.sh r02        -> gpurun_out/<tag>_*.txt ; copy what should be judged into profiles/
set -e
tag=${1:-r02}
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
C5="python3 tools/lgssm_chain.py --B 512 --T 200 --n 16 --iters 5"
C5S="python3 tools/lgssm_chain.py --B 512 --T 200 --n 16 --iters 5 --q-per-step"
stats lgssm_chain_c2 $C2
stats lgssm_chain_c5 $C5
stats lgssm_chain_c5_switching $C5S
for n in lgssm_chain_c2 lgssm_chain_c5; do rm -f $out/${tag}_${n}_pmc.txt; done
pmc lgssm_chain_c2 FETCH_SIZE $C2
pmc lgssm_chain_c2 WRITE_SIZE $C2
pmc lgssm_chain_c5 FETCH_SIZE $C5
pmc lgssm_chain_c5 WRITE_SIZE $C5
rm -rf $out/prof_bench_c2
rocprofv3 --kernel-trace --output-format csv -d $out/prof_bench_c2 -- python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-steady --no-roofline > $out/prof_bench_c2.log 2>&1
python3 tools/prof_summary.py $(ls $out/prof_bench_c2/*/*kernel_trace.csv | head -1) --steps 10 > $out/${tag}_step_graph_kernel_summary_c2.txt
rm -rf $out/prof_bench_c5
rocprofv3 --kernel-trace --output-format csv -d $out/prof_bench_c5 -- python3 bench.py --config c5 --steps 6 --warmup 2 --no-cpu-baseline --no-steady --no-roofline > $out/prof_bench_c5.log 2>&1
python3 tools/prof_summary.py $(ls $out/prof_bench_c5/*/*kernel_trace.csv | head -1) --steps 4 > $out/${tag}_step_graph_kernel_summary_c5.txt
ls -la $out/${tag}_*
