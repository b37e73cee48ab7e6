#!/bin/bash
# usage (on the GPU box, through gpurun): tools/prof_stats.sh <name> <script.py> [args...]
#   rocprofv3 --kernel-trace --stats of one python program (the program itself after `--`: no wrapper hops)
#   -> gpurun_out/<name>_stats.txt (our kernels: average us, calls), program output in gpurun_out/prof_<name>.log
set -e
name=$1; shift
out=gpurun_out
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf $out/prof_$name
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_$name -- python3 "$@" > $out/prof_$name.log 2>&1 || { tail -20 $out/prof_$name.log; exit 1; }
python3 tools/stats_filter.py $out/prof_$name > $out/${name}_stats.txt
cat $out/${name}_stats.txt
