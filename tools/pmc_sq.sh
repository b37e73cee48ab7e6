#!/bin/bash
# usage (GPU box, through gpurun): tools/pmc_sq.sh <name> <kernel substring> <script.py> [args...]
#   SQ counters of one kernel, two separate --pmc passes (no trace domains besides --kernel-trace), per dispatch
#   -> gpurun_out/<name>_sq.txt
set -e
name=$1; pat=$2; shift; shift
out=gpurun_out
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
: > $out/${name}_sq.txt
for set in "SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_INSTS_VMEM_RD" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_BUSY_CYCLES SQ_WAVES"; do
  d=$out/pmc_${name}
  rm -rf $d
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $d -- python3 "$@" > $d.log 2>&1 || { tail -5 $d.log; exit 1; }
  python3 tools/pmc_dispatches.py $d "$pat" 6 >> $out/${name}_sq.txt
  echo >> $out/${name}_sq.txt
done
cat $out/${name}_sq.txt
