#!/bin/bash
# Where do the cycles of the headline kernel go?  SQ counters for one launch per step, and for 16 steps per launch.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r4/pmc_issue; rm -rf $out; mkdir -p $out
PA="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM"
PB="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC"
PC="SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_SMEM SQ_IFETCH"
for mode in "single 1 200 1" "multi 16 40 0"; do
  set -- $mode
  if [ $4 = 1 ]; then export WF_NO_SPLIT=1; else unset WF_NO_SPLIT; fi
  i=0
  for P in "$PA" "$PB" "$PC"; do
    i=$((i+1))
    rocprofv3 --pmc $P --output-format csv -d $out/$1_$i -- python3 tools/step_loop.py --n $2 --calls $3 > $out/$1_$i.log 2>&1
  done
done
for d in $out/single_* $out/multi_*; do [ -d $d ] && { echo "== $d"; python tools/pmc_summary.py $d 65536 | grep -A12 "wf_step"; }; done
