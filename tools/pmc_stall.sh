#!/bin/bash
# usage (on the GPU box): bash tools/pmc_stall.sh <tag>   -- wave stall / issue counters of the single-step kernel
# (kernels serialise under counter collection: this is the solo behaviour of one full-batch launch)
tag=${1:-x}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export WF_SPLIT=1
p1="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_BUSY_CU_CYCLES SQ_CYCLES"
p2="SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU SQ_INSTS_SALU"
for pass in 1 2; do
  eval pmc=\$p$pass
  rm -rf gpurun_out/stall_${tag}_$pass
  rocprofv3 --pmc $pmc --output-format csv -d gpurun_out/stall_${tag}_$pass -- python3 bench.py --steps 10 --warmup 2 --fused 0 --no-cpu-baseline > gpurun_out/stall_${tag}_$pass.log 2>&1
  python tools/pmc_summary.py gpurun_out/stall_${tag}_$pass | grep -A9 "wf_step_"
done
