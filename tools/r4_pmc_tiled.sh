#!/bin/bash
# instruction counts per wave of the tiled step kernels (c4: two-launch form; mid_64: one-launch form)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r4/pmc_tiled; rm -rf $out; mkdir -p $out
export WF_NO_SPLIT=1
B="python3 bench.py --no-extras --no-cpu-baseline"
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAVES SQ_BUSY_CYCLES --output-format csv -d $out/c4 -- $B --size 256 --batch 1024 --noop --ignite-ppm 10000 --steps 60 --warmup 60 --min-ms 0 > $out/c4.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAVES SQ_BUSY_CYCLES --output-format csv -d $out/mid -- python3 bench.py --no-cpu-baseline --legs mid_64 --steps 100 --warmup 10 --fused 0 > $out/mid.log 2>&1
for d in c4 mid; do echo "== $d"; python tools/pmc_summary.py $out/$d 1 | grep -A12 "wf_step_tiled\|wf_tiled_pro"; done
