#!/bin/bash
# usage (GPU box): bash tools/profile_round.sh r01   -> gpurun_out/prof_<tag>/...
# kernel-trace stats of the default bench command, then HBM traffic counters in separate
# --pmc passes (FETCH_SIZE and WRITE_SIZE do not fit one pass; MI355X_MICROARCH.md).
tag=${1:-r01}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/prof_$tag; rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -- python3 bench.py --no-cpu-baseline > $out/bench_under_rocprof.log 2>&1
# one launch per step (no two-stream split): per-kernel durations that do not overlap each other
WF_NO_SPLIT=1 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt_single -- python3 bench.py --no-extras --no-cpu-baseline > $out/bench_single_under_rocprof.log 2>&1
# counters are collected with one launch per step too (kernels serialise under --pmc anyway)
export WF_NO_SPLIT=1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -- python3 bench.py --steps 40 --warmup 400 --no-extras --no-cpu-baseline > $out/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -- python3 bench.py --steps 40 --warmup 400 --no-extras --no-cpu-baseline > $out/pmc_write.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $out/pmc_l2 -- python3 bench.py --steps 40 --warmup 400 --no-extras --no-cpu-baseline > $out/pmc_l2.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $out/pmc_insts -- python3 bench.py --steps 40 --warmup 400 --no-extras --no-cpu-baseline > $out/pmc_insts.log 2>&1
unset WF_NO_SPLIT
# the same bench un-profiled, for the agreement check
python3 bench.py > $out/bench_plain.json 2> $out/bench_plain.err
find $out/kt -name "*kernel_stats.csv" | head -1 | xargs cat
find $out/kt_single -name "*kernel_stats.csv" | head -1 | xargs cat
for d in pmc_fetch pmc_write pmc_l2 pmc_insts; do python tools/pmc_summary.py $out/$d | grep -A10 "wf_step_"; done
tail -1 $out/bench_plain.json
