#!/bin/bash
# usage (GPU box): bash tools/profile_round.sh r04   -> gpurun_out/prof_<tag>/...
# rocprofv3 kernel-trace stats and HBM traffic counters (separate --pmc passes: FETCH_SIZE and WRITE_SIZE do not
# fit one pass; MI355X_MICROARCH.md) for
#   c3     65 536 x 14x14 (headline; default two-launch step, and one launch per step)
#   c4     1 024 x 256x256, 1 % pre-ignited, no-op (wf_step_tiled + wf_big_*; 268 MB of state: past the 256 MB Infinity Cache)
#   hbm    1 048 576 x 14x14 (wf_step_rows on 822 MB of state: true HBM streaming)
#   c2     4 096 x 10x10, no-op, 16 ticks per call (wf_step_ticks; the leg of the default bench line)
tag=${1:-r04}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/prof_$tag; rm -rf $out; mkdir -p $out
B="python3 bench.py --no-extras --no-cpu-baseline"
C4="--size 256 --batch 1024 --noop --ignite-ppm 10000 --steps 200 --warmup 20"
HB="--batch 1048576 --steps 120 --warmup 20"
C5="--batch 125000 --steps 400 --warmup 40"
C2="python3 bench.py --no-cpu-baseline --legs c2 --steps 100 --warmup 10 --fused 0"    # (the headline loop runs too; the wf_step_ticks rows are the leg's)
WIND="python3 bench.py --no-cpu-baseline --legs c3_wind --steps 100 --warmup 10 --fused 0"   # (wf_step_group<..,WIND> rows: the leg)
MID="python3 bench.py --no-cpu-baseline --legs mid_64 --steps 100 --warmup 10 --fused 0"     # (wf_step_tiled<.., SOLO> rows: the leg)
MID128="python3 bench.py --no-cpu-baseline --legs mid_128 --steps 100 --warmup 10 --fused 0"
python3 -c "import bench; print(bench.kernel_source_sha256())" > $out/kernel_source_sha256.txt
PM="--steps 40 --warmup 400"
# ---- kernel-trace stats
rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt_c3 -- python3 bench.py --no-cpu-baseline > $out/kt_c3.log 2>&1
WF_NO_SPLIT=1 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt_c3_single -- $B > $out/kt_c3_single.log 2>&1
# the default (split) headline loop alone: per-launch start / end of the two concurrent half-batch launches of every step
rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt_c3_split -- $B --steps 1500 --warmup 100 > $out/kt_c3_split.log 2>&1
python3 tools/kernel_trace_summary.py $out/kt_c3_split $out/c3_split_trace.json $out/c3_split_trace_excerpt.csv > $out/c3_split_trace.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt_c5 -- $B $C5 > $out/kt_c5.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt_c4 -- $B $C4 > $out/kt_c4.log 2>&1
WF_NO_SPLIT=1 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt_c4_single -- $B $C4 > $out/kt_c4_single.log 2>&1
WF_NO_SPLIT=1 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt_hbm_single -- $B $HB > $out/kt_hbm_single.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt_c2 -- $C2 > $out/kt_c2.log 2>&1
WF_NO_SPLIT=1 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt_mid -- $MID > $out/kt_mid.log 2>&1
# ---- counters (one launch per step: kernels serialise under --pmc anyway)
export WF_NO_SPLIT=1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $out/pmc_c3_$c -- $B $PM > $out/pmc_c3_$c.log 2>&1
  # (--min-ms 0: exactly ticks 60..120, the regime tools/c4_write_amp.py counts changed quads in)
  rocprofv3 --pmc $c --output-format csv -d $out/pmc_c4_$c -- $B $C4 --steps 60 --warmup 60 --min-ms 0 > $out/pmc_c4_$c.log 2>&1
  rocprofv3 --pmc $c --output-format csv -d $out/pmc_hbm_$c -- $B $HB --steps 40 --warmup 60 --min-ms 0 > $out/pmc_hbm_$c.log 2>&1
  rocprofv3 --pmc $c --output-format csv -d $out/pmc_c5_$c -- $B --batch 125000 $PM > $out/pmc_c5_$c.log 2>&1
  rocprofv3 --pmc $c --output-format csv -d $out/pmc_c2_$c -- $C2 > $out/pmc_c2_$c.log 2>&1
  rocprofv3 --pmc $c --output-format csv -d $out/pmc_wind_$c -- $WIND > $out/pmc_wind_$c.log 2>&1
  rocprofv3 --pmc $c --output-format csv -d $out/pmc_mid_$c -- $MID > $out/pmc_mid_$c.log 2>&1
  rocprofv3 --pmc $c --output-format csv -d $out/pmc_mid128_$c -- $MID128 > $out/pmc_mid128_$c.log 2>&1
done
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $out/pmc_c3_l2 -- $B $PM > $out/pmc_c3_l2.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $out/pmc_c3_insts -- $B $PM > $out/pmc_c3_insts.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $out/pmc_c2_insts -- $C2 > $out/pmc_c2_insts.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES --output-format csv -d $out/pmc_c3_active -- $B $PM > $out/pmc_c3_active.log 2>&1
unset WF_NO_SPLIT
# the DEFAULT headline step (three part launches of the workgroup-shared instantiation; under --pmc they run one after the other:
# the counters are per part launch, a step is their sum)
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $out/pmc_c3parts_$c -- $B $PM > $out/pmc_c3parts_$c.log 2>&1
done
python3 tools/c4_write_amp.py > $out/c4_write_amp.json 2> $out/c4_write_amp.err
# ---- the same bench un-profiled, for the agreement check
python3 bench.py > $out/bench_plain.json 2> $out/bench_plain.err
for d in kt_c3 kt_c3_single kt_c3_split kt_c5 kt_c4 kt_c4_single kt_hbm_single kt_c2 kt_mid; do echo "== $d"; find $out/$d -name "*kernel_stats.csv" | head -1 | xargs cat | grep -E "Name|wf_"; done
for d in $out/pmc_*; do [ -d $d ] && { echo "== $d"; python tools/pmc_summary.py $d 1 | grep -A12 "wf_"; }; done
tail -c 600 $out/bench_plain.json
