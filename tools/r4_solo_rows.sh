#!/bin/bash
# round 4: rows per strip of the one-launch tiled step (WF_TILED_ROWS; d = the default, wf_tiled_solo_rows)
mkdir -p gpurun_out/r4
out=gpurun_out/r4/solo_rows.txt
: > $out
run() { echo "## $*" >> $out; env "$@" 2>&1 | grep -v amdgpu.ids >> $out; }
for cfg in "64 4096 0 16 8" "32 16384 0 16 8" "32 4096 0 16 8 4" "48 8192 0 16 12 8" "128 1024 0" "64 16384 0 16 8" "64 2048 0 16 8 4" "64 1024 0 16 8 4"; do
  set -- $cfg; size=$1; batch=$2; shift 2
  for tr in "$@"; do
    run WF_TILED_ROWS=$tr timeout -k 10 120 python3 tools/step_loop.py --size $size --batch $batch --calls 300 --warm 300 || exit 1
  done
done
grep -v "^  File\|^    " $out | tail -60
