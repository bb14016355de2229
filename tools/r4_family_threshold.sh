#!/bin/bash
# round 4: where the tiled family (one-launch step) overtakes the generic one for small grids with W % 4 == 0
mkdir -p gpurun_out/r4
out=gpurun_out/r4/family_threshold.txt
: > $out
run() { echo "## $*" >> $out; env "$@" 2>&1 | grep -v amdgpu.ids >> $out; }
for cfg in "20 32768" "24 32768" "28 16384" "20 4096" "24 4096"; do
  set -- $cfg
  run X=1 timeout -k 10 120 python3 tools/step_loop.py --size $1 --batch $2 --calls 300 --warm 300 || exit 1
  run WF_FAMILY=tiled timeout -k 10 120 python3 tools/step_loop.py --size $1 --batch $2 --calls 300 --warm 300 || exit 1
done
grep -v "^  File\|^    " $out | tail -60
