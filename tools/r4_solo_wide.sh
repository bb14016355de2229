#!/bin/bash
mkdir -p gpurun_out/r4
out=gpurun_out/r4/solo_wide.txt
: > $out
run() { echo "## $*" >> $out; env "$@" 2>&1 | grep -v amdgpu.ids >> $out; }
for cfg in "128 1024" "256 1024" "256 256" "128 4096" "64 4096"; do
  set -- $cfg
  run X=1 timeout -k 10 120 python3 tools/step_loop.py --size $1 --batch $2 --calls 300 --warm 300 || exit 1
  run WF_NO_SOLO=1 timeout -k 10 120 python3 tools/step_loop.py --size $1 --batch $2 --calls 300 --warm 300 || exit 1
done
run WF_TILED_ROWS=16 timeout -k 10 120 python3 tools/step_loop.py --size 128 --batch 1024 --calls 300 --warm 300
run X=1 timeout -k 10 300 python3 bench.py --no-cpu-baseline --legs mid_64,c4 --steps 200 --warmup 20
grep -v "^  File\|^    " $out | tail -60
