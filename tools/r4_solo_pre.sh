#!/bin/bash
mkdir -p gpurun_out/r4
out=gpurun_out/r4/solo_pre.txt
: > $out
run() { echo "## $*" >> $out; env "$@" 2>&1 | grep -v amdgpu.ids >> $out; }
for cfg in "64 4096" "128 1024" "256 1024" "32 16384" "50 8192" "30 16384"; do
  set -- $cfg
  run X=1 timeout -k 10 120 python3 tools/step_loop.py --size $1 --batch $2 --calls 300 --warm 300 || exit 1
  run X=1 timeout -k 10 120 python3 tools/step_loop.py --size $1 --batch $2 --calls 300 --warm 300 || exit 1
done
grep -v "^  File\|^    " $out | tail -60
