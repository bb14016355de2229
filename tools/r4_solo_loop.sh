#!/bin/bash
# round 4: n-step calls of the tiled family's one-launch step - the step loop inside the kernel against n launches (WF_NO_SOLO_LOOP=1)
mkdir -p gpurun_out/r4
out=gpurun_out/r4/solo_loop.txt
: > $out
run() { echo "## $*" >> $out; env "$@" 2>&1 | grep -v amdgpu.ids >> $out; }
for cfg in "64 4096" "128 1024" "32 16384" "256 256"; do
  set -- $cfg
  for n in 4 16; do
    run X=1 timeout -k 10 120 python3 tools/step_loop.py --size $1 --batch $2 --n $n --calls 60 --warm 320 || exit 1
    run WF_NO_SOLO_LOOP=1 timeout -k 10 120 python3 tools/step_loop.py --size $1 --batch $2 --n $n --calls 60 --warm 320 || exit 1
  done
done
run WF_SPLIT=1 timeout -k 10 120 python3 tools/step_loop.py --size 64 --batch 4096 --n 16 --calls 60 --warm 320
run WF_SPLIT=4 timeout -k 10 120 python3 tools/step_loop.py --size 64 --batch 4096 --n 16 --calls 60 --warm 320
run WF_TILED_ROWS=16 timeout -k 10 120 python3 tools/step_loop.py --size 64 --batch 4096 --n 16 --calls 60 --warm 320
grep -v "^  File\|^    " $out | tail -60
