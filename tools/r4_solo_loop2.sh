#!/bin/bash
mkdir -p gpurun_out/r4
out=gpurun_out/r4/solo_loop2.txt
: > $out
run() { echo "## $*" >> $out; env "$@" 2>&1 | grep -v amdgpu.ids >> $out; }
for cfg in "128 256" "128 512" "64 512" "64 1024" "64 2048" "256 128" "256 512" "100 600"; do
  set -- $cfg
  run X=1 timeout -k 10 120 python3 tools/step_loop.py --size $1 --batch $2 --n 16 --calls 60 --warm 320 || exit 1
  run WF_NO_SOLO_LOOP=1 timeout -k 10 120 python3 tools/step_loop.py --size $1 --batch $2 --n 16 --calls 60 --warm 320 || exit 1
done
grep -A1 "^## " $out | grep -v "^--" | paste - - | sed "s/timeout -k 10 120 python3 tools\/step_loop.py//; s/--calls 60 --warm 320//" | cut -c1-200
