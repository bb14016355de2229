#!/bin/bash
mkdir -p gpurun_out/r4
out=gpurun_out/r4/solo_occ.txt
: > $out
run() { echo "## $*" >> $out; env "$@" 2>&1 | grep -v amdgpu.ids >> $out; }
V=$PWD/wildfire_control_python_amd/libwildfire_hip_occ8.so
C4="python3 bench.py --no-extras --no-cpu-baseline --size 256 --batch 1024 --noop --ignite-ppm 10000 --steps 200 --warmup 20"
for lib in "" "$V"; do
  run WILDFIRE_LIB=$lib timeout -k 10 120 python3 tools/step_loop.py --size 256 --batch 1024 --calls 300 --warm 300
  run WILDFIRE_LIB=$lib WF_TILED_ROWS=16 timeout -k 10 120 python3 tools/step_loop.py --size 256 --batch 1024 --calls 300 --warm 300
  run WILDFIRE_LIB=$lib WF_TILED_ROWS=16 timeout -k 10 120 python3 tools/step_loop.py --size 256 --batch 1024 --calls 300 --warm 300 --noop
  run WILDFIRE_LIB=$lib timeout -k 10 120 python3 tools/step_loop.py --size 64 --batch 4096 --calls 300 --warm 300
  run WILDFIRE_LIB=$lib timeout -k 10 120 python3 tools/step_loop.py --size 128 --batch 1024 --calls 300 --warm 300
  run WILDFIRE_LIB=$lib timeout -k 10 120 python3 tools/step_loop.py --size 256 --batch 256 --calls 300 --warm 300
done
grep -A1 "^## " $out | grep -v "^--" | paste - - | sed "s/timeout -k 10 120 python3 tools\/step_loop.py//; s/--calls 300 --warm 300//; s/WILDFIRE_LIB=[^ ]*occ8.so/occ8/; s/WILDFIRE_LIB= /default /" | cut -c1-200
