#!/bin/bash
# round 4: the one-launch tiled step - board rows fetched lazily vs at every row (libwildfire_hip_eager.so: -DWF_TILED_LAZY_BOARD=0),
# number of part launches, and the larger grids (several workgroups per environment)
mkdir -p gpurun_out/r4
out=gpurun_out/r4/solo_sweep.txt
: > $out
run() { echo "## $*" >> $out; env "$@" 2>&1 | grep -v amdgpu.ids >> $out; }
E=$PWD/wildfire_control_python_amd/libwildfire_hip_eager.so
for cfg in "64 4096" "128 1024" "256 1024"; do
  set -- $cfg
  run X=1 timeout -k 10 120 python3 tools/step_loop.py --size $1 --batch $2 --calls 300 --warm 300 || exit 1
  run WILDFIRE_LIB=$E timeout -k 10 120 python3 tools/step_loop.py --size $1 --batch $2 --calls 300 --warm 300 || exit 1
done
for sp in 1 2 4; do
  run WF_SPLIT=$sp timeout -k 10 120 python3 tools/step_loop.py --size 64 --batch 4096 --calls 300 --warm 300 || exit 1
  run WF_SPLIT=$sp timeout -k 10 120 python3 tools/step_loop.py --size 128 --batch 1024 --calls 300 --warm 300 || exit 1
done
run X=1 timeout -k 10 200 python3 bench.py --no-cpu-baseline --legs mid_64,c4 --steps 200 --warmup 20
cat $out
