#!/bin/bash
# usage (GPU box): bash tools/ab_bench4.sh name1 name2 ...  - A/B of library variants on BASELINE config 4 (1024 x 256x256)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for round in 1 2; do
for v in "$@"; do
  if [ "$v" = "main" ]; then unset WILDFIRE_LIB; else export WILDFIRE_LIB=$PWD/wildfire_control_python_amd/libwildfire_hip_$v.so; fi
  timeout -k 10 200 python bench.py --size 256 --batch 1024 --noop --ignite-ppm 10000 --steps 100 --warmup 10 --no-extras --no-cpu-baseline > gpurun_out/ab4_$v.log 2>&1
  tail -1 gpurun_out/ab4_$v.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', 'value %.4g  step_us %.2f  frac %.3f' % (d['value'], d['roofline']['launch_us'], d['roofline']['frac']))"
done; done
