#!/bin/bash
# usage (GPU box): bash tools/ab_bench.sh name1 name2 ...   (variants libwildfire_hip_<name>.so; "main" = default)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for round in 1 2; do
for v in "$@"; do
  if [ "$v" = "main" ]; then unset WILDFIRE_LIB; else export WILDFIRE_LIB=$PWD/wildfire_control_python_amd/libwildfire_hip_$v.so; fi
  timeout -k 10 200 python bench.py --steps 600 --warmup 30 --no-extras --no-cpu-baseline > gpurun_out/ab_$v.log 2>&1
  tail -1 gpurun_out/ab_$v.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', 'value %.4g  launch_us %.2f  frac %.3f  fused %.4g' % (d['value'], d['roofline']['launch_us'], d['roofline']['frac'], d.get('steps_per_call',{}).get('cell_updates_per_s_per_gpu',0)))"
done; done
