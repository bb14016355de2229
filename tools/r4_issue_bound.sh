#!/bin/bash
# Is the rows kernel bound by vector-instruction issue?  The same kernel with 4 / 8 extra (unused) vector compares per cell slot
# (+10 % / +20 % vector instructions per environment-step) at the sizes of the bench legs.
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r4
for v in main x4 x8 main; do
  if [ "$v" = "main" ]; then unset WILDFIRE_LIB; else export WILDFIRE_LIB=$PWD/wildfire_control_python_amd/libwildfire_hip_$v.so; fi
  echo "== $v"
  python3 tools/step_loop.py --batch 65536 --n 1 --calls 1500 2>/dev/null
  WF_NO_SPLIT=1 python3 tools/step_loop.py --batch 65536 --n 1 --calls 1500 2>/dev/null
  python3 tools/step_loop.py --batch 65536 --n 16 --calls 100 2>/dev/null
  python3 tools/step_loop.py --batch 125000 --n 1 --calls 800 2>/dev/null
  python3 tools/step_loop.py --batch 262144 --n 1 --calls 400 2>/dev/null
  python3 tools/step_loop.py --batch 1048576 --n 1 --calls 100 --warm 200 2>/dev/null
done
