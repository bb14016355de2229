#!/bin/bash
# HIP runtime switches (process environment, read when the runtime initialises) against the headline step and the n-step call of configs[1]
cd "$GRAFT_REPO_ROOT"
run() { echo -n "[$*] "; env "$@" python bench.py --no-extras --no-cpu-baseline --steps 2000 --warmup 100 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step']*1e3,3), round(d['roofline']['frac'],4), 'x%d' % d['roofline']['launches_per_step'])" | tr "\n" " "; env "$@" python3 tools/step_loop.py --size 10 --batch 4096 --n 16 --noop --calls 300 --warm 160 2>/dev/null | sed "s/.*call: \([0-9.]*\) us per call.*x\([0-9]*\)\]$/ c2 \1 (x\2)/"; }
run WF_X=1
run HIP_FORCE_DEV_KERNARG=1
run HIP_FORCE_DEV_KERNARG=0
run AMD_OPT_FLUSH=0
run AMD_OPT_FLUSH=1
run AMD_DIRECT_DISPATCH=0
run GPU_MAX_HW_QUEUES=8
run GPU_MAX_HW_QUEUES=8 WF_SPLIT=4
run GPU_MAX_HW_QUEUES=2
run GPU_STREAMOPS_CP_WAIT=1
run WF_X=1
