#!/bin/bash
cd "$GRAFT_REPO_ROOT"
run() { echo -n "[$*] "; env "$@" python bench.py --no-extras --no-cpu-baseline --steps 2000 --warmup 100 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step']*1e3,3), round(d['roofline']['frac'],4))"; }
run WF_X=1
run GPU_NUM_MEM_DEPENDENCY=0
run GPU_NUM_MEM_DEPENDENCY=1024
run GPU_FLUSH_ON_EXECUTION=1
run HSA_ENABLE_SDMA=0
run HSA_XNACK=0
run ROC_ACTIVE_WAIT_TIMEOUT=1000
run GPU_MAX_COMMAND_BUFFERS=32
run HIP_LAUNCH_BLOCKING=0
run HSA_ENABLE_INTERRUPT=0
run WF_X=1
