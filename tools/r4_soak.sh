#!/bin/bash
# Long differential runs on the round-4 paths: the device API in the one-generation shape (three part launches, shared agent phases), in the
# two-halves shape, and n-step calls of wf_step_ticks as two / four part launches (tools/soak.py: host API)
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r4_soak; mkdir -p $O
dev() { name=$1; shift; timeout -k 10 500 python tools/soak_device.py "$@" > $O/$name.log 2>&1; echo "$name: $(tail -1 $O/$name.log | cut -c1-200)"; }
run() { name=$1; shift; env "$@" timeout -k 10 280 python tools/soak.py $STEPS $BATCH $SIZE $SEED > $O/$name.log 2>&1; echo "$name: $(tail -1 $O/$name.log | cut -c1-160)"; }
dev d65536 300 65536 14 5 25
dev d49152 200 49152 14 7 20
dev d90000 150 90000 14 9 25
dev d125000 100 125000 14 3 25
dev d10x10 200 65536 10 11 25
STEPS=640 SEED=11
BATCH=4001 SIZE=10 run k10 OPTIONS=rivers,toggle STEPS_PER_CALL=7
BATCH=4096 SIZE=10 run k10b STEPS_PER_CALL=16
BATCH=3000 SIZE=14 run k14 STEPS_PER_CALL=33
BATCH=8000 SIZE=16 run k16 STEPS_PER_CALL=16
BATCH=2100 SIZE=8 run k8 STEPS_PER_CALL=5
