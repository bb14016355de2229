#!/bin/bash
# Long differential runs (tools/soak.py: every step rewards / done, every 25 steps planes, records, counts, observation)
# over the kernel families and the reference's off-by-default options touched in round 3
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r3_soak; mkdir -p $O
run() { name=$1; shift; env "$@" timeout -k 10 280 python tools/soak.py $STEPS $BATCH $SIZE $SEED > $O/$name.log 2>&1; echo "$name: $(tail -1 $O/$name.log | cut -c1-160)"; }
STEPS=400 BATCH=4100 SEED=11
SIZE=8  run rt8   OPTIONS=rivers,toggle
SIZE=12 run rt12  OPTIONS=rivers,toggle
SIZE=16 run rt16  OPTIONS=rivers,toggle
SIZE=13 run rt13  OPTIONS=rivers,toggle
SIZE=20 run rt20  OPTIONS=rivers,toggle
BATCH=1500 SIZE=32 run rt32 OPTIONS=rivers,toggle
SIZE=10 run w10   OPTIONS=wind
SIZE=12 run w12   OPTIONS=wind,rivers
SIZE=16 run w16   OPTIONS=wind,toggle
SIZE=8  run w8    OPTIONS=wind,rivers,toggle
SIZE=13 run w13   OPTIONS=wind,rivers,toggle
BATCH=9000 SIZE=14 run w14split OPTIONS=wind,rivers
BATCH=9000 SIZE=14 run w14multi OPTIONS=wind STEPS_PER_CALL=4
BATCH=9000 SIZE=10 run r10multi OPTIONS=rivers STEPS_PER_CALL=6
BATCH=9000 SIZE=10 run r10res OPTIONS=rivers,toggle STEPS_PER_CALL=6 WF_RESIDENT=1
# tiled family (every other step walks the batch backwards: WF_SWEEP_ALT, default on) and n-step calls on wf_step_ticks
STEPS=300
BATCH=600 SIZE=64 run t64 OPTIONS=rivers,toggle
BATCH=130 SIZE=128 run t128 OPTIONS=rivers
BATCH=130 SIZE=128 run t128fwd OPTIONS=rivers WF_SWEEP_ALT=0
STEPS=640
BATCH=4001 SIZE=10 run k10 OPTIONS=rivers,toggle STEPS_PER_CALL=7
BATCH=3000 SIZE=14 run k14 STEPS_PER_CALL=33
BATCH=8000 SIZE=16 run k16 STEPS_PER_CALL=16
