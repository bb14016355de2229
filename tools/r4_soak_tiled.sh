#!/bin/bash
# Long differential runs of the tiled family's one-launch step (device API, part launches on the context's streams) against the oracle:
# every step rewards / done, every `check` steps the whole state and the fused observation (tools/soak_device.py)
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r4_soak_tiled; mkdir -p $O
dev() { name=$1; shift; timeout -k 10 400 python tools/soak_device.py "$@" > $O/$name.log 2>&1; echo "$name: $(tail -1 $O/$name.log | cut -c1-200)"; }
dev t64 250 4096 64 5 25
dev t128 250 1024 128 7 25
dev t32 250 16384 32 9 25
dev t48 250 6000 48 3 25
dev t28 250 12000 28 19 25
dev t24 250 16000 24 23 25
dev t256_small 120 200 256 11 20
dev t100 200 1500 100 13 25
WF_TILED_ROWS=16 dev t64_rows16 150 4096 64 15 25
WF_NO_SOLO=1 dev t64_two_launch 150 4096 64 17 25
# the step loop inside the kernel (n-step calls of jobs of fewer than 8 192 waves; tools/soak.py: host API, state checked every 25 steps)
run() { name=$1; shift; env "$@" timeout -k 10 300 python tools/soak.py $STEPS $BATCH $SIZE $SEED > $O/$name.log 2>&1; echo "$name: $(tail -1 $O/$name.log | cut -c1-200)"; }
STEPS=480 SEED=21
BATCH=1024 SIZE=64 run loop64 STEPS_PER_CALL=16
BATCH=300 SIZE=128 run loop128 STEPS_PER_CALL=8
BATCH=3000 SIZE=32 run loop32 STEPS_PER_CALL=60
BATCH=4000 SIZE=24 run loop24_rare OPTIONS=rivers,toggle STEPS_PER_CALL=9
BATCH=1500 SIZE=48 run loop48_rare OPTIONS=rivers,toggle STEPS_PER_CALL=12
BATCH=100 SIZE=256 run loop256 STEPS_PER_CALL=5
BATCH=2500 SIZE=64 run loop64_parts STEPS_PER_CALL=16
