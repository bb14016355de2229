#!/bin/bash
# soak runs of wf_step_ticks (n-step calls of batches that are not split) against the oracle + the configs[1] variant table
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r3_ticks; mkdir -p $O
OPTIONS=rivers,toggle STEPS_PER_CALL=7 timeout -k 10 300 python tools/soak.py 700 4001 10 11 > $O/soak_a.log 2>&1; tail -2 $O/soak_a.log
STEPS_PER_CALL=33 timeout -k 10 300 python tools/soak.py 660 3000 14 12 > $O/soak_b.log 2>&1; tail -2 $O/soak_b.log
STEPS_PER_CALL=16 timeout -k 10 300 python tools/soak.py 640 8000 16 13 > $O/soak_c.log 2>&1; tail -2 $O/soak_c.log
OPTIONS=rivers STEPS_PER_CALL=64 timeout -k 10 300 python tools/soak.py 640 2000 8 14 > $O/soak_d.log 2>&1; tail -2 $O/soak_d.log
timeout -k 10 400 python tools/c2_variants.py > $O/c2_variants.txt 2>&1; cat $O/c2_variants.txt
