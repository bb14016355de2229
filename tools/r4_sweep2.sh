#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r4
cfgs=()
for g in 2 4 8 16; do cfgs+=("WF_GROUP=$g" "WF_GROUP=$g WF_NO_SPLIT=1"); done
for sh in 2 4 8; do for g in 2 4 8; do cfgs+=("WF_SHARED=$sh WF_GROUP=$g" "WF_SHARED=$sh WF_GROUP=$g WF_NO_SPLIT=1"); done; done
bash tools/knob_sweep.sh gpurun_out/r4/sweep2.txt "${cfgs[@]}" > /dev/null
cat gpurun_out/r4/sweep2.txt | paste - - 
