#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r4
cfgs=("" "WF_GROUP=4" "WF_GROUP=4 WF_SPLIT=3" "WF_GROUP=4 WF_SPLIT=4" "WF_SPLIT=3" "WF_GROUP=4 WF_SHARED=4" "WF_GROUP=4 WF_SHARED=4 WF_SPLIT=3" "")
bash tools/knob_sweep.sh gpurun_out/r4/sweep3.txt "${cfgs[@]}" > /dev/null
cat gpurun_out/r4/sweep3.txt | paste - -
