#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r4
cfgs=()
for rep in 1 2; do
for g in 4 8; do for sh in 0 2 4 8; do for sp in 2 3 4; do
  c="WF_GROUP=$g WF_SPLIT=$sp"; [ $sh != 0 ] && c="$c WF_SHARED=$sh"
  cfgs+=("$c")
done; done; done; done
bash tools/knob_sweep.sh gpurun_out/r4/sweep4.txt "${cfgs[@]}" > /dev/null
cat gpurun_out/r4/sweep4.txt | paste - - | sort | awk '{print}' 
