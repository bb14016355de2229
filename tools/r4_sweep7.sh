#!/bin/bash
cd "$GRAFT_REPO_ROOT"
for b in 125000 262144; do
for rep in 1 2; do
  for cfg in "WF_X=1" "WF_GROUP=8 WF_SHARED=2" "WF_GROUP=8 WF_SHARED=4" "WF_GROUP=8 WF_SHARED=8" "WF_GROUP=8 WF_SHARED=4 WF_SPLIT=3" "WF_GROUP=8 WF_SHARED=8 WF_SPLIT=3" "WF_GROUP=6 WF_SHARED=4" "WF_GROUP=6 WF_SHARED=4 WF_SPLIT=3" "WF_SPLIT=3"; do
    echo -n "batch $b [$cfg]: "
    env $cfg python3 tools/step_loop.py --batch $b --n 1 --calls 600 --warm 200 2>/dev/null | sed 's/.*call = //'
  done
done; done
