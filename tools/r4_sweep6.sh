#!/bin/bash
cd "$GRAFT_REPO_ROOT"
for b in 24576 32768 40960 49152 57344 65536 73728 81920 98304; do
  echo -n "batch $b:"
  for cfg in "WF_X=0" "WF_GROUP=4 WF_SHARED=4 WF_SPLIT=2" "WF_GROUP=4 WF_SHARED=4 WF_SPLIT=3" "WF_GROUP=8 WF_SHARED=4 WF_SPLIT=3" "WF_GROUP=4 WF_SPLIT=3"; do
    env $cfg python3 tools/step_loop.py --batch $b --n 1 --calls 1500 --warm 300 2>/dev/null | sed 's/.*call = \([0-9.]*\) us per step.*/  \1/' | tr -d '\n'
  done; echo
done
