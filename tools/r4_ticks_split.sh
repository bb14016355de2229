#!/bin/bash
# n-step calls of wf_step_ticks as 1 launch against 2 / 3 / 4 part launches on the context's streams
cd "$GRAFT_REPO_ROOT"
for size in 10 14 8; do for b in 512 1024 2048 4096 8192 16384; do for n in 16; do
  echo -n "size $size batch $b n=$n:"
  for cfg in "WF_TICKS_SPLIT=0" "WF_TICKS_SPLIT=1 WF_SPLIT=2" "WF_TICKS_SPLIT=1 WF_SPLIT=3" "WF_TICKS_SPLIT=1 WF_SPLIT=4"; do
    env $cfg python3 tools/step_loop.py --size $size --batch $b --n $n --noop --calls 300 --warm 160 2>/dev/null | sed 's/.*call: \([0-9.]*\) us per call.*x\([0-9]*\)\]$/  \1 (x\2)/' | tr -d '\n'
  done; echo
done; done; done
echo "with actions:"
for b in 2048 4096 8192; do echo -n "size 10 batch $b n=16 actions:"; for cfg in "WF_TICKS_SPLIT=0" "WF_TICKS_SPLIT=1 WF_SPLIT=2"; do env $cfg python3 tools/step_loop.py --size 10 --batch $b --n 16 --calls 300 --warm 160 2>/dev/null | sed 's/.*call: \([0-9.]*\) us per call.*/  \1/' | tr -d '\n'; done; echo; done
