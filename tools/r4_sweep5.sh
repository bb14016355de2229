#!/bin/bash
# default (G by batch, private phases, 2 launches per step) against G=4 + phases A/C shared by 4 waves + 3 launches per step, by batch size and shape
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r4
for size in 14 10 12; do for b in 16384 32768 65536 125000 262144 1048576; do
  for cfg in "" "WF_GROUP=4 WF_SHARED=4 WF_SPLIT=3" "WF_GROUP=4 WF_SHARED=4 WF_SPLIT=2" "WF_GROUP=8 WF_SHARED=4 WF_SPLIT=3"; do
    calls=$((80000000 / b)); [ $calls -gt 1500 ] && calls=1500
    echo -n "size $size batch $b [$cfg]: "
    env $cfg python3 tools/step_loop.py --size $size --batch $b --n 1 --calls $calls --warm 200 2>/dev/null | sed 's/.*call = //'
  done
done; done
