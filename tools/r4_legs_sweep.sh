#!/bin/bash
cd "$GRAFT_REPO_ROOT"
for cfg in "WF_X=1" "WF_SPLIT=3" "WF_SPLIT=4" "WF_GROUP=4" "WF_GROUP=4 WF_SPLIT=3" "WF_GROUP=16" "WF_SPLIT=2"; do
  echo -n "[$cfg] "
  env $cfg python bench.py --no-cpu-baseline --legs c3_wind,mid_64,c4 --steps 100 --warmup 10 --fused 0 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('  '.join('%s %.2f us (x%d)' % (k, v['us_per_call'], v['roofline']['launches_per_call']) for k,v in d['configs'].items()))"
done
