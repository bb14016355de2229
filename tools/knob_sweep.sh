#!/bin/bash
# Sweep of the WF_* experiment switches on the headline workload (one bench line per setting).
# usage: tools/knob_sweep.sh OUTFILE "VAR=val VAR=val" "VAR=val" ...   ("" = defaults)
out=$1; shift
: > "$out"
for cfg in "$@"; do
  echo "== $cfg" >> "$out"
  env $cfg python bench.py --no-extras --no-cpu-baseline --steps 2000 --warmup 100 2>> "$out.err" | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('us/step %.2f  frac %.3f  kernel %s x%d' % (d['roofline']['launch_us'], d['roofline']['frac'], d['roofline']['kernel'], d['roofline']['launches_per_step']))" >> "$out"
done
cat "$out"
