#!/bin/bash
# Sweep of the WF_* experiment switches (one bench line per setting).
# usage: tools/knob_sweep.sh OUTFILE "VAR=val VAR=val [-- bench args]" ...   ("" = defaults)
out=$1; shift
: > "$out"
for cfg in "$@"; do
  envs="${cfg%%--*}"; args=""
  if [[ "$cfg" == *"--"* ]]; then args="--${cfg#*--}"; fi
  echo "== $cfg" >> "$out"
  env $envs python bench.py --no-extras --no-cpu-baseline --steps 1500 --warmup 100 $args 2>> "$out.err" | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('us/step %.2f  frac %.3f  kernel %s x%d' % (d['roofline']['launch_us'], d['roofline']['frac'], d['roofline']['kernel'], d['roofline']['launches_per_step']))" >> "$out"
done
cat "$out"
