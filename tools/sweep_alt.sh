#!/bin/bash
# Experiment: big-grid steps walking the batch forwards and backwards in turn (WF_SWEEP_ALT=1) against always forwards:
# BASELINE configs[3] (1 024 x 256x256, 268 MB: just past the 256 MB Infinity Cache) and a mid size, us per step.
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/sweep_alt; mkdir -p $O
for alt in 0 1 0 1; do
  for split in "" 1; do
    if [ -n "$split" ]; then export WF_NO_SPLIT=1; else unset WF_NO_SPLIT; fi
    WF_SWEEP_ALT=$alt timeout -k 10 200 python bench.py --legs c4,mid_64 --no-cpu-baseline --steps 20 --warmup 5 --fused 0 > $O/b_$alt$split.json 2> $O/b_$alt$split.err || { echo failed; tail -3 $O/b_$alt$split.err; exit 1; }
    python - $alt "$split" $O/b_$alt$split.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[3]).read().strip().splitlines()[-1])
c = d.get("configs", {})
print(f"WF_SWEEP_ALT={sys.argv[1]} {'one launch sequence' if sys.argv[2] else 'three streams      '}: " + "  ".join(f"{k}: {v['us_per_call']:7.2f} us ({v['roofline']['frac']:.3f})" for k, v in c.items()))
PY
  done
done
unset WF_NO_SPLIT
WF_SWEEP_ALT=1 timeout -k 10 600 python -m pytest tests -x -q -m gpu -k "big_grid or config4 or tiled or arbitrary_uploaded" > $O/tests.log 2>&1; tail -3 $O/tests.log
