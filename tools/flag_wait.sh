#!/bin/bash
# Experiment: synchronous host-buffer calls waiting on a pinned completion flag (WF_FLAG_WAIT=us) against hipStreamSynchronize.
# usage: gpurun -- 'bash tools/flag_wait.sh'   -> gpurun_out/flag_wait.txt
set -o pipefail
mkdir -p gpurun_out
out=gpurun_out/flag_wait.txt
: > $out
for fw in 0 300 0 300 5000; do
  WF_FLAG_WAIT=$fw timeout -k 10 240 python bench.py --legs none --no-cpu-baseline --steps 300 --warmup 20 > gpurun_out/flag_wait_$fw.json 2> gpurun_out/flag_wait_$fw.err || { echo "WF_FLAG_WAIT=$fw failed" >> $out; tail -5 gpurun_out/flag_wait_$fw.err >> $out; exit 1; }
  python - $fw >> $out <<'PY'
import json, sys
fw = sys.argv[1]
r = json.loads(open(f"gpurun_out/flag_wait_{fw}.json").read().strip().splitlines()[-1])
h, a = r.get("host_buffer_api", {}), r.get("single_env_adapter", {})
print(f"WF_FLAG_WAIT={fw}: wf_step us {h.get('us_per_step'):.1f}  lent {h.get('lent_buffers_us_per_step'):.1f}  adapter {a}")
PY
done
cat $out
