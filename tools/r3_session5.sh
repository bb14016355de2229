#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r3_s5; mkdir -p $O
for mode in 1 2 3; do
WF_HOST_STAGE_MODE=$mode timeout -k 10 200 python bench.py --no-cpu-baseline --legs none > $O/bench_m$mode.json 2> $O/bench_m$mode.err; python -c "
import json; d=json.load(open('$O/bench_m$mode.json')); h=d.get('host_buffer_api')
print('mode $mode: host_api us', h['us_per_step'], 'lent', h['lent_buffers_us_per_step'])"
done
WF_HOST_STAGE_MIN=1000000000 timeout -k 10 200 python bench.py --no-cpu-baseline --legs none > $O/bench_off.json 2> $O/bench_off.err; python -c "
import json; d=json.load(open('$O/bench_off.json')); h=d.get('host_buffer_api')
print('no mirrors: host_api us', h['us_per_step'], 'lent', h['lent_buffers_us_per_step'])"
timeout -k 10 900 python -m pytest tests -m gpu -q > $O/gpu_tests.log 2>&1; echo "pytest rc=$?" | tee -a $O/gpu_tests.log
tail -4 $O/gpu_tests.log
