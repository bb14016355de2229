#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r3_s4; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "lent or resident or adapter or api_edge or plain_c or learner or host" > $O/tests.log 2>&1; echo "pytest rc=$?" | tee -a $O/tests.log
tail -4 $O/tests.log
for m in 0 1000000000; do
WF_HOST_STAGE_MIN=$m timeout -k 10 200 python bench.py --no-cpu-baseline --legs none > $O/bench_$m.json 2> $O/bench_$m.err; python -c "
import json; d=json.load(open('$O/bench_$m.json'))
print('stage_min $m: host_api', d.get('host_buffer_api')); print('adapter', d.get('single_env_adapter'))"
done
timeout -k 10 200 python tools/adapter_latency.py > $O/adapter_latency.txt 2>&1; cat $O/adapter_latency.txt
