#!/bin/bash
# Round-3 GPU session 3: wind on the wave family + everything else; then the bench line
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r3_s3; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "wind" > $O/tests_wind.log 2>&1; rc=$?
tail -5 $O/tests_wind.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 900 python -m pytest tests -m gpu -q > $O/gpu_tests.log 2>&1; echo "pytest rc=$?" | tee -a $O/gpu_tests.log
tail -4 $O/gpu_tests.log
timeout -k 10 400 python bench.py > $O/bench.json 2> $O/bench.err; python -c "
import json; d=json.load(open('$O/bench.json'))
print('headline us', d['roofline']['launch_us'], 'frac', d['roofline']['frac'], 'single', d['roofline'].get('single_launch'))
print('steps_per_call', d.get('steps_per_call')); print('obs', d.get('step_plus_observation'))
print('host_api', d.get('host_buffer_api')); print('adapter', d.get('single_env_adapter'))
for k, v in d['configs'].items(): print(k, v.get('us_per_call'), v.get('roofline', {}).get('frac'), v.get('roofline', {}).get('kernel'), v.get('get_state'), v.get('error'))
print('cpu', d['cpu_baseline']['value'], d['cpu_baseline']['cores'])"
