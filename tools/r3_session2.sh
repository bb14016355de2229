#!/bin/bash
# Round-3 GPU session 2: resident step loop - correctness first, then what it buys
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r3_s2; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "resident or multi_step or config2 or rows_layout or fused or kernel_switch or split_fire or arbitrary" > $O/tests_resident.log 2>&1; rc=$?
tail -5 $O/tests_resident.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 200 python tools/c2_variants.py > $O/c2_variants.txt 2>&1; cat $O/c2_variants.txt
timeout -k 10 200 python tools/graph_steps.py > $O/graph_steps.txt 2>&1; cat $O/graph_steps.txt
timeout -k 10 300 python bench.py --no-cpu-baseline --legs c2 > $O/bench_c2.json 2> $O/bench_c2.err; python -c "
import json; d=json.load(open('$O/bench_c2.json')); print('headline', d['roofline']['launch_us'], d['roofline']['frac']); print('steps_per_call', d.get('steps_per_call')); print('c2', d['configs']['c2']['us_per_call'], d['configs']['c2']['roofline']['frac'], d['configs']['c2']['roofline']['kernel']); print('host_api', d.get('host_buffer_api')); print('adapter', d.get('single_env_adapter'))"
