#!/bin/bash
# What the driver runs at round end (GPU suite with -x, smoke, default bench) on the final sources, plus the bench records kept under profiles/
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r4_final; mkdir -p $O
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $O/gpu_tests.log 2>&1; rc=$?; echo "pytest rc=$rc" | tee -a $O/gpu_tests.log; tail -3 $O/gpu_tests.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $O/smoke.log; tail -3 $O/smoke.log
timeout -k 10 500 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_args.json 2> $O/bench_driver_args.err; echo "bench rc=$?"
timeout -k 10 500 python bench.py > $O/bench_plain.json 2> $O/bench_plain.err; echo "bench rc=$?"
python -c "
import json
for f in ('bench_driver_args','bench_plain'):
    d=json.load(open('$O/'+f+'.json')); print(f, 'value %.4g' % d['value'], 'ms_per_step', d['ms_per_step'], 'frac %.3f' % d['roofline']['frac'], 'traffic', d['roofline']['traffic'], 'steps', d['steps'])
    print({k:(round(v['roofline']['frac'],3), (v['roofline'].get('traffic') or {}).get('hbm_bytes_per_call') if isinstance(v['roofline'].get('traffic'),dict) else v['roofline'].get('traffic')) for k,v in d.get('configs',{}).items()})"
