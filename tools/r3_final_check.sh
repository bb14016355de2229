#!/bin/bash
# What the driver runs at round end (GPU suite with -x, smoke, default bench), plus soak runs of the options touched in round 3
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r3_final; mkdir -p $O
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $O/gpu_tests.log 2>&1; rc=$?; echo "pytest rc=$rc" | tee -a $O/gpu_tests.log; tail -3 $O/gpu_tests.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $O/smoke.log; tail -3 $O/smoke.log
timeout -k 10 200 python tools/leak_check.py 60 > $O/leak.log 2>&1; cat $O/leak.log
OPTIONS=wind,rivers,toggle timeout -k 10 300 python tools/soak.py 500 8205 14 7 > $O/soak_wind.log 2>&1; tail -2 $O/soak_wind.log
OPTIONS=rivers timeout -k 10 300 python tools/soak.py 700 8205 10 8 > $O/soak_rivers.log 2>&1; tail -2 $O/soak_rivers.log
OPTIONS=rivers,toggle STEPS_PER_CALL=5 timeout -k 10 300 python tools/soak.py 600 20000 14 9 > $O/soak_multi.log 2>&1; tail -2 $O/soak_multi.log
timeout -k 10 500 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_args.json 2> $O/bench_driver_args.err; echo "bench rc=$?"
timeout -k 10 500 python bench.py > $O/bench_plain.json 2> $O/bench_plain.err; echo "bench rc=$?"
python -c "
import json
for f in ('bench_driver_args','bench_plain'):
    d=json.load(open('$O/'+f+'.json')); print(f, 'value %.4g' % d['value'], 'ms_per_step', d['ms_per_step'], 'frac %.3f' % d['roofline']['frac'], 'traffic', d['roofline']['traffic'], 'steps', d['steps'])
    print({k:(v['roofline']['frac'], v['roofline'].get('traffic')) for k,v in d.get('configs',{}).items()})"
