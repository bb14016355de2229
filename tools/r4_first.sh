#!/bin/bash
# round 4, first GPU call: issue-rate ubench, the lent-buffer anomaly of the driver's arguments, instruction-count sensitivity
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r4
timeout -k 10 120 tools/ubench/bin/issue2 > gpurun_out/r4/issue2.txt 2>&1
WF_BENCH_DEBUG=1 timeout -k 10 300 python bench.py --steps 20 --warmup 5 --legs none > gpurun_out/r4/bench_driver.json 2> gpurun_out/r4/bench_driver.err
WF_BENCH_DEBUG=1 timeout -k 10 300 python bench.py --legs none --no-cpu-baseline > gpurun_out/r4/bench_plain.json 2> gpurun_out/r4/bench_plain.err
bash tools/knob_sweep.sh gpurun_out/r4/sweep1.txt "" "WF_SHARED=1" "WF_NO_SPLIT=1" "WF_SHARED=1 WF_NO_SPLIT=1" "WF_GROUP=4" "WF_GROUP=4 WF_NO_SPLIT=1" "WF_SPLIT=3" "WF_SPLIT=4" ""
