#!/bin/bash
# Round-3 GPU session 1: the whole GPU suite, then the per-call-cost experiments (usage on the GPU box: bash tools/r3_session1.sh)
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r3_s1; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -q > $O/gpu_tests.log 2>&1; echo "pytest rc=$?" | tee -a $O/gpu_tests.log
tail -3 $O/gpu_tests.log
B="python bench.py --no-extras --no-cpu-baseline --steps 1500 --warmup 100"
line() { python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('$1: us/step %.2f  frac %.3f  kernel %s x%d' % (d['roofline']['launch_us'], d['roofline']['frac'], d['roofline']['kernel'], d['roofline']['launches_per_step']))"; }
{
for rep in 1 2; do
for v in main wpb8 wpb16; do
  if [ "$v" = "main" ]; then unset WILDFIRE_LIB; else export WILDFIRE_LIB=$PWD/wildfire_control_python_amd/libwildfire_hip_$v.so; fi
  timeout -k 10 120 $B 2>>$O/ab.err | line "$v split"
  WF_NO_SPLIT=1 timeout -k 10 120 $B 2>>$O/ab.err | line "$v single-launch"
done; done
unset WILDFIRE_LIB
WF_GRAPH=1 timeout -k 10 120 $B 2>>$O/ab.err | line "graph of one step (fork, 2 launches, join)"
WF_GRAPH=1 WF_NO_SPLIT=1 timeout -k 10 120 $B 2>>$O/ab.err | line "graph of one step, one launch"
} > $O/per_call_cost.txt 2>&1
cat $O/per_call_cost.txt
timeout -k 10 200 python tools/graph_steps.py > $O/graph_steps.txt 2>&1; cat $O/graph_steps.txt
timeout -k 10 200 python tools/c2_variants.py > $O/c2_variants.txt 2>&1; cat $O/c2_variants.txt
timeout -k 10 200 python tools/adapter_latency.py > $O/adapter_latency.txt 2>&1; cat $O/adapter_latency.txt
timeout -k 10 200 python tools/get_state_rate.py $O/get_state_rate.json > $O/get_state_rate.txt 2>&1; cat $O/get_state_rate.txt
