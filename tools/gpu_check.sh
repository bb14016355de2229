#!/bin/bash
# usage (on the GPU box, via gpurun): bash tools/gpu_check.sh <tag> [pytest|nopytest]
# parity tests -> bench -> PMC instruction counters, everything under gpurun_out/
tag=${1:-x}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
if [ "${2:-pytest}" = "pytest" ]; then
  timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_$tag.log 2>&1
  tail -4 gpurun_out/pytest_$tag.log
fi
timeout -k 10 300 python bench.py --steps 100 --warmup 20 --no-cpu-baseline > gpurun_out/bench_$tag.log 2>&1
tail -1 gpurun_out/bench_$tag.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('value %.4g  launch_us %.2f  frac %.3f  fused %.4g' % (d['value'], d['roofline']['launch_us'], d['roofline']['frac'], d.get('steps_per_call',{}).get('cell_updates_per_s_per_gpu',0)))"
rm -rf gpurun_out/pmc_$tag
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD --output-format csv -d gpurun_out/pmc_$tag -- python3 bench.py --steps 10 --warmup 2 --fused 0 --no-cpu-baseline > gpurun_out/pmc_$tag.log 2>&1
python tools/pmc_summary.py gpurun_out/pmc_$tag | grep -A9 "wf_step_"
