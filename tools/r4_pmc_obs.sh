#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r4/pmc_obs; rm -rf $out; mkdir -p $out
export WF_NO_SPLIT=1
rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -- python3 tools/step_obs_loop.py --size 64 --batch 4096 --calls 100 > $out/kt.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS --output-format csv -d $out/pmc -- python3 tools/step_obs_loop.py --size 64 --batch 4096 --calls 50 > $out/pmc.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmcw -- python3 tools/step_obs_loop.py --size 64 --batch 4096 --calls 50 > $out/pmcw.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmcf -- python3 tools/step_obs_loop.py --size 64 --batch 4096 --calls 50 > $out/pmcf.log 2>&1
find $out/kt -name "*kernel_stats.csv" | head -1 | xargs cat | grep -E "Name|wf_"
for d in pmc pmcw pmcf; do echo "== $d"; python tools/pmc_summary.py $out/$d 1 | grep -A10 "wf_step_tiled"; done
