#!/bin/bash
# Round 5, GPU call 1 (prepared at the end of round 4, whose GPU budget ended with call 22):
#   1. the GPU suite as the driver runs it;
#   2. the open A/Bs of round 4: the grouped DyConv epilogue (MQ_DYCONV_EPILOGUE_GROUPED) and the clamps of the fusion-layer BERT copies inside
#      their kernels (MQ_BERT_CLAMP_FUSED): default / each / both, 4 rounds in turn, 60 steps;
#   3. rocprofv3 kernel-trace stats + the two PMC traffic passes at HEAD (conv3x3_group_kernel, dyconv_fuse_group_kernel are new names);
#   4. the default bench line.
cd ${GRAFT_REPO_ROOT:-/root/repo}
rm -rf gpurun_out/*; mkdir -p gpurun_out
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
timeout 900 python -m pytest tests/ -x -q -m gpu > gpurun_out/r05c1_pytest.log 2>&1; tail -4 gpurun_out/r05c1_pytest.log | cut -c1-300
for i in 1 2 3 4; do
  for env in "MQ_NONE=0" "MQ_DYCONV_EPILOGUE_GROUPED=1" "MQ_BERT_CLAMP_FUSED=1" "MQ_DYCONV_EPILOGUE_GROUPED=1 MQ_BERT_CLAMP_FUSED=1"; do
    echo -n "$env: "; env $env timeout 60 python bench.py --steps 60 --warmup 3 --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'])"
  done
done > gpurun_out/r05c1_switch_ab.txt 2>&1; cat gpurun_out/r05c1_switch_ab.txt
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_default -o bench -- python $R/bench.py --steps 5 --warmup 2 --no-graph --no-extras > $R/gpurun_out/r05c1_prof.log 2>&1
f=$(find /tmp/prof_default -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $R/gpurun_out/r05c1_kernel_stats.csv && head -14 $f | cut -c1-150
cd $R
MQ_ROUND=r05 timeout 500 bash tools/pmc_traffic.sh > gpurun_out/r05c1_pmc.log 2>&1
python tools/pmc_reduce.py gpurun_out/pmc_r05 gpurun_out/r05_pmc_traffic.json > /dev/null 2>&1
( time timeout 600 python bench.py ) > gpurun_out/r05c1_bench.log 2> gpurun_out/r05c1_bench.time; tail -1 gpurun_out/r05c1_bench.log | cut -c1-400; tail -3 gpurun_out/r05c1_bench.time
