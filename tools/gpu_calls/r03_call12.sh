#!/bin/bash
# Round 3, GPU call 12: the qkv projection inside the window attention at C = 192 as well (weights streamed per head): microbench,
# the whole GPU suite with MQ_SWIN_QKV_FUSED=2, end-to-end A/B, default bench as the driver runs it.
cd ${GRAFT_REPO_ROOT:-/root/repo}
rm -rf gpurun_out/*; mkdir -p gpurun_out
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
MQ_MICRO_ONLY=window_qkv timeout 200 python tools/microbench.py gpurun_out/r03c12_micro_window_qkv.json > gpurun_out/r03c12_micro.log 2>&1; grep "kernel\|Error\|error" gpurun_out/r03c12_micro.log | cut -c1-220
for v in MQ_SWIN_QKV_FUSED=2 NONE=0 MQ_SWIN_QKV_FUSED=2 NONE=1; do
  env $v timeout 200 python bench.py --steps 30 --warmup 3 --no-extras > gpurun_out/r03c12_ab_$v.log 2>&1
  echo "$v: $(tail -1 gpurun_out/r03c12_ab_$v.log | cut -c1-140)"
done
MQ_SWIN_QKV_FUSED=2 MQ_LADDER_OUT=$R/gpurun_out/r03c12_ladder.jsonl timeout 1200 python -m pytest tests -q -m gpu > gpurun_out/r03c12_pytest.log 2>&1; tail -6 gpurun_out/r03c12_pytest.log | cut -c1-400
( time timeout 600 python bench.py ) > gpurun_out/r03c12_bench_default.log 2> gpurun_out/r03c12_bench_default.time; tail -1 gpurun_out/r03c12_bench_default.log | cut -c1-300; tail -4 gpurun_out/r03c12_bench_default.time
