#!/bin/bash
# Round 3, GPU call 13 (last of the round): the final window_attn_qkv kernel (bias in LDS, unconditional X loads) -- microbench,
# device parity of the Swin path, and the default bench of the committed state (SWIN_QKV_FUSED = 2).
cd ${GRAFT_REPO_ROOT:-/root/repo}
rm -rf gpurun_out/*; mkdir -p gpurun_out
export TMPDIR=/tmp
MQ_MICRO_ONLY=window_qkv timeout 120 python tools/microbench.py gpurun_out/r03c13_micro_window_qkv.json > gpurun_out/r03c13_micro.log 2>&1; grep "kernel\|Error\|error" gpurun_out/r03c13_micro.log | cut -c1-200
timeout 200 python -m pytest tests -q -m gpu -k "window or swin_fpn or check_full_model or alternate" > gpurun_out/r03c13_pytest.log 2>&1; tail -3 gpurun_out/r03c13_pytest.log | cut -c1-300
( time timeout 280 python bench.py ) > gpurun_out/r03c13_bench_default.log 2> gpurun_out/r03c13_bench_default.time; tail -1 gpurun_out/r03c13_bench_default.log | cut -c1-300; tail -4 gpurun_out/r03c13_bench_default.time
