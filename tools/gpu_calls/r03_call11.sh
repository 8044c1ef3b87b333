#!/bin/bash
# Round 3, GPU call 11: window attention with the qkv projection inside (first device run): microbench, parity, end-to-end A/B.
cd ${GRAFT_REPO_ROOT:-/root/repo}
rm -rf gpurun_out/*; mkdir -p gpurun_out
export TMPDIR=/tmp
MQ_MICRO_ONLY=window_qkv timeout 200 python tools/microbench.py gpurun_out/r03c11_micro_window_qkv.json > gpurun_out/r03c11_micro.log 2>&1; grep "kernel\|Error\|error" gpurun_out/r03c11_micro.log | cut -c1-220
timeout 400 python -m pytest tests -q -m gpu -k "window or swin_fpn or check_full_model or fusion_layer" > gpurun_out/r03c11_pytest.log 2>&1; tail -4 gpurun_out/r03c11_pytest.log | cut -c1-300
for v in NONE=0 MQ_SWIN_QKV_FUSED=0 NONE=1 MQ_SWIN_QKV_FUSED=0; do
  env $v timeout 200 python bench.py --steps 30 --warmup 3 --no-extras > gpurun_out/r03c11_ab_$v.log 2>&1
  echo "$v: $(tail -1 gpurun_out/r03c11_ab_$v.log | cut -c1-140)"
done
