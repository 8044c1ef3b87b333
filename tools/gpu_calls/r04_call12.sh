#!/bin/bash
# Round 4, GPU call 12: the tail blocks of the fused Swin MLP beside the main kernel (side stream): parity + end-to-end A/B; DCNv2 with 8 waves A/B
cd ${GRAFT_REPO_ROOT:-/root/repo}
rm -rf gpurun_out/*; mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "check_swin_mlp or swin_fpn or check_full_model or hip_graph_replay or b8_graph" > gpurun_out/r04c12_pytest.log 2>&1; grep -E "passed|failed|Error|assert|Fatal" gpurun_out/r04c12_pytest.log | tail -6 | cut -c1-800
for v in "MQ_SWIN_MLP_TAIL_STREAM=0" "MQ_SWIN_MLP_TAIL_STREAM=1" "MQ_SWIN_MLP_TAIL_STREAM=0" "MQ_SWIN_MLP_TAIL_STREAM=1" "MQ_DCN_WAVES=8"; do
  env $v timeout 300 python bench.py --steps 30 --warmup 3 --no-extras > gpurun_out/r04c12_ab_$v.log 2>&1
  echo "$v: rc=$? $(tail -1 gpurun_out/r04c12_ab_$v.log | cut -c1-200)"
done
