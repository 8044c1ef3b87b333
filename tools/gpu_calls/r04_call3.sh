#!/bin/bash
# Round 4, GPU call 3: the staggered micro-batch schedule (detector._staggered_program): parity test + end-to-end A/B (1 / 2 / 4 lanes, HW queues)
cd ${GRAFT_REPO_ROOT:-/root/repo}
rm -rf gpurun_out/*; mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "staggered or hip_graph_replay" > gpurun_out/r04c3_pytest.log 2>&1; grep -E "passed|failed|Error|assert" gpurun_out/r04c3_pytest.log | tail -5 | cut -c1-600
for v in "MQ_MICRO_BATCHES=1" "MQ_MICRO_BATCHES=2" "MQ_MICRO_BATCHES=4" "MQ_MICRO_BATCHES=2 GPU_MAX_HW_QUEUES=8" "MQ_MICRO_BATCHES=4 GPU_MAX_HW_QUEUES=8" "MQ_MICRO_BATCHES=1 GPU_MAX_HW_QUEUES=8" "MQ_MICRO_BATCHES=1"; do
  n=$(echo $v | tr ' ' '_')
  env $v timeout 200 python bench.py --steps 30 --warmup 3 --no-extras > gpurun_out/r04c3_ab_$n.log 2>&1
  echo "$v: $(tail -1 gpurun_out/r04c3_ab_$n.log | cut -c1-200)"
done
