#!/bin/bash
# Round 4, GPU call 5: software-pipelined lanes (flat forks): capture probe, parity test, end-to-end A/B (lanes x HW queues)
cd ${GRAFT_REPO_ROOT:-/root/repo}
rm -rf gpurun_out/*; mkdir -p gpurun_out
export TMPDIR=/tmp
env MQ_MICRO_BATCHES=2 timeout 120 python tools/stagger_probe.py > gpurun_out/r04c5_probe.log 2>&1; echo "probe rc=$? $(grep -E 'PROBE_OK|Error' gpurun_out/r04c5_probe.log | tail -2 | cut -c1-200)"
timeout 300 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "staggered" > gpurun_out/r04c5_pytest.log 2>&1; grep -E "passed|failed|Error|assert|Fatal" gpurun_out/r04c5_pytest.log | tail -5 | cut -c1-600
for v in "MQ_MICRO_BATCHES=1" "MQ_MICRO_BATCHES=2" "MQ_MICRO_BATCHES=4" "MQ_MICRO_BATCHES=2 GPU_MAX_HW_QUEUES=8" "MQ_MICRO_BATCHES=4 GPU_MAX_HW_QUEUES=8" "MQ_MICRO_BATCHES=1 GPU_MAX_HW_QUEUES=8" "MQ_MICRO_BATCHES=1 GPU_MAX_HW_QUEUES=16" "MQ_MICRO_BATCHES=1"; do
  n=$(echo $v | tr ' ' '_')
  env $v timeout 200 python bench.py --steps 30 --warmup 3 --no-extras > gpurun_out/r04c5_ab_$n.log 2>&1
  echo "$v: rc=$? $(tail -1 gpurun_out/r04c5_ab_$n.log | cut -c1-200)"
done
