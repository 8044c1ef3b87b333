#!/bin/bash
# Round 5, GPU call 11: the host side of a replayed step (memoised small inputs are not copied again into the graph's static buffers; the BoxLists of
# a batch are views of three batched tensors instead of three launches per image): tests + headline; A/Bs: the language front on the main stream
# instead of beside Swin, 4 / 2 hardware queues now that the default path forks only the text stream.
cd ${GRAFT_REPO_ROOT:-/root/repo}
rm -rf gpurun_out/*; mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "boundary or hip_graph or backbone_and_caption or b8_graph or score_aggregation or (test_block and check_full_model)" > gpurun_out/r05c11_pytest.log 2>&1; tail -3 gpurun_out/r05c11_pytest.log | cut -c1-300
for i in 1 2; do
  for env in "MQ_NONE=0" "MQ_FRONT_SIDE_STREAM=0" "GPU_MAX_HW_QUEUES=4" "GPU_MAX_HW_QUEUES=2"; do
    echo -n "$env: "; env $env timeout 90 python bench.py --steps 60 --warmup 3 --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'])"
  done
done > gpurun_out/r05c11_switch_ab.txt 2>&1; cat gpurun_out/r05c11_switch_ab.txt
timeout 300 python bench.py --workload lvis --chunk-batch 32 --steps 2 --warmup 2 --no-extras 2>/dev/null | tail -1 | cut -c1-200
