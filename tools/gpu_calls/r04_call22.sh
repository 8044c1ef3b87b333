#!/bin/bash
# Round 4, GPU call 22 (last of the round's budget): the model-level GPU tests on the new default (grouped offset conv), the grouped DyConv
# epilogue (MQ_DYCONV_EPILOGUE_GROUPED=1) as an isolated body, and the headline step with / without it on the same box
cd ${GRAFT_REPO_ROOT:-/root/repo}
rm -rf gpurun_out/*; mkdir -p gpurun_out
L=gpurun_out/r04c22.log
( timeout 85 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "test_hip_graph_replay_matches_eager or test_full_model_without_vision_queries or test_mq_glip_l_family or test_bf16_mq_glip_l_family or (test_block and (dyconv or full_model or conv3x3)) or (test_bf16_block and (dyconv or full_model))" 2>&1 | tail -4
  timeout 45 python tests/test_gpu_parity.py dyconv_epilogue_group 2>&1 | tail -3
  for v in 1 0 1; do
    echo "MQ_DYCONV_EPILOGUE_GROUPED=$v"; MQ_DYCONV_EPILOGUE_GROUPED=$v timeout 40 python bench.py --steps 40 --warmup 3 --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'])"
  done ) > $L 2>&1
cat $L | cut -c1-400
