#!/bin/bash
# round 5, call 19: smoke() + the block / family / kernel-selection / graph tests that walk the text path, on the last commit
mkdir -p gpurun_out/c19
export GPU_MAX_HW_QUEUES=8
timeout 90 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE_OK')" 2>&1 | tail -3 | tee gpurun_out/c19/smoke.log
timeout 270 python -m pytest tests/test_gpu_parity.py -v --durations=12 -k "test_block or test_bf16_block or test_f32_block or test_groundingdino_block or test_alternate_kernel_selection or test_opt_in_kernel or test_benchmark_configuration_b8_graph_replay or test_backbone_and_caption_caches or test_staggered or test_mq_glip_l_family" > gpurun_out/c19/pytest.log 2>&1
grep -E "PASSED|FAILED|ERROR|passed|failed|s call" gpurun_out/c19/pytest.log | tail -40
