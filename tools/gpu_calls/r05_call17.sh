#!/bin/bash
# round 5, call 17: the two fused text kernels with their weights in MFMA B-fragment order (1 KiB consecutive per load instruction): timings + parity
mkdir -p gpurun_out/c17
export GPU_MAX_HW_QUEUES=8
MQ_MICRO_ONLY=gcp_attn timeout 200 python tools/microbench.py gpurun_out/c17/gcp_attn.json > gpurun_out/c17/micro_gcp.log 2>&1; tail -2 gpurun_out/c17/micro_gcp.log
MQ_MICRO_ONLY=bert_attn timeout 200 python tools/microbench.py gpurun_out/c17/bert_attn.json > gpurun_out/c17/micro_bert.log 2>&1; tail -4 gpurun_out/c17/micro_bert.log
timeout 300 python -m pytest tests/test_gpu_parity.py -q -x -k "gcp or bert" 2>&1 | tail -3
