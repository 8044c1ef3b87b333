#!/bin/bash
# round 5, call 15: what the fused GCP kernel's time is made of (ablation launches) + the parity test of the kernel after adding the template parameter
mkdir -p gpurun_out/c15
export GPU_MAX_HW_QUEUES=8
MQ_MICRO_ONLY=gcp_attn timeout 240 python tools/microbench.py gpurun_out/c15/gcp_ablation.json > gpurun_out/c15/micro.log 2>&1
tail -4 gpurun_out/c15/micro.log
timeout 200 python -m pytest tests/test_gpu_parity.py -q -x -k "gcp" 2>&1 | tail -3
