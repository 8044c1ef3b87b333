#!/bin/bash
# GPU call 9: whole GPU suite + default bench line + MQ-GroundingDINO bench (kernel / stage breakdown).  Tight timeouts: call 8 lost
# 38 minutes in a rocprofv3 run of the GroundingDINO bench (trace of ~10^5 kernels) -- no rocprof on that workload.
cd ${GRAFT_REPO_ROOT:-/root/repo}
rm -rf gpurun_out/*; mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests -q -m gpu > gpurun_out/r02_pytest9.log 2>&1; tail -4 gpurun_out/r02_pytest9.log
timeout 150 python bench.py --workload mq-gdino-t --steps 10 --warmup 3 > gpurun_out/r02_bench9_gdino_b16.log 2>&1; tail -1 gpurun_out/r02_bench9_gdino_b16.log | cut -c1-300
MQ_GDINO_FUSED_RELU=0 timeout 150 python bench.py --workload mq-gdino-t --steps 10 --warmup 3 > gpurun_out/r02_bench9_gdino_b16_norelu.log 2>&1; tail -1 gpurun_out/r02_bench9_gdino_b16_norelu.log | cut -c1-300
timeout 200 python bench.py --steps 10 --warmup 3 --no-lang-b64 > gpurun_out/r02_bench9_default.log 2>&1; tail -1 gpurun_out/r02_bench9_default.log | cut -c1-300
