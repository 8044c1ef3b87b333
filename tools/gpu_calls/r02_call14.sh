#!/bin/bash
# GPU call 14: MQ-GroundingDINO bench line with the CPU baseline of the same family on the box's host cores.
cd ${GRAFT_REPO_ROOT:-/root/repo}
rm -rf gpurun_out/*; mkdir -p gpurun_out
timeout 170 python bench.py --workload mq-gdino-t --steps 10 --warmup 3 --cpu-baseline > gpurun_out/r02_bench14_gdino_b16.log 2>&1; tail -1 gpurun_out/r02_bench14_gdino_b16.log | cut -c1-200; grep -o '"cpu_baseline".*' gpurun_out/r02_bench14_gdino_b16.log | cut -c1-500
