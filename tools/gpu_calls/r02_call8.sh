#!/bin/bash
# GPU call 8: MQ-GroundingDINO bench (B = 16 and B = 8) + rocprof kernel stats of the B = 16 run; GDINO pytest group.
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python bench.py --workload mq-gdino-t --steps 10 --warmup 3 > gpurun_out/r02_bench8_gdino_b16.log 2>&1; tail -1 gpurun_out/r02_bench8_gdino_b16.log | cut -c1-1500
timeout 600 python bench.py --workload mq-gdino-t --batch 4 --steps 10 --warmup 3 > gpurun_out/r02_bench8_gdino_b4.log 2>&1; tail -1 gpurun_out/r02_bench8_gdino_b4.log | cut -c1-400
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d ${GRAFT_REPO_ROOT}/gpurun_out/r02_prof8 -o bench -- python ${GRAFT_REPO_ROOT}/bench.py --workload mq-gdino-t --steps 5 --warmup 2 > ${GRAFT_REPO_ROOT}/gpurun_out/r02_bench8_prof.log 2>&1
cd ${GRAFT_REPO_ROOT}
f=$(find gpurun_out/r02_prof8 -name "*kernel_stats.csv" | head -1); head -40 $f | cut -c1-200
timeout 900 python -m pytest tests/test_gpu_parity.py -q -k "groundingdino" 2>&1 | tail -5
